#!/usr/bin/env python3
"""Per-kernel (and per grid size) durations out of a rocprofv3 rocpd database: python tools/rocpd_kernels.py DB [name-substring ...]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pats = sys.argv[2:] or [""]
where = " or ".join("name like ?" for _ in pats)
q = f"select name, grid_x * max(grid_y, 1), count(*), avg(end - start) / 1000.0, min(end - start) / 1000.0 from kernels where {where} group by name, grid_x * max(grid_y, 1) order by name, 2"
print(f"{'kernel':70s} {'grid':>10s} {'calls':>6s} {'avg us':>10s} {'min us':>10s}")
for name, grid, calls, avg, mn in c.execute(q, [f"%{p}%" for p in pats]):
    print(f"{name[:70]:70s} {grid:10d} {calls:6d} {avg:10.2f} {mn:10.2f}")
