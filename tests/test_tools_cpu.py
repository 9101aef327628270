"""Host-side evidence tooling (no GPU): the two scripts that turn rocprofv3 output into the records DESIGN.md and
bench.py quote -- the PMC traffic record (gfx950 FETCH_SIZE correction, full-size dispatches only) and the mean duration
of the full-size launches of a kernel trace."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "void gpso::leaf_tiles_bf16_kernel<2, float, 0, true>(unsigned int const*)"


def _write_csv(path, header, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(header)
        w.writerows(rows)


def test_pmc_traffic_record(tmp_path):
    head = ["Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"]
    # pass 1: two full-size dispatches and one short one (the self-test's) -- only the full-size ones count
    _write_csv(str(tmp_path / "pmc1" / "x" / "1_counter_collection.csv"), head,
               [[KERNEL, 1048576, "FETCH_SIZE", 1000.0], [KERNEL, 1048576, "FETCH_SIZE", 3000.0], [KERNEL, 4096, "FETCH_SIZE", 9.0],
                [KERNEL, 1048576, "GRBM_GUI_ACTIVE", 8.0e6], ["void other_kernel()", 1048576, "FETCH_SIZE", 7.0e9]])
    _write_csv(str(tmp_path / "pmc2" / "x" / "2_counter_collection.csv"), head,
               [[KERNEL, 1048576, "WRITE_SIZE", 500.0], [KERNEL, 4096, "WRITE_SIZE", 1.0]])
    _write_csv(str(tmp_path / "pmc3" / "x" / "3_counter_collection.csv"), head,
               [[KERNEL, 1048576, "SQ_VALU_MFMA_BUSY_CYCLES", 5.12e8]])
    out = str(tmp_path / "rec.json")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic_json.py"), out, "c3", "leaf_tiles_bf16_kernel<2",
                    "test kernel", str(tmp_path / "pmc1"), str(tmp_path / "pmc2"), str(tmp_path / "pmc3")], check=True,
                   capture_output=True)
    rec = json.load(open(out))
    assert rec["FETCH_SIZE_KB"] == 2000.0 and rec["WRITE_SIZE_KB"] == 500.0 and rec["grid_threads"] == 1048576
    # FETCH_SIZE doubled (64 B counted per 128-B request on gfx950), WRITE_SIZE as read; both in KB
    assert rec["traffic_bytes_per_launch"] == int(2 * 2000.0 * 1024 + 500.0 * 1024)
    assert abs(rec["matrix_pipe_busy"] - 5.12e8 / 1024.0 / (8.0e6 / 8.0)) < 1e-12
    assert rec["dispatches_averaged"]["FETCH_SIZE"] == 2


def test_full_size_launch_mean(tmp_path):
    head = ["Kernel_Name", "Grid_Size", "Start_Timestamp", "End_Timestamp"]
    _write_csv(str(tmp_path / "prof" / "x" / "1_kernel_trace.csv"), head,
               [[KERNEL, 1048576, 1000, 1800], [KERNEL, 1048576, 5000, 5900], [KERNEL, 8192, 7000, 7200],
                ["void other_kernel()", 1048576, 0, 10 ** 9]])
    out = str(tmp_path / "full.json")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_fullsize.py"), str(tmp_path / "prof"),
                    "leaf_tiles_bf16_kernel<2", out], check=True, capture_output=True)
    rec = json.load(open(out))
    assert rec["launches"] == 3 and rec["full_size_launches"] == 2
    assert rec["mean_ns_full_size"] == 850.0 and abs(rec["mean_ns_all_launches"] - (800 + 900 + 200) / 3) < 1e-9


def test_roofline_fit_prices_a_fit_against_the_arithmetic_that_ran():
    """``roofline_fit`` always used the f32 MFMA peak -- a two-level float fit on fp16 pieces came out at
    frac 1.388.  The peak now follows HipGPEngine.fit_math(); the committed round-5 times give fractions below 1."""
    sys.path.insert(0, ROOT)
    import bench

    c5 = bench.roofline_fit(16384, 40, "float32", {"posterior": 14.6, "nlml_grad": 20.2, "append_k7": 0.40}, "f16x3")
    assert c5["peak"] == 2500.0 / 3 and c5["fit_math"] == "f16x3"
    assert abs(c5["nlml_grad"]["frac"] - 0.26) < 0.01 and abs(c5["posterior"]["frac"] - 0.12) < 0.01  # (the verdict's recomputation)
    assert bench.roofline_fit(8192, 20, "float32", {"posterior": 3.7}, "bf16x6")["peak"] == 2500.0 / 6
    c3 = bench.roofline_fit(2048, 12, "float32", {"posterior": 0.556}, "f32")
    assert c3["peak"] == 157.3 and abs(c3["posterior"]["frac"] - 0.034) < 0.002
    f64 = bench.roofline_fit(8192, 20, "mixed", {"posterior": 9.64}, "f64")
    assert f64["peak"] == 78.6 and abs(f64["posterior"]["frac"] - 0.24) < 0.01
    for rec in (c5, c3, f64):
        assert all(rec[k]["frac"] < 1.0 for k in ("posterior", "nlml_grad", "append_k7") if k in rec)


def test_dma_barrier_check_flags_a_barrier_behind_a_back_edge(tmp_path):
    """tools/check_dma_barriers.py (round 6): a wave must not reach an s_barrier with LDS-DMAs of its own possibly in flight.  The
    shape hipcc 7.2 produced for the split kernels -- DMAs issued at the END of a loop body, the publishing barrier at the loop's
    HEADER with lgkmcnt(0) only -- is flagged; the same loop with the wait in front of the barrier is clean; a counted wait in
    front of a raw barrier (a ring deeper than two) is reported with its count and accepted only by name."""
    bad = """
_Z3badv:
\ts_mov_b32 s0, 0
.LBB0_1:
\ts_waitcnt lgkmcnt(0)
\ts_barrier
\tds_read_b128 v[0:3], v4
\tglobal_load_lds_dwordx4 v[8:9], off
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
.Lfunc_end0:
"""
    good = bad.replace("_Z3badv", "_Z4goodv").replace("s_waitcnt lgkmcnt(0)", "s_waitcnt vmcnt(0) lgkmcnt(0)").replace("LBB0", "LBB1").replace("end0", "end1")
    ring = """
_Z4ringv:
.LBB2_1:
\tglobal_load_lds_dwordx4 v[8:9], off
\tglobal_load_lds_dwordx4 v[8:9], off offset:1024
\ts_waitcnt vmcnt(2)
\ts_barrier
\ts_cbranch_scc1 .LBB2_1
\ts_endpgm
.Lfunc_end2:
"""
    p = tmp_path / "k.s"
    p.write_text(bad + good + ring)
    tool = os.path.join(ROOT, "tools", "check_dma_barriers.py")
    r = subprocess.run([sys.executable, tool, str(p)], capture_output=True, text=True)
    out = r.stdout.splitlines()
    assert r.returncode == 1
    assert any(l.startswith("PENDING") and "_Z3badv" in l and "1 of 1 barriers" in l for l in out), out
    assert any(l.startswith("ok") and "_Z4goodv" in l for l in out), out
    assert any(l.startswith("PENDING") and "_Z4ringv" in l and "up to 2 " in l for l in out), out
    r = subprocess.run([sys.executable, tool, "--allow=ringv", str(p)], capture_output=True, text=True)
    assert any(l.startswith("by design") and "_Z4ringv" in l for l in r.stdout.splitlines())
    p.write_text(good + ring)
    assert subprocess.run([sys.executable, tool, "--allow=ringv", str(p)], capture_output=True, text=True).returncode == 0
