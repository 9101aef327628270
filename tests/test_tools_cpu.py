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
