#!/usr/bin/env python3
"""Writes profiles/README.md: ONE ROW PER FILE FAMILY (file name without its round prefix, configurations and seeds folded),
with the rounds that hold a member and what the family is.  Re-run after adding files:  python tools/profiles_index.py"""
import os
import re
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WHAT = [  # (regex on the family name, what it is / how it is made)
    (r"^bench_CFG\.json$", "the ONE JSON line of `bench.py` per workload (`--workload c2|c3|c3f64|c4|c5`; round 6 also `_float64` / `_mixed` = `--dtype`, `_seedN` = `--seed N`): value, roofline, roofline_fit, cpu_baseline, fit_ms"),
    (r"^bench_CFG_(float64|mixed)\.json$", "`bench.py --workload c4|c5 --dtype float64|mixed`: the reference's own dtype (and mixed) at the large shares (round 6)"),
    (r"^bench_CFG_seedN\.json$", "`bench.py --seed 1..4` (SURVEY 8d: seeds 0..4; the default line is seed 0)"),
    (r"^bench_grow_dN\.json$", "`bench.py --leaves grow --depth 8|11|12` on C2 / C3 / C4's posterior: SURVEY 8(d) family A, leaves generated on the device; predictions/s by reference rows and by distinct rows"),
    (r"^bench_CFG_(native|bf16x6)\.json$", "the C3 line with the predict math pinned (`--math`)"),
    (r"^bench_CFG_under_rocprof\.json$", "the line `bench.py` printed INSIDE the rocprofv3 run of the same command (its HIP-event kernel time is what `kernel_stats.csv` must agree with)"),
    (r"^bench_CFG_kernel_stats\.csv$", "`rocprofv3 --kernel-trace --stats` summary of `bench.py --steps 50 --warmup 5`"),
    (r"^bench_CFG_rocprof_fullsize\.json$", "`tools/rocprof_fullsize.py`: mean duration of the FULL-SIZE launches of the dominant kernel in that trace"),
    (r"^bench_CFG_(final_head|driver_command)\.json$", "the default line re-measured on the round's final HEAD / with the driver's `--steps 20 --warmup 5`"),
    (r"^pmc_leaf_tiles", "`tools/pmc_traffic_json.py` over three separate `--pmc` passes (FETCH_SIZE+TCC_HIT+GRBM_GUI_ACTIVE | WRITE_SIZE+TCC_MISS+TCC_REQ | SQ_*): HBM bytes per launch (gfx950-corrected), matrix-pipe busy -- the record `bench.py` reads `roofline.traffic` from"),
    (r"^pmc_(summary|sq_leaf_tiles|wait_breakdown)", "raw per-kernel counter means of the PMC passes (`tools/pmc_summary.py`)"),
    (r"^pmc_fit|^fit_CFG_pmc", "PMC passes of the fit's GEMM kernels"),
    (r"^fit_CFG_(posterior|grad)_(timeline\.txt|kernel_stats\.csv)$", "`tools/collect_fit_timeline.sh`: per-kernel time and idle gaps of one float fit (posterior / NLML+gradient) at C3 / C4 / C5"),
    (r"^fit_CFG_f64", "round 6: per-launch timelines of the float64 two-level fit -- round 5's sequential schedule (`_before`) and the overlapped variants (`tools/fit_timeline.py`)"),
    (r"^fit_CFG_(timeline|kernel_stats|kernels_f16planes)", "earlier per-kernel accounts of a fit"),
    (r"^fit_overlap_modes", "`tools/fit_overlap_ab.py`: GPSO_OPT_FIT_OVERLAP = 0 / 1 / 2 / 3 alternating in one process, same bits"),
    (r"^fit_(bench|experiments|analysis)", "what was built, measured and kept or not in the fit kernels of that round (fp16 / bf16 planes, persistent chains, 128-wide steps)"),
    (r"^append_(bench|experiments)|^ab_bits_append", "`gpso_append`: times beside the from-scratch fit (`tools/append_bench.py`), the versions measured on the way incl. round 6's fused-epilogue variant (slower, not kept), bit comparison with the round before"),
    (r"^predict_experiments", "the split predict kernel: every variant with same-box A/B numbers, stamps, ablations; round 6: instruction census, issue-port budget, the stagger"),
    (r"^(ab_|ablation_times)", "raw `tools/ab_time.py` / `tools/ab_bits.py` / `tools/ab_bench.sh` (round 6: through `bench.py` itself) outputs of single changes: two builds alternating on one box"),
    (r"^(step_timeline|explore_overhead|host_overhead|small_call_modes|graph_probe|loop_profile)", "host / launch overhead of a call: device timeline of bench steps, wall vs device of small calls, launch-sequence modes"),
    (r"^loop_(bench|refit_every|large_n)", "caller-level runs (`tools/loop_bench.py`, `tools/loop_large_n.py`): whole GPSO runs against the CPU oracle loop; `refit_every`; round 6: D = 12 with 2 000 initial points"),
    (r"^(sweep|c16_check|pad_bench|gen_probe|gen_choice|split_math_accuracy|grad_error|precision_)", "accuracy / throughput sweeps over N x D, predict math, generation type (`tools/sweep.py`, `c16_check.py`, `split_math_accuracy.py`, ...)"),
    (r"^float_errors", "the float parity errors `pytest -m gpu -s` prints (the bounds in `tests/test_gpu_parity.py` are <= 5x these) and the winner-rule census"),
    (r"^fuzz", "`tools/fuzz_gpu.py` sweeps (seed in the name): randomised GPU-vs-oracle cases; round 6 adds appends at N in [2000, 5000]"),
    (r"^(race_probe|first_launch_hashes|packed_mean)", "run-to-run determinism probes and the record of round 2's packed-FP32 wrong-mean hunt"),
    (r"^(issue_probe|lat_probe|lds_conflicts|diag_phases|small_fit_phases|leaf_phases|leaf_bf16_phases|syrk_bench|lookahead_probe|overlap_probe)", "micro-probes (`tools/micro/*.hip`): issue rates, latencies, LDS conflicts, `s_memtime` phase stamps"),
    (r"^vendor_yardstick", "`tools/vendor_yardstick.py`: the same fits / predictions through torch -> rocSOLVER / rocBLAS on the same box (test-only yardstick)"),
    (r"^(linv_tile_census|screen_census)", "CPU censuses that decided NOT to build something: fragments of L^-1 far below the fp16 scale (round 5), survivors of a screened best-UCB call (round 6)"),
    (r"^hyperopt_notpd_probe", "`tools/hyperopt_notpd_probe.py`: where a float32 hyper-parameter search loses positive definiteness, what the escalation to `mixed` costs, the oracle's fit beside it"),
    (r"^(step32|pmc_16_vs_32)", "round 6: the fused step on the 32x32x16 matrix instruction beside the 16x16x32 one (`tools/step32_ab.py`; PMC: clocks, instruction counts): fewer clocks at a lower clock, 7 % slower, not kept"),
    (r"^(boundary_fit|step_census)", "round 6: what a row-block boundary of the split kernel costs (kernel ms over N at D = 12, least squares); the instruction census of the final kernel's hot loop (`tools/count_step_instructions.py`)"),
    (r"^asan_cpu", "host-side ASan + UBSan run of the CPU suites (`tools/run_asan_cpu.sh`)"),
    (r"^gputest", "a full `pytest -m gpu` log of that round"),
]


def family(name):
    m = re.match(r"^(r\d+[a-z]?)_(.*)$", name)
    if not m:
        return None, None
    rnd, rest = m.groups()
    rest = re.sub(r"seed\d+", "seedN", rest)
    rest = re.sub(r"_(c2|c3f64|c3|c4|c5)(?=[_.])", "_CFG", rest)
    rest = re.sub(r"_d\d+\.", "_dN.", rest)
    return rnd, rest


fams = defaultdict(lambda: defaultdict(int))
for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
    rnd, fam = family(f)
    if fam:
        fams[fam][rnd[:3]] += 1
rows = []
for fam, rounds in sorted(fams.items()):
    what = next((w for rx, w in WHAT if re.search(rx, fam)), "(see the round's section of DESIGN.md / docs/history)")
    rows.append((fam, ", ".join(f"{r} ({n})" if n > 1 else r for r, n in sorted(rounds.items())), what))
out = ["# profiles/", "",
       "rocprofv3 summaries, bench lines and experiment records the numbers in DESIGN.md / bench.py are checked against.",
       "Naming: `rNN[x]_<family>`; `CFG` below stands for a configuration (`c2` ... `c5`), `seedN` / `dN` for a seed / depth.",
       "The current round's bundle (`r06_*`) comes from ONE collector, run on the final code: `bash tools/collect_profiles.sh r06",
       "a|b|c` on the GPU box (a = bench lines + kernel stats + PMC passes + step timeline; b = fit timelines, loop benches,",
       "fuzz, float errors; c = append bench, fit modes, yardstick).  PMC passes are separate runs with `--kernel-trace` only.",
       "", "| family | rounds (files) | what |", "|---|---|---|"]
out += [f"| `{a}` | {b} | {c} |" for a, b, c in rows]
open(os.path.join(ROOT, "profiles", "README.md"), "w").write("\n".join(out) + "\n")
print(len(rows), "families")
