#!/usr/bin/env python3
"""
Headline benchmark: leaf-UCB predictions/s (+ GP-fit ms) on synthetic {N_train, D, N_leaves}.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5] [--dtype float32|mixed|float64]
                    [--seed 0..4] [--leaves batch|grow --depth d]

A "step" = one pass of the hot path over one leaf batch: ``gpso_best_ucb`` (prep -> leaf-tile MFMA
kernel -> finalize -> arg-max) on leaves ALREADY RESIDENT in HBM, posterior resident.  N = 1 runs
BASELINE.json's single-GPU performance configuration C3 (D=12, N_train=2048, 64k leaves, fp32).
For N > 1 (launched by ``python -m torch.distributed.run --nproc-per-node N ...``) every rank
gets the same number of leaves (weak scaling): rank 0 fits, the predict-ready posterior is
broadcast over RCCL (ONE ncclBroadcast of a contiguous range; ``posterior_broadcast_ms``) -- and, timed beside it,
every rank repeats the same deterministic fit itself (``posterior_replicate_ms``, fingerprints compared; ``--posterior``
picks which of the two the timed steps use) --, each rank scores its shard, the winners are all-gathered -- all of it
inside the library (gpso_comm_init / gpso_broadcast_posterior / gpso_best_ucb_sharded).

Order of a run: fit (timed by the library's events: ``fit_ms``) -> [broadcast] -> leaves to HBM -> ``--settle-s`` seconds
(default 0.3) of UNTIMED calls of the same step, so that the device holds busy-state clocks (reported as ``settle_s``;
with 20 / 5 steps the kernel otherwise runs at start-up clocks: 0.93 ms against 0.86) -> W untimed warm-up steps ->
barrier + synchronize -> EXACTLY K timed steps -> synchronize + barrier -> CPU baseline and accuracy report (untimed).

Beside the contract's fields the line carries (round 5) ``ms_per_step_pipelined`` / ``value_pipelined`` (the same K steps with two
calls in flight through gpso_best_ucb_begin / _end; the headline ``value`` keeps the synchronous protocol), ``fit_ms`` =
{posterior, nlml_grad, append_k7 (gpso_append of 7 points), hyperopt (one warm-started L-BFGS-B fit: wall / device ms,
evaluations)} with ``roofline_fit``, and for N > 1 ``allgather_us`` (the winners' all-gather + fold per step on rank 0).

Rank 0 prints ONE JSON line.  ``roofline`` is for the dominant kernel (leaf_tiles_kernel):
achieved = algorithmic FLOPs per launch (N^2 + 2ND + 20N per leaf) / its average
duration measured with HIP events on the library's stream inside the timed region.
``cpu_baseline`` times the CPU oracle (numpy/scipy float64, the stand-in for the reference's
GPflow path, which cannot be installed here) on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

# rank 0 prints ONE JSON line on stdout.  RCCL writes its version banner (NCCL_DEBUG=VERSION, as set on the GPU boxes)
# straight to file descriptor 1: under the launcher, descriptor 1 is pointed at stderr for the duration of the run and
# handed back for the one line
_STDOUT_FD = None


def _stdout_to_stderr():
    global _STDOUT_FD
    sys.stdout.flush()
    _STDOUT_FD = os.dup(1)
    os.dup2(2, 1)


def _stdout_back():
    global _STDOUT_FD
    if _STDOUT_FD is not None:
        sys.stdout.flush()
        os.dup2(_STDOUT_FD, 1)
        os.close(_STDOUT_FD)
        _STDOUT_FD = None

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (D, N_train, leaves per GPU, dtype, label)
    "c2": (6, 256, 4096, "float64", "C2: D=6 N_train=256 leaves=4096 fp64 Matern52"),
    "c3": (12, 2048, 65536, "float32", "C3: D=12 N_train=2048 leaves=65536/GPU fp32 Matern52"),
    "c4": (20, 8192, 32768, "float32", "C4 (one GPU's share): D=20 N_train=8192 leaves=32768/GPU fp32 Matern52"),
    "c3f64": (12, 2048, 65536, "float64", "C3 shape in fp64: D=12 N_train=2048 leaves=65536/GPU fp64 Matern52"),
    "c5": (40, 16384, 131072, "float32", "C5 (one GPU's share): D=40 N_train=16384 leaves=131072/GPU fp32 Matern52"),
}
PEAK_TFLOPS = {"float32": 157.3, "float64": 78.6}  # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak; a split product costs 6 (bf16x6) or 3 (f16x3, bf16x3) MFMAs
# HBM bytes per launch of the dominant kernel come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE
# cannot be read from inside the process); the committed record of the latest collection:
# (the newest record that exists wins: the current round's collection first)
PMC_TRAFFIC = {("c3", "native"): ["profiles/r02f_pmc_leaf_tiles_c3.json"],
               ("c3", "bf16x6"): ["profiles/r02h_pmc_leaf_tiles_bf16x6_c3.json"],
               ("c3", "f16x3"): ["profiles/r06_pmc_leaf_tiles_f16x3_c3.json", "profiles/r05_pmc_leaf_tiles_f16x3_c3.json"]}
# what a fit's achieved FLOP rate is priced against: the dense MFMA peak of the arithmetic its LARGE PRODUCTS ran on
# (gpso_last_count(ctx, 2)); a split product costs 6 (bf16 pieces) or 3 (fp16 pieces) 16-bit MFMAs
FIT_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6, "small": 78.6, "bf16x6": PEAK_BF16_TFLOPS / 6, "f16x3": PEAK_BF16_TFLOPS / 3}


def pmc_traffic(workload, math_mode):
    for rel in PMC_TRAFFIC.get((workload, math_mode), []):
        path = os.path.join(ROOT, rel)
        if os.path.isfile(path):
            break
    else:
        return None, None
    with open(path) as fh:
        rec = json.load(fh)
    return rec.get("traffic_bytes_per_launch"), os.path.relpath(path, ROOT)


def synthetic(n, d, m, seed=0):
    from tests.helpers import synthetic_leaves, synthetic_problem

    X, y = synthetic_problem(n, d, seed=seed)
    return X, y, synthetic_leaves(m, d, seed=seed + 1)


def cpu_baseline(X, y, theta, leaves, varsigma, budget_s=12.0):
    """Time the CPU oracle on a bounded sample of the workload (rank 0, N = 1 only)."""
    from oracle import gpr

    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    th = gpr.Theta(*theta)
    gpr.posterior(th, X, y)  # warm-up: BLAS thread pool start, page faults
    fits = []
    for _ in range(3):
        t0 = time.perf_counter()
        post = gpr.posterior(th, X, y)
        fits.append(time.perf_counter() - t0)
    fit_s = float(np.median(fits))
    # The oracle's predict pass is numpy elementwise work (the kernel map) around two BLAS calls: one Python thread uses ONE
    # core for most of it.  The stated baseline runs leaf chunks (SURVEY 8d processes leaves in chunks) on a thread pool (numpy and BLAS
    # release the GIL) with BLAS held to one thread per worker, so that every host core works; the per-chunk winners are
    # folded with np.argmax's rule.
    from concurrent.futures import ThreadPoolExecutor

    workers = max(1, min(os.cpu_count() or 1, 64))
    chunk = 1024  # (small chunks: 64 of them in a 65 536-leaf batch, so that every worker has one)

    def best_of(lo):
        i, mu, var, ucb = gpr.best_ucb(post, leaves[lo:lo + chunk], varsigma)
        return ucb, -(lo + i)

    def one_pass(sample_rows):
        starts = list(range(0, sample_rows, chunk))
        with ThreadPoolExecutor(min(workers, len(starts))) as pool:
            return max(pool.map(best_of, starts))

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # noqa: BLE001
        threadpool_limits = None
    t0 = time.perf_counter()
    gpr.best_ucb(post, leaves[:2048], varsigma)
    per_2k = time.perf_counter() - t0
    est_rate = 2048 / max(per_2k, 1e-6) * min(workers, 8)  # (a guess to size the sample; measured below)
    n_chunks = int(max(1, min(leaves.shape[0] // chunk, budget_s * est_rate / chunk / 3)))
    sample = leaves[: n_chunks * chunk]
    rates, spent = [], 0.0
    while spent < budget_s or len(rates) < 2:
        t0 = time.perf_counter()
        if threadpool_limits is not None:
            with threadpool_limits(limits=1):
                one_pass(sample.shape[0])
        else:
            one_pass(sample.shape[0])
        dt = time.perf_counter() - t0
        rates.append(sample.shape[0] / dt)
        spent += dt
        if len(rates) >= 8:
            break
    threads = max(threads, min(workers, n_chunks))
    return {
        "value": float(np.median(rates)), "unit": "predictions/s", "cores": int(threads), "kind": "port",
        "sample": f"first {sample.shape[0]} of the {leaves.shape[0]} leaves, float64 numpy/scipy oracle "
                  f"(LAPACK potrf/trsm), median of {len(rates)} passes, {spent:.1f} s of CPU work",
        "fit_ms_posterior": fit_s * 1e3,
        "gflops": float(np.median(rates)) * (X.shape[0] ** 2 + 2 * X.shape[0] * X.shape[1] + 20 * X.shape[0]) / 1e9,
        "fit_gflops": (X.shape[0] ** 3 / 3 + X.shape[0] ** 2 * X.shape[1] + 10 * X.shape[0] ** 2) / fit_s / 1e9,
        "predict_threads": int(min(workers, n_chunks)),
        "note": "the numpy/scipy restatement of the reference's algorithm, not a tuned CPU implementation: the predict pass "
                "runs 1024-leaf chunks on a thread pool (one BLAS thread per worker; numpy's elementwise kernel map is "
                "single-threaded per chunk), the fit (LAPACK potrf / trsm) uses the BLAS's own threads; fit time is the "
                "median of 3 warm calls",
    }, post


def roofline_fit(n, d, dtype, fit_ms, fit_math):
    """GP-fit half of the metric against the MFMA peak of THE ARITHMETIC THAT RAN (``fit_math`` = HipGPEngine.fit_math() of
    the timed evaluations: single-level float fit -> f32 157.3; float64 / mixed -> f64 78.6; two-level float fit on fp16
    pieces -> 2500 / 3 = 833.3, on bf16 pieces -> 2500 / 6 = 416.7 TFLOP/s f32-equivalent).  Algorithmic FLOPs (SURVEY.md
    8d): posterior fit N^3/3 + N^2 D + 10 N^2; one NLML + gradient evaluation adds 2 N^3 / 3 (K^-1 from L) + 2 H N^2 (H = 4
    hyper-parameters).  Times are the library's HIP-event medians (gpso_last_ms(ctx, 2)) of 5 evaluations each."""
    if not fit_ms:
        return None
    fit_math = fit_math or ("f64" if dtype != "float32" else "f32")
    peak = FIT_PEAK_TFLOPS[fit_math]
    f_post = n ** 3 / 3 + n * n * d + 10 * n * n
    f_grad = f_post + 2 * n ** 3 / 3 + 2 * 4 * n * n
    two_level = fit_math in ("f16x3", "bf16x6") or (fit_math == "f64" and -(-n // 128) * 128 > 2560)
    out = {"bound": "mfma", "peak": peak, "unit": "TFLOP/s", "fit_math": fit_math,
           "dominant_kernel": ("small_fit_kernel (one launch)" if fit_math == "small" else
                               "gemm_bf16_kernel (16-bit-piece rank-1024 updates / level-doubling inverse / K^-1) + potrf_step_kernel "
                               "chain of the diagonal blocks, looked ahead on a side stream" if fit_math in ("f16x3", "bf16x6") else
                               "gemm128_kernel (SYRK / TRSM / level-doubling inverse) + potrf_step_kernel chain" if two_level else
                               "potrf_step_kernel (one launch per 64 columns: diagonal-block chain + trailing update)"),
           "peak_note": {"f32": "dense f32 MFMA peak", "f64": "dense f64 MFMA peak", "small": "dense f64 MFMA peak (a one-launch, latency-bound fit)",
                         "f16x3": "f32-equivalent FLOPs; every large product is 3 fp16 MFMAs with f32 accumulation: bound = 2500 / 3",
                         "bf16x6": "f32-equivalent FLOPs; every large product is 6 bf16 MFMAs with f32 accumulation: bound = 2500 / 6"}[fit_math],
           "flops_posterior": f_post, "flops_nlml_grad": f_grad}
    for key, fl in (("posterior", f_post), ("nlml_grad", f_grad)):
        if key in fit_ms:
            ach = fl / (fit_ms[key] * 1e-3) / 1e12
            out[key] = {"ms": fit_ms[key], "achieved": ach, "frac": ach / peak}
    if fit_ms.get("append_k7") == fit_ms.get("append_k7") and "append_k7" in fit_ms:  # (present and not NaN)
        # gpso_append is HBM-bound: two passes over the lower triangle of L^-1 (N^2 s bytes) + O(N k) vectors
        sz = 4 if dtype == "float32" else 8
        by = (n - 7) ** 2 * sz
        gbs = by / (fit_ms["append_k7"] * 1e-3) / 1e9
        out["append_k7"] = {"ms": fit_ms["append_k7"], "bound": "hbm", "algorithmic_bytes": by, "achieved": gbs,
                            "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                            "speedup_over_posterior_fit": fit_ms["posterior"] / fit_ms["append_k7"] if "posterior" in fit_ms else None}
    return out


def hyperopt_fit(dtype, device, X, y, theta):
    """What gpso/gp_surrogate.py:500-503 does per GP update: ONE L-BFGS-B maximum-likelihood fit of the hyper-parameters,
    warm-started near the optimum as the reference's re-used model is -- from (l, s2, noise, c) = (1.3 l*, 1.5, 3e-3, 0)
    -- through the drop-in model class on an engine of the workload's dtype that the MODEL owns (hyper-opt wall
    time with iteration count).  A float32 search that loses positive definiteness reopens the engine as "mixed" (float64
    fit) and starts over (pygpso_amd/kernels.py: Scipy.minimize): the line then says so -- ``dtype`` = the engine the
    search finished on, ``escalations``, and the wall time INCLUDES the abandoned float32 evaluations."""
    from pygpso_amd.kernels import Constant, Matern52, Scipy
    from pygpso_amd.model import HipGPR

    model = HipGPR(data=(X, y[:, None]), kernel=Matern52(lengthscales=1.3 * theta[1], variance=1.5),
                   mean_function=Constant(0.0), noise_variance=3.0e-3, dtype=dtype, device=device)
    dev_ms, timed_engine = [], [None]
    inner = model._loss_and_grad

    def timed(u):
        if model.engine is not timed_engine[0]:  # (the first call, and after an escalation)
            model.engine.set_timing(1)
            timed_engine[0] = model.engine
        out = inner(u)
        dev_ms.append(model.engine.last_ms(2))
        return out

    model._loss_and_grad = timed
    t0 = time.perf_counter()
    res = Scipy().minimize(model.training_loss, model.trainable_variables)
    wall = (time.perf_counter() - t0) * 1e3
    out = {"wall_ms": wall, "device_ms": float(np.sum(dev_ms)), "evaluations": int(res.nfev), "iterations": int(res.nit),
           "evaluations_in_all": int(model.num_loss_evals), "dtype": model.engine.dtype_name, "requested_dtype": dtype,
           "escalations": int(model.fit_escalations), "fit_math": model.engine.fit_math(),
           "nlml": float(res.fun), "theta": {k: np.asarray(v).tolist() for k, v in model.parameter_dict().items()},
           "start": "l = 1.3 x 0.25 sqrt(D), s2 = 1.5, noise 3e-3, c = 0; SciPy L-BFGS-B defaults (as gpflow.optimizers.Scipy)"}
    model.engine.close()
    return out


def split_bf16_report(eng, leaves_dev, leaves_all, varsigma, m_total, flops_per_leaf, post, steps):
    """Time gpso_best_ucb with the L^-1 apply on the bf16 matrix cores (3 / 6 bf16 MFMAs per f32
    product) and measure the error of each mode -- and of native f32 -- against the float64 oracle."""
    from oracle import gpr

    rep = {"note": "gpso_set_option(GPSO_OPT_PREDICT_MATH): every predict math on the same leaves in the same run; "
                   "the headline value above is the default (GPSO_MATH_AUTO)"}
    sample = leaves_all[:2048]
    ref = gpr.predict_y(post, sample) if post is not None else None
    time.sleep(0.5)  # let the BLAS worker threads of the CPU-baseline leg stop spinning
    for mode in ("native", "bf16x6", "f16x3", "f16x3+f32-contraction", "bf16x3"):
        # (f16x3 contracts x.x* on the fp16 pipe by default; "+f32-contraction" is rounds 1-3's f32 matrix instruction)
        eng.set_predict_math(mode.split("+")[0])
        eng.set_contraction("f32" if "+" in mode else "auto")
        for _ in range(3):
            eng.best_ucb(leaves_dev, varsigma)
        t0 = time.perf_counter()
        ks = []
        for _ in range(steps):
            eng.best_ucb(leaves_dev, varsigma)
            ks.append(eng.last_ms(0))
        dt = time.perf_counter() - t0
        tfl = flops_per_leaf * m_total / (np.mean(ks) * 1e-3) / 1e12
        bound = PEAK_TFLOPS["float32"] if mode == "native" else PEAK_BF16_TFLOPS / int(mode.split("+")[0][-1])
        entry = {"value": m_total * steps / dt, "unit": "predictions/s", "kernel_ms": float(np.mean(ks)),
                 "algorithmic_tflops": tfl, "bound_tflops": bound, "frac_of_bound": tfl / bound}
        if ref is not None:
            mean, var = eng.predict(sample.astype(np.float32))
            entry["max_abs_err_mean_vs_f64_oracle"] = float(np.max(np.abs(mean - ref[0])))
            entry["max_abs_err_var_vs_f64_oracle"] = float(np.max(np.abs(var - ref[1])))
        rep[mode] = entry
    eng.set_contraction("auto")
    eng.set_predict_math("auto")
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--noise", type=float, default=1.0e-3, help="noise variance of the synthetic posterior")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hyperopt", action="store_true", help="skip fit_ms.hyperopt (one L-BFGS-B fit: seconds at C5 in float64)")
    ap.add_argument("--timing-every", type=int, default=4,
                    help="the leaf-tile kernel is timed with HIP events on every k-th timed step (1 = every step)")
    ap.add_argument("--settle-s", type=float, default=0.3,
                    help="seconds of untimed hot-path calls before the warm-up steps (device clocks settle; 0 = none)")
    ap.add_argument("--posterior", default="broadcast", choices=["broadcast", "replicate"],
                    help="--gpus N > 1: how the peers get the posterior the timed steps use -- ONE ncclBroadcast of the "
                         "fitting rank's predict-ready range, or the same deterministic fit on every rank (fingerprints "
                         "compared).  Both are timed and reported whichever is chosen (SURVEY 8e: measure both)")
    ap.add_argument("--dtype", default=None, choices=["float32", "mixed", "float64"],
                    help="arithmetic of the engine (default: the workload's).  float64 is the reference's own (gpflow.default_float, "
                         "gpso/gp_surrogate.py:490-495) and the drop-in surrogate's default; mixed = float64 fit + float predict")
    ap.add_argument("--seed", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="seed of the synthetic problem (SURVEY 8d: seeds 0..4; leaves use seed + 1); stated in the line")
    ap.add_argument("--leaves", default="batch", choices=["batch", "grow"],
                    help="batch: SURVEY 8(d) family B -- the first M rows of a seeded U[0,1]^(M x D), resident in HBM.  grow: family A "
                         "-- the call the optimiser really makes (gpso/optimisation.py:366-382 on gpso/param_space.py:175-200): "
                         "LeafNode.grow(--depth) of the two outer children of a box at tree depth 3, generated ON the device by "
                         "gpso_best_ucb_grow; value counts the REFERENCE's rows, value_distinct_rows the rows actually scored")
    ap.add_argument("--depth", type=int, default=11, help="--leaves grow: depth of the grown sub-trees (8, 11, 12: (3^d - 1) / 2 rows per box)")
    ap.add_argument("--math", default="auto", choices=["auto", "native", "bf16x3", "bf16x6", "f16x3"],
                    help="predict math of float32 workloads (auto = the library default: the first rung of the ladder "
                         "f16x3 -> bf16x6 -> native f32 the posterior's self-test passes with)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from pygpso_amd import HipGPEngine
    from pygpso_amd import distributed as D
    from scipy.special import erfcinv

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                             "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    # launched by torch.distributed.run (RANK set): always take the distributed code path, also for
    # --nproc-per-node 1, so the RCCL plumbing is exercised identically at every N
    use_dist = "RANK" in os.environ
    if use_dist:
        _stdout_to_stderr()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    d, n, m_per_gpu, dtype, label = WORKLOADS[args.workload]
    if args.dtype is not None and args.dtype != dtype:
        dtype = args.dtype
        label += f" -- run as --dtype {dtype}"
    float_predict = dtype in ("float32", "mixed")
    grow = args.leaves == "grow"
    boxes = None
    if grow:
        # SURVEY 8(d) family A: a box at tree depth 3 (the unit box split three times along its widest dimension -- first
        # maximum --, left child each time) and its two OUTER children, bounds lo + i (w / 3) as gpso/param_space.py:257-307
        lo_b, hi_b = np.zeros(d), np.ones(d)
        for k in range(3):
            hi_b[k] = lo_b[k] + 1 * ((hi_b[k] - lo_b[k]) / 3)
        w3 = (hi_b[3 % d] - lo_b[3 % d]) / 3
        kids = []
        for i in (0, 2):
            b = np.stack([lo_b, hi_b], axis=1)
            b[3 % d] = (lo_b[3 % d] + i * w3, lo_b[3 % d] + (i + 1) * w3)
            kids.append(b)
        boxes = np.ascontiguousarray(np.stack(kids))
        rows_ref = 2 * (3 ** args.depth - 1) // 2
        rows_distinct = 2 * 3 ** (args.depth - 1)
        m_per_gpu = rows_ref  # (the value counts the reference's rows; under --gpus N the rows of the two boxes are SHARDED)
        label += f" -- leaves: grow(depth {args.depth}) of the two outer children of a depth-3 box = {rows_ref} reference rows, {rows_distinct} distinct"
    m_total = m_per_gpu * (1 if grow else world)
    varsigma = float(erfcinv(0.01))
    X, y, leaves_all = synthetic(n, d, 1 if grow else m_total, seed=args.seed)
    theta = ("Matern52", 0.25 * math.sqrt(d), 1.0, args.noise, float(y.mean()))

    math_opt = args.math if float_predict else "native"
    eng = HipGPEngine(dtype, device=local_rank, predict_math=math_opt)
    # ---- fit on rank 0 (timed separately), broadcast the predict-ready posterior ----------------
    fit_ms = {}
    fit_math = [None]

    def fit_here(timed):
        eng.set_data(X, y)
        if timed:
            for name, want_grad in (("posterior", False), ("nlml_grad", True)):
                ts = []
                for _ in range(5):
                    eng.fit_eval(*theta, want_grad=want_grad)
                    ts.append(eng.last_ms(2))
                fit_ms[name] = float(np.median(ts))
            # gpso_append: the posterior of the first N - 7 points extended by the last 7 at the same hyper-parameters
            # (the optimiser's iterations add 1-7 points; two passes over L^-1 instead of a refit)
            ts = []
            for _ in range(5):
                eng.set_data(X[:-7], y[:-7])
                eng.fit_eval(*theta, want_grad=False)
                _, in_place = eng.append(X[-7:], y[-7:])
                ts.append(eng.last_ms(2) if in_place else float("nan"))
            fit_ms["append_k7"] = float(np.median(ts))
            fit_math[0] = eng.fit_math()
            if not args.no_hyperopt:
                # (a float32 factorisation can lose positive definiteness where L-BFGS-B's line search steps into tiny noise --
                # N = 16 384 in float32 does; the reference's float64 path does not: the model then reopens its engine as
                # "mixed" and the search starts over -- hyperopt.dtype says where it finished)
                fit_ms["hyperopt"] = hyperopt_fit(dtype, local_rank, X, y, theta)
            eng.set_data(X, y)
        eng.fit_eval(*theta, want_grad=False)

    math_mode = "native"
    if rank == 0:
        fit_here(timed=True)
        if float_predict:  # what GPSO_MATH_AUTO settled on for this posterior (its self-test ran here)
            math_mode = eng.precision_info()["predict_math"]
    bcast_ms = repl_ms = None
    posterior_bytes = None
    hashes_agree = resident_agree = None
    distribution = "single GPU"
    if use_dist:
        # the library's own RCCL group (C-ABI): rank 0 creates the id, the launcher's process group only
        # carries the 128 bytes; torch.distributed is otherwise used for the barrier and the MAX of the
        # timings the bench contract asks for -- no tensor of the hot path goes through it
        box = [D.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        eng.comm_init(rank, world, box[0])

        def by_broadcast():
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            D.broadcast_posterior(eng, src=0)  # gpso_broadcast_posterior: ONE ncclBroadcast, device to device, synchronous
            dist.barrier()
            return (time.perf_counter() - t0) * 1e3

        def by_replication():
            # every rank runs the same bit-deterministic fit on its own copy of (X, y): no bulk collective
            if rank != 0:
                fit_here(timed=False)  # untimed: code-object load and first-launch effects (rank 0 has fitted before)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            fit_here(timed=False)
            eng.synchronize()
            dist.barrier()
            return (time.perf_counter() - t0) * 1e3

        def fingerprints():
            hs = [None] * world
            dist.all_gather_object(hs, eng.posterior_hash())
            return hs

        # both ways, the chosen one last (its posterior is the one the timed steps use; the bits are the same)
        if args.posterior == "broadcast":
            repl_ms = by_replication()
            hs = fingerprints()
            bcast_ms = by_broadcast()
        else:
            bcast_ms = by_broadcast()
            repl_ms = by_replication()
            hs = fingerprints()
        hashes_agree = len(set(hs)) == 1
        if not hashes_agree:
            # reported in the line, never fatal for the measurement: the timed steps then run on ONE posterior, rank 0's
            print(f"[bench] replicated fits disagree across ranks: {[f'{h:016x}' for h in hs]} -- broadcasting rank 0's",
                  file=sys.stderr, flush=True)
            if args.posterior == "replicate":
                by_broadcast()
                args.posterior = "broadcast"
        resident_agree = len(set(fingerprints())) == 1
        if not resident_agree:
            print("[bench] posterior fingerprints differ across ranks at the start of the timed steps", file=sys.stderr, flush=True)
        posterior_bytes = int(eng.posterior_span()[2])
        distribution = ("fit on rank 0, ONE RCCL broadcast of the contiguous predict-ready range of its posterior arena "
                        "(gpso_broadcast_posterior)" if args.posterior == "broadcast" else
                        "the same deterministic fit on every rank, fingerprints compared (gpso_posterior_hash); no bulk collective")

    # ---- this rank's leaf shard, resident in HBM before the timed region --------------------------
    lo, hi = (0, m_total) if grow else D.shard_range(m_total, rank, world)
    np_dtype = np.float32 if float_predict else np.float64
    leaves_dev = None if grow else torch.from_numpy(np.ascontiguousarray(leaves_all[lo:hi].astype(np_dtype))).cuda(local_rank)

    def step():
        if grow:
            # gpso_best_ucb_grow: the centres are generated on the device (bit-identical to LeafNode.grow), every distinct one
            # scored once; the global winner of the two boxes is folded on the host from the two records
            if use_dist:
                idx, mean, var, ucb = D.best_ucb_grow_sharded(eng, boxes, args.depth, varsigma)
            else:
                idx, mean, var, ucb = eng.best_ucb_grow(boxes, args.depth, varsigma)
            j = int(np.argmax(ucb))
            return int(idx[j]) + j * (rows_ref // 2), float(mean[j]), float(var[j]), float(ucb[j])
        if use_dist:  # gpso_best_ucb_sharded: local scoring, RCCL all-gather of the winners, fold on device
            idx, mean, var, ucb = D.best_ucb_sharded(eng, leaves_dev, m_total, varsigma)
        else:
            idx, mean, var, ucb = eng.best_ucb(leaves_dev, varsigma)
        return int(idx[0]), float(mean[0]), float(var[0]), float(ucb[0])

    # clock settle (untimed, BEFORE the W warm-up steps the contract asks for): the fit, the uploads and the process
    # start-up leave the GPU at whatever clock the power management had reached; the same call is repeated for
    # --settle-s seconds so that warm-up and timed steps run at the clocks a busy device holds.  Reported as "settle_s".
    # (the NUMBER of settle steps must be the same on every rank -- each step is a collective call under --gpus N: it is
    # derived from one measured step, maximum over the ranks)
    if args.settle_s > 0:
        t_one = time.perf_counter()
        winner = step()
        t_one = time.perf_counter() - t_one
        if use_dist:
            t = torch.tensor([t_one], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_one = float(t.item())
        for _ in range(min(2000, int(math.ceil(args.settle_s / max(t_one, 1e-6))))):
            winner = step()
    for _ in range(args.warmup):
        winner = step()
    # the dominant kernel's duration is measured INSIDE the timed region with HIP events on the library's stream -- on
    # every TIMING_EVERY-th step: the event pairs and the elapsed-time queries cost 10-25 us of a ~0.85 ms step, so a
    # sample of the steps is instrumented (GPSO_OPT_TIMING = k), the others run as a caller without a stopwatch runs them
    every = max(1, min(args.timing_every, args.steps))
    eng.set_timing(every)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    tile_ms, coll_ms = [], []
    t0 = time.perf_counter()
    for i in range(args.steps):
        winner = step()
        if i % every == 0:
            tile_ms.append(eng.last_ms(0))
            if use_dist:
                coll_ms.append(eng.last_ms(3))  # ncclAllGather of the winners + the fold kernel, HIP events on the library's stream
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng.set_timing(1)
    # the same K steps through the non-blocking pair, two deep (gpso_best_ucb_begin / _end): the device work of step i + 1
    # is queued before the host reads the result of step i, so the GPU does not idle over the host's round trip.  Reported
    # BESIDE the headline (which keeps the synchronous protocol above); single-GPU runs only (the sharded call is a
    # collective and stays blocking).
    pipelined_ms = pipelined_same = None
    if not use_dist and not grow:
        for _ in range(3):
            eng.best_ucb_end(eng.best_ucb_begin(leaves_dev, varsigma))
        torch.cuda.synchronize()
        t0p = time.perf_counter()
        tk = eng.best_ucb_begin(leaves_dev, varsigma)
        for _ in range(args.steps - 1):
            tk2 = eng.best_ucb_begin(leaves_dev, varsigma)
            pw = eng.best_ucb_end(tk)
            tk = tk2
        pw = eng.best_ucb_end(tk)
        torch.cuda.synchronize()
        pipelined_ms = (time.perf_counter() - t0p) / args.steps * 1e3
        pipelined_same = bool(int(pw[0][0]) == winner[0] and float(pw[3][0]) == winner[3])  # (reported, never fatal)

    if rank == 0:
        kern_ms = float(np.mean(tile_ms))
        flops_per_leaf = n * n + 2 * n * d + 20 * n
        # rows the leaf-tile kernel scores per launch: this rank's shard; under --leaves grow the DISTINCT rows (the library
        # drops the centre children that repeat their parent bit for bit: gpso_last_count(ctx, 0))
        scored = (eng.last_count(0) if grow else hi - lo)
        achieved = flops_per_leaf * scored / (kern_ms * 1e-3) / 1e12
        traffic, traffic_src = (None, None) if grow or args.dtype is not None else pmc_traffic(args.workload, math_mode)
        peak = PEAK_TFLOPS["float32" if float_predict else "float64"] if math_mode == "native" else PEAK_BF16_TFLOPS / int(math_mode[-1])
        out = {
            "metric": "leaf_ucb_predictions_per_sec",
            "value": m_total * args.steps / elapsed,
            "unit": "predictions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_pipelined": pipelined_ms,  # two calls in flight (gpso_best_ucb_begin / _end); the headline is synchronous
            "value_pipelined": (m_total / (pipelined_ms * 1e-3)) if pipelined_ms else None,
            "pipelined_same_winner_bits": pipelined_same,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if float_predict else "f64",  # the arithmetic class of the predict path ("mixed": float64 fit + float predict)
            "engine_dtype": dtype,
            "data": "synthetic",
            "seed": args.seed,
            "config": {
                "workload": label if world == 1 else
                f"{label} -- weak scaling: the same {m_per_gpu} leaves on each of the {world} GPUs ({m_total} in all); "
                + ("this IS BASELINE.json configs[3] (C4: 262144 leaves over 8 GPUs)" if (args.workload, world) == ("c4", 8) else
                   "this IS BASELINE.json configs[4] (C5: 1M leaves over 8 GPUs)" if (args.workload, world) == ("c5", 8) else
                   "BASELINE.json's 8-GPU configs are --workload c4 / c5 with --gpus 8 (the default keeps C3's per-GPU work "
                   "at every N so that the driver's scaling curve compares like with like)"),
                "D": d, "N_train": n, "leaves_per_gpu": m_per_gpu,
                "leaves_total": m_total, "kernel": "Matern52", "lengthscale": theta[1],
                "noise_variance": theta[3], "parallelism": f"leaf-shard x{world}",
                "leaves_resident_in_hbm": not grow, "predict_math": math_mode, "predict_math_option": math_opt,
                "seed": args.seed, "engine_dtype": dtype,
                **({"leaves": "grow", "depth": args.depth, "reference_rows": rows_ref, "distinct_rows": rows_distinct,
                    "boxes": boxes.tolist()} if grow else {"leaves": "batch"}),
            },
            # --leaves grow: ``value`` counts the rows of the reference's list (what LeafNode.grow would have handed to
            # gp_eval_best_ucb); the kernels score every DISTINCT centre once
            **({"value_distinct_rows": scored * args.steps / elapsed * (world if use_dist else 1),
                "rows_scored_per_call": scored} if grow else {}),
            "fit_ms": fit_ms,
            "roofline_fit": roofline_fit(n, d, dtype, fit_ms, fit_math[0]),
            "posterior_broadcast_ms": bcast_ms,
            "posterior_replicate_ms": repl_ms,
            "posterior_fingerprints_agree": hashes_agree,          # replicated fits: one fingerprint on every rank
            "posterior_resident_agree": resident_agree,            # ... and the posterior the timed steps run on
            "posterior_mode": args.posterior if use_dist else None,
            "posterior_bytes": posterior_bytes,
            "rccl_world": world if use_dist else None,
            # the per-step collective of the sharded call on rank 0 (sampled like kernel_ms): a flat weak-scaling curve can be
            # told from a hidden collective cost by this number
            "allgather_us": (float(np.mean(coll_ms)) * 1e3 if coll_ms else None),
            "posterior_distribution": distribution,
            "settle_s": args.settle_s,
            "winner": {"index": winner[0], "ucb": winner[3]},
            "roofline": {
                "kernel": "leaf_tiles_kernel" if math_mode == "native" else f"leaf_tiles_bf16_kernel({math_mode})",
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "peak_note": "dense f32 / f64 MFMA peak" if math_mode == "native" else
                f"algorithmic (f32-equivalent) FLOPs; the kernel forms every f32 product from {math_mode[-1]} {'fp16' if math_mode[0] == 'f' else 'bf16'} MFMAs "
                f"with f32 accumulation, so the bound is the dense bf16 / fp16 peak / {math_mode[-1]} = {peak:.0f} TFLOP/s",
                "traffic": traffic, "traffic_unit": "bytes/launch (HBM, PMC, gfx950-corrected)",
                "traffic_source": traffic_src,
                "algorithmic_bytes": int(scored * (d + 3) * (4 if float_predict else 8)
                                         + (n * n // 2 + n * d + n) * (4 if float_predict else 8)),
                "kernel_ms": kern_ms, "kernel_ms_samples": len(tile_ms), "kernel_timed_every": every,
                # the same algorithmic flops against round 2's bound (six bf16 MFMAs per product): the fp16 split does
                # half the matrix work for an f32-class product, so the figures of the two rounds compare on this line
                **({"achieved_over_bf16x6_bound": achieved / (PEAK_BF16_TFLOPS / 6)} if math_mode == "f16x3" else {}),
                "flops_per_leaf": flops_per_leaf, "leaves_per_launch": scored,
            },
        }
        post = None
        if grow and world == 1 and not args.no_cpu_baseline:
            # the checker for the grown leaves: the oracle grows the first box on the host (small depths only) and scores it
            from oracle import gpr, tree

            if args.depth <= 9:
                post = gpr.posterior(gpr.Theta(*theta), X, y)
                ucbs = []
                for b in boxes:
                    rows = tree.grow([tuple(r) for r in b], args.depth)
                    mr, vr = gpr.predict_y(post, rows)
                    ucbs.append(mr + varsigma * vr)
                ucb_ref = np.concatenate(ucbs)
                out["winner"]["oracle_argmax"] = int(np.argmax(ucb_ref))
                out["winner"]["oracle_ucb_gap"] = float(ucb_ref.max() - ucb_ref[winner[0]])
        elif world == 1 and not args.no_cpu_baseline:
            cb, post = cpu_baseline(X, y, theta, leaves_all, varsigma)
            out["cpu_baseline"] = cb
            # the checker: the GPU winner must be (within fp tolerance) the oracle's winner on the sample
            from oracle import gpr

            n_s = int(cb["sample"].split()[1])  # "first <n> of the ..."
            mean_ref, var_ref = gpr.predict_y(post, leaves_all[: max(n_s, 1)])
            ucb_ref = mean_ref + varsigma * var_ref
            if winner[0] < n_s:
                # the oracle's own arg-max over the sample and how far (in oracle UCB) the GPU's winner is from it
                out["winner"]["oracle_ucb_at_index"] = float(ucb_ref[winner[0]])
                out["winner"]["oracle_argmax_over_sample"] = int(np.argmax(ucb_ref))
                out["winner"]["oracle_ucb_gap"] = float(ucb_ref.max() - ucb_ref[winner[0]])
                out["winner"]["same_argmax_as_oracle"] = bool(int(np.argmax(ucb_ref)) == winner[0]) if n_s == leaves_all.shape[0] else None
        if world == 1 and float_predict and math_opt == "auto" and not grow:
            # every predict math on the same leaves in the same run: throughput + accuracy vs the oracle
            out["predict_math_modes"] = split_bf16_report(eng, leaves_dev, leaves_all, varsigma, m_total,
                                                  flops_per_leaf, post, max(3, args.steps // 2))
        _stdout_back()
        print(json.dumps(out), flush=True)
        if use_dist:
            _stdout_to_stderr()  # (library chatter at tear-down)
    if use_dist:
        eng.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
