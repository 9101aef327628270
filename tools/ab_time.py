#!/usr/bin/env python3
"""A/B timing of builds of libgpso_hip.so inside ONE GPU call (devices differ by up to ~10 % in what a compute-bound
kernel does, so a number from another box says nothing about a change): every library named on the command line is run
in a fresh process, alternating A B A B ..., and the medians are printed side by side.

    python tools/ab_time.py [--what predict|fit|small] [--rounds 3] [--shape N D M] pygpso_amd/libgpso_hip_base.so pygpso_amd/libgpso_hip.so

--what predict: leaf-tile kernel ms (library events) and wall ms per gpso_best_ucb call, per predict math
--what fit:     posterior / NLML+gradient ms (library events) at the shape
--what small:   wall us per fit_eval_u + best_ucb_grow at N = 52, D = 2 (the optimiser loop's calls)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(args):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch  # noqa: F401  (device buffers for resident leaves)

    from pygpso_amd import HipGPEngine
    from tests.helpers import synthetic_leaves, synthetic_problem

    n, d, m = args.shape
    out = {}
    X, y = synthetic_problem(n, d, seed=0)
    theta = ("Matern52", 0.25 * np.sqrt(d) * np.ones(1), 1.0, 1e-3, float(y.mean()))
    if args.what == "predict":
        leaves = torch.from_numpy(synthetic_leaves(m, d).astype(np.float32)).cuda()
        for math in args.math.split(","):
            eng = HipGPEngine("float32", predict_math=math)
            if os.environ.get("AB_ROW_LOOP"):  # "<lib>:rl0" on the command line: GPSO_OPT_ROW_LOOP = 0 (one row block per workgroup)
                eng.set_row_loop(int(os.environ["AB_ROW_LOOP"]))
            eng.set_data(X, y)
            eng.fit_eval(*theta, want_grad=False)
            t_end = time.perf_counter() + 0.3
            while time.perf_counter() < t_end:
                eng.best_ucb(leaves, 2.0)
            ks = []
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eng.best_ucb(leaves, 2.0)
                ks.append(eng.last_ms(0))
            wall = (time.perf_counter() - t0) / args.steps * 1e3
            out[math] = {"kernel_ms": float(np.median(ks)), "step_ms": wall}
            eng.set_timing(False)  # the same call without the library's event pairs (gpso_last_ms reads 0 then)
            for _ in range(20):
                eng.best_ucb(leaves, 2.0)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eng.best_ucb(leaves, 2.0)
            out[math]["step_ms_no_events"] = (time.perf_counter() - t0) / args.steps * 1e3
            eng.close()
    elif args.what == "fit":
        for dtype in args.dtype.split(","):
            eng = HipGPEngine(dtype)
            eng.set_data(X, y)
            for name, g in (("posterior", False), ("nlml_grad", True)):
                ts = []
                for _ in range(args.steps):
                    eng.fit_eval(*theta, want_grad=g)
                    ts.append(eng.last_ms(2))
                out[f"{dtype}.{name}"] = {"ms": float(np.median(ts[2:])), "min": float(np.min(ts[2:]))}
            eng.close()
    else:  # small: the optimiser loop's calls
        from oracle import tree  # (geometry helper only)

        X, y = synthetic_problem(52, 2, seed=0)
        eng = HipGPEngine("float64")
        eng.set_timing(False)
        eng.set_data(X, y)
        u = np.array([0.3, 0.5, -6.0, 0.1])
        kids = tree.split_bounds([(0.0, 1.0)] * 2)
        boxes = np.array([kids[0], kids[2]])
        for _ in range(50):
            eng.fit_eval_u("Matern52", u, 1, True)
            eng.best_ucb_grow(boxes, 5, 1.82)
        t0 = time.perf_counter()
        for _ in range(args.steps * 10):
            eng.fit_eval_u("Matern52", u, 1, True)
        t1 = time.perf_counter()
        for _ in range(args.steps * 10):
            eng.best_ucb_grow(boxes, 5, 1.82)
        t2 = time.perf_counter()
        out["fit_eval_u_us"] = (t1 - t0) / (args.steps * 10) * 1e6
        out["best_ucb_grow_us"] = (t2 - t1) / (args.steps * 10) * 1e6
    print("AB_RESULT " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--what", default="predict", choices=["predict", "fit", "small"])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--shape", type=int, nargs=3, default=[2048, 12, 65536])
    ap.add_argument("--math", default="f16x3")
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--worker", action="store_true")
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    res = {lib: [] for lib in args.libs}
    for r in range(args.rounds):
        for lib in args.libs:
            path, _, opt = lib.partition(":rl")
            env = dict(os.environ, GPSO_HIP_LIB=os.path.abspath(path), GPSO_HIP_LIB_OLDER="1")
            if opt:
                env["AB_ROW_LOOP"] = opt
            cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--what", args.what, "--steps", str(args.steps),
                   "--shape", *map(str, args.shape), "--math", args.math, "--dtype", args.dtype]
            p = subprocess.run(cmd, env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("AB_RESULT ")]
            if not line:
                print(lib, "FAILED", p.stderr[-800:], flush=True)
                continue
            res[lib].append(json.loads(line[0][len("AB_RESULT "):]))
            print(f"round {r} {os.path.basename(lib)}: {line[0][len('AB_RESULT '):]}", flush=True)
    import statistics

    print("---- medians over rounds")
    for lib, runs in res.items():
        if not runs:
            continue
        keys = runs[0].keys()
        summary = {}
        for k in keys:
            v0 = runs[0][k]
            if isinstance(v0, dict):
                summary[k] = {kk: round(statistics.median(r[k][kk] for r in runs), 4) for kk in v0}
            else:
                summary[k] = round(statistics.median(r[k] for r in runs), 3)
        print(os.path.basename(lib), json.dumps(summary))


if __name__ == "__main__":
    main()
