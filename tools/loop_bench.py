#!/usr/bin/env python3
"""
Caller-level benchmark: a whole GPSO run (update -> explore ->
select loop) with the HIP engine behind the drop-in classes, against the CPU oracle restatement of
the reference loop (numpy/scipy GP + Python/numpy ternary growth, as the reference does it).

    python tools/loop_bench.py [--dim 4] [--depth 7] [--budget 80]

Prints one JSON line: wall seconds of both, per-phase shares of the GPU run, and whether the two
runs agree (best point, evaluation counts) -- they must: float64 on both sides.
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def objective(p):
    p = np.asarray(p)
    return float(np.exp(-np.sum((p - 0.3) ** 2) * 4.0) + 0.5 * np.exp(-np.sum((p + 0.4) ** 2) * 6.0)
                 + 0.05 * np.sin(3.0 * p.sum()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=4)
    ap.add_argument("--depth", type=int, default=7)
    ap.add_argument("--budget", type=int, default=80)
    ap.add_argument("--refit-every", type=int, default=1,
                    help="c > 1 (opt-in, not the reference's behaviour): the surrogate re-optimises its hyper-parameters on every "
                         "c-th update only and extends the device posterior by the new points in between (gpso_append)")
    ap.add_argument("--skip-cpu", action="store_true", help="do not run the CPU oracle loop beside it")
    a = ap.parse_args()
    bounds = [[-1.0, 1.0]] * a.dim

    from oracle import gpso_loop
    from pygpso_amd import GPRSurrogate, GPSOptimiser, ParameterSpace

    def surrogate():
        surr = GPRSurrogate.default()
        surr.refit_every = max(1, a.refit_every)
        return surr

    # first pass: includes the one-off costs (library + code-object load, context creation, allocations)
    t0 = time.perf_counter()
    space = ParameterSpace(parameter_names=[f"p{i}" for i in range(a.dim)], parameter_bounds=bounds)
    GPSOptimiser(space, gp_surrogate=surrogate(), exploration_depth=a.depth, budget=a.budget).run(objective)
    t_cold = time.perf_counter() - t0

    t0 = time.perf_counter()
    space = ParameterSpace(parameter_names=[f"p{i}" for i in range(a.dim)], parameter_bounds=bounds)
    opt = GPSOptimiser(space, gp_surrogate=surrogate(), exploration_depth=a.depth, budget=a.budget)
    phases = {"_gp_update": 0.0, "_tree_explore": 0.0, "_tree_select": 0.0}
    for name in phases:
        fn = getattr(opt, name)

        def timed(*args, _fn=fn, _name=name, **kw):
            t = time.perf_counter()
            try:
                return _fn(*args, **kw)
            finally:
                phases[_name] += time.perf_counter() - t

        setattr(opt, name, timed)
    best = opt.run(objective)
    t_gpu = time.perf_counter() - t0
    model = opt.gp_surr.gpflow_model

    if a.skip_cpu or a.refit_every > 1:  # (the oracle loop restates the reference: it always re-optimises)
        print(json.dumps({"dim": a.dim, "depth": a.depth, "budget": a.budget, "refit_every": a.refit_every,
                          "evaluations": opt.n_eval_counter, "iterations": opt.iterations,
                          "gp_loss_evaluations": model.num_loss_evals, "hip_seconds": t_gpu, "hip_phase_seconds": phases,
                          "best_score_hip": best.score_mu, "training_points": int(model.data[0].shape[0])}))
        return
    t0 = time.perf_counter()
    st = gpso_loop.LoopState(bounds, depth=a.depth, budget=a.budget)
    best_ref = gpso_loop.run(st, objective)
    t_cpu = time.perf_counter() - t0

    print(json.dumps({
        "dim": a.dim, "depth": a.depth, "budget": a.budget,
        "leaves_per_child": (3 ** a.depth - 1) // 2,
        "evaluations": opt.n_eval_counter, "iterations": opt.iterations,
        "gp_loss_evaluations": model.num_loss_evals,
        "leaf_predictions_cpu_run": st.n_leaf_predictions,
        "hip_seconds": t_gpu, "hip_seconds_first_run_in_process": t_cold, "hip_phase_seconds": phases,
        "cpu_oracle_seconds": t_cpu, "speedup": t_cpu / t_gpu,
        "same_evaluation_counts": [t[0] for t in opt.trace] == [t[0] for t in st.trace],
        "best_score_hip": best.score_mu, "best_score_cpu": best_ref["mu"],
        "best_coord_max_abs_diff": float(np.max(np.abs(best.normed_coord - best_ref["coord"]))),
    }))


if __name__ == "__main__":
    main()
