#!/usr/bin/env python3
"""cProfile of one warm GPSO run (D = 2, depth 5, budget 50: the G6 protocol) with the HIP engine: where the host time of
the loop goes.  Run on the GPU box."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.loop_bench import objective  # noqa: E402
from pygpso_amd import GPSOptimiser, ParameterSpace  # noqa: E402

dim, depth, budget = 2, 5, 50
bounds = [[-1.0, 1.0]] * dim


def run():
    space = ParameterSpace(parameter_names=[f"p{i}" for i in range(dim)], parameter_bounds=bounds)
    opt = GPSOptimiser(space, exploration_depth=depth, budget=budget)
    opt.run(objective)
    return opt


run()
t0 = time.perf_counter()
run()
print(f"warm run: {(time.perf_counter() - t0) * 1e3:.1f} ms")
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
