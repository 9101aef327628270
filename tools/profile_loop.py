import cProfile, pstats, sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
from tools.loop_bench import objective
from pygpso_amd import GPSOptimiser, ParameterSpace
dim, depth, budget = 4, 7, 80
def run():
    space = ParameterSpace(parameter_names=[f"p{i}" for i in range(dim)], parameter_bounds=[[-1.0, 1.0]] * dim)
    opt = GPSOptimiser(space, exploration_depth=depth, budget=budget)
    return opt.run(objective)
run()  # warm
pr = cProfile.Profile(); pr.enable(); t = time.perf_counter(); run(); el = time.perf_counter() - t; pr.disable()
print("run seconds", el)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
