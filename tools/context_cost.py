import sys, time, numpy as np
sys.path.insert(0, '.')
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_problem
X, y = synthetic_problem(52, 2, seed=0)
e = HipGPEngine("float64"); e.set_data(X, y); e.fit_eval("Matern52", [0.3], 1.0, 1e-3, 0.0); e.close()
for rep in range(3):
    t0 = time.perf_counter(); e = HipGPEngine("float64"); t1 = time.perf_counter()
    e.set_data(X, y); e.fit_eval("Matern52", [0.3], 1.0, 1e-3, 0.0); t2 = time.perf_counter()
    e.best_ucb_grow(np.array([[[0.0, 0.3]] * 2]), 5, 1.8); t3 = time.perf_counter()
    e.fit_eval("Matern52", [0.3], 1.0, 1e-3, 0.0); t4 = time.perf_counter()
    e.close(); t5 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.2f} ms | first set_data+fit {1e3*(t2-t1):.2f} | first grow call {1e3*(t3-t2):.2f} | second fit {1e3*(t4-t3):.3f} | close {1e3*(t5-t4):.2f}")
