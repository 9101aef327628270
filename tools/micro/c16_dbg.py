import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from pygpso_amd import HipGPEngine
from tests.helpers import synthetic_leaves, synthetic_problem
def run(n, d, label, kw, after=None, contr=None):
    X, y = synthetic_problem(n, d, seed=0)
    eng = HipGPEngine("float32", **kw)
    if contr: eng.set_contraction(contr)
    eng.set_data(X, y)
    eng.fit_eval("Matern52", np.array([0.6]), 1.0, 1e-3, float(y.mean()), want_grad=False)
    if after: eng.set_predict_math(after)
    try:
        mt, vt = eng.predict(X[:64])
        print(n, d, label, "train err %.3g" % np.abs(mt - y[:64]).max(), vt[:2], flush=True)
    except Exception as e:
        print(n, d, label, "EXC", str(e)[:90], flush=True)
    eng.close()
for n, d in ((256, 6), (512, 6), (2048, 12)):
    run(n, d, "pinned nocheck", dict(predict_math="f16x3", generation="float32", precision_check=False))
    run(n, d, "auto nocheck", dict(generation="float32", precision_check=False))
    run(n, d, "auto check", dict(generation="float32"))
    run(n, d, "pinned-after nocheck", dict(generation="float32", precision_check=False), after="f16x3")
    run(n, d, "pinned nocheck f32contr", dict(predict_math="f16x3", generation="float32", precision_check=False), contr="f32")
    run(n, d, "pinned bf16x6 nocheck", dict(predict_math="bf16x6", generation="float32", precision_check=False))
