"""
TEST DOUBLE: an object with the ``HipGPEngine`` interface whose arithmetic is the CPU oracle.

Lives under tests/ on purpose -- it lets the CPU-only suite exercise the HOST logic of the product
(point store, tree, explore/select/update loop, save/resume, leaf sharding) without a GPU.  It is
never importable from ``pygpso_amd`` and is not a fallback: the product raises without the HIP
extension / a device.
"""
import numpy as np

from oracle import gpr, tree


class OracleEngine:
    def __init__(self, dtype="float64", device=0):
        self.n = self.d = 0
        self.post = None
        self._ms = 0.0

    def set_data(self, X, y):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64).reshape(-1)
        self.n, self.d = self.X.shape
        self.post = None

    def fit_eval(self, kernel, lengthscales, variance, noise, mean_c, want_grad=True):
        th = gpr.Theta(kernel, lengthscales, variance, noise, mean_c)
        if want_grad:
            f, g = gpr.nlml_and_grad(th, self.X, self.y)
        else:
            f, g = gpr.posterior(th, self.X, self.y).nlml, None
        self.post = gpr.posterior(th, self.X, self.y)
        return f, g

    def predict(self, xs, out=None):
        return gpr.predict_y(self.post, np.asarray(xs, dtype=np.float64))

    def best_ucb(self, xs, varsigma, seg_off=None):
        xs = np.asarray(xs, dtype=np.float64)
        so = np.array([0, xs.shape[0]]) if seg_off is None else np.asarray(seg_off)
        mean, var = gpr.predict_y(self.post, xs) if xs.shape[0] else (np.empty(0), np.empty(0))
        ucb = mean + varsigma * var
        idx, mu, vv, uu = [], [], [], []
        for a, b in zip(so[:-1], so[1:]):
            if b <= a:
                idx.append(-1); mu.append(np.nan); vv.append(np.nan); uu.append(np.nan)
                continue
            i = int(np.argmax(ucb[a:b]))
            idx.append(i); mu.append(mean[a + i]); vv.append(var[a + i]); uu.append(ucb[a + i])
        return np.array(idx), np.array(mu), np.array(vv), np.array(uu)

    def grow_rows(self, depth):
        return tree.grow_count(depth)

    def grow(self, bounds, depth):
        b = np.asarray(bounds, dtype=np.float64)
        if b.ndim == 2:
            return tree.grow([tuple(r) for r in b], depth)
        return np.stack([tree.grow([tuple(r) for r in bb], depth) for bb in b])

    def best_ucb_grow(self, bounds, depth, varsigma):
        b = np.asarray(bounds, dtype=np.float64)
        if b.ndim == 2:
            b = b[None]
        rows = tree.grow_count(depth)
        coords = np.vstack([tree.grow([tuple(r) for r in bb], depth) for bb in b])
        return self.best_ucb(coords, varsigma, np.arange(b.shape[0] + 1) * rows)

    def last_ms(self, what=0):
        return self._ms

    # -- the broadcast protocol of pygpso_amd.distributed, over host memory -----------------------
    _HYPER = 8 + 48

    def alloc_posterior(self, n, d):
        self.n, self.d = int(n), int(d)
        self._state = [np.zeros(self._HYPER), np.zeros((self.n, self.d)), np.zeros((self.n, self.n)),
                       np.zeros(self.n), np.zeros(self.n)]

    def posterior_tensors(self):
        import torch

        if self.post is not None:
            th = self.post.theta
            h = np.zeros(self._HYPER)
            h[:7] = [self.n, self.d, gpr.KERNELS.index(th.kernel), th.lengthscales.shape[0],
                     th.variance, th.noise, th.mean_c]
            h[8:8 + th.lengthscales.shape[0]] = th.lengthscales
            self._state = [h, self.post.X.copy(), self.post.L.copy(), self.post.alpha.copy(),
                           self.post.y.copy()]
        return [torch.from_numpy(a.reshape(-1)) for a in self._state]

    def adopt_posterior(self):
        h, X, Lc, alpha, y = self._state
        n_ls = int(h[3])
        post = gpr.Posterior()
        post.theta = gpr.Theta(gpr.KERNELS[int(h[2])], h[8:8 + n_ls], h[4], h[5], h[6])
        post.X, post.L, post.alpha, post.y, post.nlml = X, Lc, alpha, y, float("nan")
        self.post = post
