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
        self.sync_n = -1  # rows the peers of a group hold of this posterior (mirror of api.hip: sync_n); -1: unknown

    def set_data(self, X, y):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64).reshape(-1)
        self.n, self.d = self.X.shape
        self.post = None
        self.sync_n = -1

    def fit_eval(self, kernel, lengthscales, variance, noise, mean_c, want_grad=True):
        self.sync_n = -1
        th = gpr.Theta(kernel, lengthscales, variance, noise, mean_c)
        if want_grad:
            f, g = gpr.nlml_and_grad(th, self.X, self.y)
        else:
            f, g = gpr.posterior(th, self.X, self.y).nlml, None
        self.post = gpr.posterior(th, self.X, self.y)
        return f, g

    def append(self, Xnew, ynew):
        """What ``HipGPEngine.append`` answers: the posterior of the old + new points at the resident hyper-parameters
        (here simply refitted by the oracle) -> (nlml, in_place)."""
        th = self.post.theta
        keep = self.sync_n  # (an in-place append leaves the record of what the peers hold alone)
        self.set_data(np.vstack([self.X, np.atleast_2d(Xnew)]), np.concatenate([self.y, np.asarray(ynew).reshape(-1)]))
        self.post = gpr.posterior(th, self.X, self.y)
        self.sync_n = keep
        return self.post.nlml, True

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

    # -- posterior hand-off of pygpso_amd.distributed.HostGroup (host memory) ------------------------
    def export_posterior(self):
        th = self.post.theta
        return dict(kernel=th.kernel, lengthscales=np.array(th.lengthscales), variance=th.variance,
                    noise=th.noise, mean_c=th.mean_c, X=self.post.X.copy(), L=self.post.L.copy(),
                    alpha=self.post.alpha.copy(), y=self.post.y.copy())

    # -- the rows an append wrote (host mirror of gpso_posterior_dirty_ranges / gpso_broadcast_posterior_rows) -----------
    def export_rows(self, n_base):
        """What a peer holding the first ``n_base`` rows lacks: the new rows of X, y and L (the old rows of a Cholesky
        factor do not change when rows are appended), all of alpha."""
        p = self.post
        return dict(n_base=int(n_base), X=p.X[n_base:].copy(), y=p.y[n_base:].copy(), L_rows=p.L[n_base:].copy(),
                    alpha=p.alpha.copy())

    def import_rows(self, st):
        p, nb = self.post, st["n_base"]
        assert p.X.shape[0] == nb
        n1 = nb + st["X"].shape[0]
        L = np.zeros((n1, n1))
        L[:nb, :nb] = p.L
        L[nb:, :] = st["L_rows"]
        p.X, p.y, p.L, p.alpha = np.vstack([p.X, st["X"]]), np.concatenate([p.y, st["y"]]), L, st["alpha"]
        self.n = n1

    def import_posterior(self, st):
        post = gpr.Posterior()
        post.theta = gpr.Theta(st["kernel"], st["lengthscales"], st["variance"], st["noise"], st["mean_c"])
        post.X, post.L, post.alpha, post.y, post.nlml = st["X"], st["L"], st["alpha"], st["y"], float("nan")
        self.post = post
        self.n, self.d = post.X.shape
