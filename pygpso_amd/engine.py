"""
``HipGPEngine``: one GP posterior resident on one MI355X, driven through the C-ABI.

It plays the role the ``gpflow.models.GPR`` object plays for the reference
(``gpso/gp_surrogate.py:488-503``): it owns the training data, the hyper-parameters and the
factorisation, evaluates the training loss (+ gradient) and answers ``predict_y``.  All arithmetic
happens in the hand-written HIP kernels of ``libgpso_hip.so``; this file only marshals pointers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _is_device_tensor(x):
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


def _producer_stream(x):
    """hipStream_t (int) on which the framework that owns the device tensor ``x`` is currently
    queueing work for x's device.  Only torch tensors are recognised -- through the module the CALLER
    has already imported; this package never imports torch itself.  None: unknown (the C-ABI then
    relies on its documented contract: device buffers must be ready before the call)."""
    import sys

    torch = sys.modules.get("torch")
    if torch is None or not isinstance(x, torch.Tensor):
        return None
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # (the pointer without a Stream object: this sits in every call)
    if raw is not None and x.device.index is not None:
        return int(raw(x.device.index))
    return int(torch.cuda.current_stream(x.device).cuda_stream)


class HipGPEngine:
    def __init__(self, dtype="float64", device=0, predict_math=None, generation=None,
                 precision_check=None, tol_var=None, tol_mean=None):
        """``dtype``: "float64" (fit and predict in double: the reference's arithmetic), "float32" (fit
        and predict apply in float) or "mixed" (fit in double, predict apply in float: the
        hyper-parameter path is bit-identical to float64).
        ``predict_math`` (float-predict engines only): "auto" (default: "bf16x6" where the posterior's
        shape allows it and its self-test passes with it, else "native"), "native" f32 MFMA, or the
        split-bf16 modes "bf16x6" (six bf16 MFMAs per f32 product, f32 accumulation: f32-class
        accuracy) / "bf16x3" (|d var| ~ 2e-5 sigma^2) on the bf16 matrix cores.
        ``generation`` (float-predict engines): "auto" (default: float when the posterior's self-test
        passes with it, else double), "float64" (r^2 of the cross-Gram tile formed in double) or
        "float32" (GPflow's GEMM form in float: faster, |d r^2| ~ 1e-5).
        ``precision_check`` / ``tol_var`` / ``tol_mean``: the self-test that guards float predictions
        (include/gpso_hip.h: gpso_precision_info); on by default for float-predict engines."""
        self._lib = L.load()
        if dtype is np.float64:
            dtype = "float64"
        elif dtype is np.float32:
            dtype = "float32"
        self.dtype_name = {L.F64: "float64", L.F32: "float32", L.MIXED: "mixed"}[L.DTYPE_IDS[dtype]]
        self.dtype = L.DTYPE_IDS[dtype]
        self.device = int(device)
        handle = C.c_void_p()
        rc = self._lib.gpso_create(C.byref(handle), self.device, self.dtype)
        if rc != L.OK:
            raise L.GpsoHipError(rc, self._lib.gpso_last_error(None).decode())
        self._h = handle
        self.n = 0
        self.d = 0
        self.rank, self.world = 0, 1
        self._u_bufs = {}
        self._win_bufs = {}
        if predict_math is not None and not (predict_math in ("native", "f32") and self.dtype == L.F64):
            self.set_predict_math(predict_math)
        if generation is not None:
            self.set_generation(generation)
        if precision_check is not None:
            self.set_precision_check(precision_check)
        if tol_var is not None or tol_mean is not None:
            self.set_tolerances(tol_var, tol_mean)

    # -- plumbing ----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpso_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            msg = self._lib.gpso_last_error(self._h).decode()
            if rc == L.E_NOTPD:
                raise np.linalg.LinAlgError(msg)
            if rc == L.E_ARG:
                raise ValueError(msg)
            if rc == L.E_PRECISION:
                raise L.GpsoPrecisionError(rc, msg)
            raise L.GpsoHipError(rc, msg)
        return rc

    def set_predict_math(self, mode):
        self._check(self._lib.gpso_set_option(self._h, L.OPT_PREDICT_MATH, L.MATH_IDS[mode]))

    def set_generation(self, mode):
        """"auto" (default) | "float64" | "float32": arithmetic of the cross-Gram x.x* contraction and r^2."""
        self._check(self._lib.gpso_set_option(self._h, L.OPT_GENERATION, L.GEN_IDS[mode]))

    def set_timing(self, on):
        """GPSO_OPT_TIMING: record the event pairs ``last_ms`` reads (default on); off saves two to four HIP calls per
        entry point -- what a loop of small evaluations wants."""
        self._check(self._lib.gpso_set_option(self._h, L.OPT_TIMING, int(on)))  # (True / 1: every call, k: every k-th)

    def set_split_kernel(self, which):
        """GPSO_OPT_SPLIT_KERNEL: "auto" (the fused step) | "two-phase" (round 3's step): same bits, different speed."""
        self._check(self._lib.gpso_set_option(self._h, L.OPT_SPLIT_KERNEL, {"auto": 0, "two-phase": 1, "fused16": 2, "fused32": 3}[which]))

    def set_contraction(self, which):
        """GPSO_OPT_CONTRACTION: "auto" / "f16" (the x.x* contraction of the fp16-split kernel on the fp16 pipe under
        float generation) | "f32" (the f32 matrix instruction, as in rounds 1-3)."""
        self._check(self._lib.gpso_set_option(self._h, L.OPT_CONTRACTION, {"auto": 0, "f32": 1, "f16": 2}[which]))

    def set_small_calls(self, on):
        """GPSO_OPT_SMALL_CALLS: the short launch sequences for best-UCB calls on small batches (default on; same bits).
        True / 1: three launches, or one where that measures faster; 2: three only; 3: one wherever it applies; 0: general."""
        self._check(self._lib.gpso_set_option(self._h, L.OPT_SMALL_CALLS, int(on)))

    def set_row_loop(self, on):
        """GPSO_OPT_ROW_LOOP (process-wide): 1 / True = a workgroup of the split predict kernels keeps its leaf tile and loops
        over row blocks, the launcher choosing how many workgroups share a tile (default); 0 = one row block per workgroup
        (rounds 1-5); v >= 2 = exactly min(v, row blocks) workgroups per leaf tile.  Same bits."""
        self._check(self._lib.gpso_set_option(self._h, L.OPT_ROW_LOOP, int(on)))

    def set_precision_check(self, on):
        self._check(self._lib.gpso_set_option(self._h, L.OPT_PRECISION_CHECK, 1 if on else 0))

    def set_tolerances(self, tol_var=None, tol_mean=None):
        """Self-test tolerances: |d var| <= tol_var * sigma^2, |d mean| <= tol_mean * max|y - c|."""
        if tol_var is not None:
            self._check(self._lib.gpso_set_option_f64(self._h, L.OPTF_TOL_VAR, float(tol_var)))
        if tol_mean is not None:
            self._check(self._lib.gpso_set_option_f64(self._h, L.OPTF_TOL_MEAN, float(tol_mean)))

    def precision_info(self, raise_on_fail=False):
        """Self-test of the resident posterior (runs it if needed): measured errors of the predict path
        at the training inputs against their closed form, and the tolerances they are held to."""
        out = np.zeros(12, dtype=np.float64)
        rc = self._lib.gpso_precision_info(self._h, L.dptr(out))
        if rc not in (L.OK, L.E_PRECISION) or (rc == L.E_PRECISION and raise_on_fail):
            self._check(rc)
        keys = ("max_abs_err_mean", "max_abs_err_var", "max_abs_y_minus_c", "min_var", "max_abs_alpha",
                "kernel_variance", "tol_mean_abs", "tol_var_abs", "amplification", "max_kinv_diag")
        info = dict(zip(keys, (float(v) for v in out)))
        info["generation"] = "float32" if out[10] else "float64"
        info["predict_math"] = {L.MATH_NATIVE: "native", L.MATH_BF16X3: "bf16x3", L.MATH_BF16X6: "bf16x6", L.MATH_F16X3: "f16x3"}[int(out[11])]
        info["passed"] = rc == L.OK
        return info

    def wait_stream(self, stream_ptr):
        """Order this engine's stream behind the work queued on another hipStream_t (int pointer;
        0 / None = the legacy default stream)."""
        self._check(self._lib.gpso_wait_stream(self._h, C.c_void_p(stream_ptr or None)))

    def set_fit_single_level_max(self, npad_max):
        """Tuning / test hook (GPSO_OPT_FIT_SINGLE_LEVEL_MAX): 0 forces the two-level Cholesky path."""
        self._check(self._lib.gpso_set_option(self._h, L.OPT_FIT_SINGLE_LEVEL_MAX, int(npad_max)))

    def set_stream(self, stream_ptr):
        """Run on an existing hipStream_t (int pointer, e.g. torch.cuda.Stream().cuda_stream)."""
        self._check(self._lib.gpso_set_stream(self._h, C.c_void_p(stream_ptr or None)))

    def synchronize(self):
        self._check(self._lib.gpso_synchronize(self._h))

    def last_ms(self, what=0):
        return float(self._lib.gpso_last_ms(self._h, int(what)))

    def last_count(self, what=0):
        """0: leaves the kernels scored in the last predict-type call, 1: leaves of the reference's list."""
        return int(self._lib.gpso_last_count(self._h, int(what)))

    FIT_MATH = {L.FITMATH_NONE: None, L.FITMATH_SMALL: "small", L.FITMATH_F32: "f32", L.FITMATH_F64: "f64",
                L.FITMATH_BF16X6: "bf16x6", L.FITMATH_F16X3: "f16x3"}

    def fit_math(self):
        """Arithmetic the large products of the last ``fit_eval`` ran on (``gpso_last_count(ctx, 2)``, GPSO_FITMATH_*):
        "small" (N <= 128, one launch), "f32" / "f64" (the matrix instruction of the fit type), "bf16x6" / "f16x3" (the
        two-level float fit's 16-bit pieces)."""
        return self.FIT_MATH.get(self.last_count(2))

    @property
    def padded_n(self):
        return int(self._lib.gpso_padded_n(self._h))

    # -- fit ---------------------------------------------------------------------------------
    def set_data(self, X, y):
        X = L.as_f64(X)
        if X.ndim != 2:
            raise ValueError("X must be [N, D]")
        y = L.as_f64(np.asarray(y).reshape(-1), (X.shape[0],))
        self._check(self._lib.gpso_set_data(self._h, L.dptr(X), L.dptr(y), X.shape[0], X.shape[1]))
        self.n, self.d = X.shape

    @staticmethod
    def _theta_args(kernel, lengthscales, variance, noise, mean_c):
        kid = L.KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
        ls = L.as_f64(np.atleast_1d(lengthscales))
        return kid, ls, int(ls.shape[0]), float(variance), float(noise), float(mean_c)

    def fit_eval(self, kernel, lengthscales, variance, noise, mean_c, want_grad=True):
        """One NLML (+ gradient w.r.t. the constrained hyper-parameters) evaluation; leaves the
        posterior resident.  Returns (nlml, grad or None); grad order (ls..., variance, noise, c)."""
        kid, ls, n_ls, var, nz, mc = self._theta_args(kernel, lengthscales, variance, noise, mean_c)
        nlml = C.c_double()
        grad = np.empty(n_ls + 3, dtype=np.float64) if want_grad else None
        self._check(self._lib.gpso_fit_eval(self._h, kid, L.dptr(ls), n_ls, var, nz, mc,
                                            C.byref(nlml), L.dptr(grad) if want_grad else None))
        return nlml.value, grad

    def fit_eval_u(self, kernel, u, n_ls, train_mean, mean_c_fixed=0.0):
        """One loss evaluation in the optimiser's unconstrained variables (``gpso_fit_eval_u``: transforms and chain
        rule inside the library).  Returns (nlml, grad_u, theta): theta = constrained (ls..., variance, noise, mean)."""
        kid = L.KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
        n_u = int(n_ls) + 2 + (1 if train_mean else 0)
        key = (int(n_ls), bool(train_mean))  # (n_u alone is ambiguous: theta_out has n_ls + 3 entries whatever train_mean)
        buf = self._u_bufs.get(key)
        if buf is None:  # marshalling buffers are made once per problem shape: this call is the optimiser's inner loop
            ua = np.empty(n_u, dtype=np.float64)
            ga = np.empty(n_u, dtype=np.float64)
            ta = np.empty(int(n_ls) + 3, dtype=np.float64)
            nl = C.c_double()
            buf = self._u_bufs[key] = (ua, ga, ta, nl, L.dptr(ua), L.dptr(ga), L.dptr(ta), C.byref(nl))
        ua, ga, ta, nl, up, gp, tp, nlp = buf
        ua[:] = u
        rc = self._lib.gpso_fit_eval_u(self._h, kid, up, int(n_ls), 1 if train_mean else 0, float(mean_c_fixed), nlp, gp, tp)
        if rc < 0:
            self._check(rc)
        return nl.value, ga.copy(), ta.copy()

    def append(self, Xnew, ynew):
        """``gpso_append``: k new training points extend the resident posterior at its hyper-parameters (two passes over
        L^-1 instead of a factorisation).  Returns (nlml of the N + k points, in_place): ``in_place`` False when the
        library refitted from scratch instead (pad crossing, k > 64, N + k <= 128: ``last_message()`` says which)."""
        Xn = L.as_f64(np.atleast_2d(Xnew))
        if Xn.ndim != 2 or Xn.shape[1] != self.d:
            raise ValueError(f"Xnew must be [k, {self.d}]")
        yn = L.as_f64(np.asarray(ynew).reshape(-1), (Xn.shape[0],))
        nlml = C.c_double()
        try:
            rc = self._check(self._lib.gpso_append(self._h, L.dptr(Xn), L.dptr(yn), Xn.shape[0], C.byref(nlml)))
        finally:
            # whatever happened, N is what the library says it holds (a failed append leaves the first N points, in place
            # or refitted: include/gpso_hip.h)
            n, d = C.c_int64(), C.c_int()
            if self._lib.gpso_problem_shape(self._h, C.byref(n), C.byref(d)) == L.OK:
                self.n = int(n.value)
        return nlml.value, rc == L.OK

    def last_message(self):
        return self._lib.gpso_last_error(self._h).decode()

    def set_posterior(self, X, Lchol, alpha, kernel, lengthscales, variance, noise, mean_c):
        X = L.as_f64(X)
        n, d = X.shape
        Lc = L.as_f64(Lchol, (n, n))
        al = L.as_f64(np.asarray(alpha).reshape(-1), (n,))
        kid, ls, n_ls, var, nz, mc = self._theta_args(kernel, lengthscales, variance, noise, mean_c)
        self._check(self._lib.gpso_set_posterior(self._h, L.dptr(X), L.dptr(Lc), L.dptr(al), n, d,
                                                 kid, L.dptr(ls), n_ls, var, nz, mc))
        self.n, self.d = n, d

    # -- predict -----------------------------------------------------------------------------
    def _winner_bufs(self, nseg):
        """Output arrays of a best-UCB call and their ctypes pointers, made once per segment count: these calls are the
        optimiser's inner loop (the arrays are copied out, so callers may keep what they get)."""
        bufs = self._win_bufs.get(nseg)
        if bufs is None:
            idx = np.empty(nseg, dtype=np.int64)
            mean = np.empty(nseg, dtype=np.float64)
            var = np.empty(nseg, dtype=np.float64)
            ucb = np.empty(nseg, dtype=np.float64)
            bufs = self._win_bufs[nseg] = (idx, mean, var, ucb, L.i64ptr(idx), L.dptr(mean), L.dptr(var), L.dptr(ucb))
        return bufs

    def _leaf_args(self, xs):
        """-> (pointer, xs_dtype, xs_mem, M, keepalive)"""
        if _is_device_tensor(xs):
            if xs.dim() != 2 or not xs.is_contiguous():
                raise ValueError("device leaves must be a contiguous [M, D] tensor")
            name = str(xs.dtype)
            if name.endswith("float64"):
                dt = L.F64
            elif name.endswith("float32"):
                dt = L.F32
            else:
                raise ValueError(f"unsupported leaf dtype {xs.dtype}")
            if self.d and xs.shape[1] != self.d:
                raise ValueError(f"leaves have D={xs.shape[1]}, model has D={self.d}")
            self._order_after(xs)
            return C.c_void_p(xs.data_ptr()), dt, L.MEM_DEVICE, int(xs.shape[0]), xs
        a = np.asarray(xs)
        if a.ndim != 2:
            raise ValueError("leaves must be [M, D]")
        if self.d and a.shape[0] and a.shape[1] != self.d:
            raise ValueError(f"leaves have D={a.shape[1]}, model has D={self.d}")
        if a.dtype == np.float32:
            a = np.ascontiguousarray(a)
            dt = L.F32
        else:
            a = np.ascontiguousarray(a, dtype=np.float64)
            dt = L.F64
        return C.c_void_p(a.ctypes.data), dt, L.MEM_HOST, int(a.shape[0]), a

    def _order_after(self, tensor):
        """The engine runs on its own stream: queue it behind whatever the tensor's framework has
        pending on its current stream (reads of leaves still being written / writes into outputs still
        being read would otherwise race)."""
        ps = _producer_stream(tensor)
        if ps is not None:
            self.wait_stream(ps)

    def predict(self, xs, out=None):
        """predict_y: (mean[M], var[M]) float64; var includes the noise variance.  ``out`` may be a
        pair of float64 CUDA tensors of length M to keep the results on the device."""
        ptr, dt, mem, m, keep = self._leaf_args(xs)
        if out is not None:
            mean_t, var_t = out
            self._order_after(mean_t)
            self._order_after(var_t)
            self._check(self._lib.gpso_predict(self._h, ptr, dt, mem, m, C.c_void_p(mean_t.data_ptr()),
                                               C.c_void_p(var_t.data_ptr()), L.MEM_DEVICE))
            return mean_t, var_t
        mean = np.empty(m, dtype=np.float64)
        var = np.empty(m, dtype=np.float64)
        self._check(self._lib.gpso_predict(self._h, ptr, dt, mem, m, C.c_void_p(mean.ctypes.data),
                                           C.c_void_p(var.ctypes.data), L.MEM_HOST))
        return mean, var

    def best_ucb(self, xs, varsigma, seg_off=None):
        """gp_eval_best_ucb per segment -> (idx, mean, var, ucb) arrays of length nseg."""
        ptr, dt, mem, m, keep = self._leaf_args(xs)
        if seg_off is None:
            nseg, so_ptr = 1, None
        else:
            so = np.ascontiguousarray(seg_off, dtype=np.int64)
            nseg, so_ptr = int(so.shape[0] - 1), L.i64ptr(so)
        idx, mean, var, ucb, pi, pm, pv, pu = self._winner_bufs(nseg)
        rc = self._lib.gpso_best_ucb(self._h, ptr, dt, mem, m, so_ptr, nseg, float(varsigma), pi, pm, pv, pu)
        if rc < 0:
            self._check(rc)
        return idx.copy(), mean.copy(), var.copy(), ucb.copy()

    def best_ucb_begin(self, xs, varsigma, seg_off=None):
        """Non-blocking ``best_ucb``: enqueues the call and returns a ticket for ``best_ucb_end``; two calls may be in
        flight (``gpso_best_ucb_begin``).  The leaves must stay untouched until the call is ended."""
        ptr, dt, mem, m, keep = self._leaf_args(xs)
        if seg_off is None:
            nseg, so_ptr = 1, None
        else:
            so = np.ascontiguousarray(seg_off, dtype=np.int64)
            nseg, so_ptr = int(so.shape[0] - 1), L.i64ptr(so)
        ticket = self._check(self._lib.gpso_best_ucb_begin(self._h, ptr, dt, mem, m, so_ptr, nseg, float(varsigma)))
        self._open_tickets = getattr(self, "_open_tickets", {})
        self._open_tickets[ticket] = (nseg, keep)
        return ticket

    def best_ucb_grow_begin(self, bounds, depth, varsigma):
        b = L.as_f64(bounds)
        if b.ndim == 2:
            b = b[None]
        nseg, d, _ = b.shape
        if d != self.d:
            raise ValueError(f"bounds have D={d}, model has D={self.d}")
        ticket = self._check(self._lib.gpso_best_ucb_grow_begin(self._h, L.dptr(b), nseg, int(depth), float(varsigma)))
        self._open_tickets = getattr(self, "_open_tickets", {})
        self._open_tickets[ticket] = (nseg, None)
        return ticket

    def best_ucb_end(self, ticket):
        """Wait for the call behind ``ticket`` -> (idx, mean, var, ucb) arrays of length nseg."""
        nseg, _keep = self._open_tickets.pop(ticket)
        idx = np.empty(nseg, dtype=np.int64)
        mean = np.empty(nseg, dtype=np.float64)
        var = np.empty(nseg, dtype=np.float64)
        ucb = np.empty(nseg, dtype=np.float64)
        self._check(self._lib.gpso_best_ucb_end(self._h, int(ticket), L.i64ptr(idx), L.dptr(mean), L.dptr(var), L.dptr(ucb)))
        return idx, mean, var, ucb

    # -- ternary geometry ------------------------------------------------------------------------
    def grow_rows(self, depth):
        return int(self._lib.gpso_grow_rows(int(depth)))

    def grow(self, bounds, depth):
        """bounds [nseg, D, 2] (or [D, 2]) -> centres [nseg, rows, D] (or [rows, D]) float64."""
        b = L.as_f64(bounds)
        single = b.ndim == 2
        if single:
            b = b[None]
        nseg, d, two = b.shape
        assert two == 2
        rows = self.grow_rows(depth)
        out = np.empty((nseg, rows, d), dtype=np.float64)
        self._check(self._lib.gpso_grow(self._h, L.dptr(b), nseg, d, int(depth), L.dptr(out)))
        return out[0] if single else out

    def best_ucb_grow(self, bounds, depth, varsigma):
        b = L.as_f64(bounds)
        if b.ndim == 2:
            b = b[None]
        nseg, d, _ = b.shape
        if d != self.d:
            raise ValueError(f"bounds have D={d}, model has D={self.d}")
        idx, mean, var, ucb, pi, pm, pv, pu = self._winner_bufs(nseg)
        rc = self._lib.gpso_best_ucb_grow(self._h, L.dptr(b), nseg, int(depth), float(varsigma), pi, pm, pv, pu)
        if rc < 0:
            self._check(rc)
        return idx.copy(), mean.copy(), var.copy(), ucb.copy()

    # -- multi-GPU group (RCCL behind the C-ABI; see pygpso_amd/distributed.py) ------------------
    def comm_init(self, rank, world, unique_id):
        """Join a group (collective: every rank calls it with the same 128-byte id of
        ``distributed.unique_id()``)."""
        buf = C.create_string_buffer(bytes(unique_id), L.UNIQUE_ID_BYTES)
        self._check(self._lib.gpso_comm_init(self._h, int(rank), int(world), buf))
        self.rank, self.world = int(rank), int(world)

    def comm_destroy(self):
        self._check(self._lib.gpso_comm_destroy(self._h))
        self.rank, self.world = 0, 1

    def broadcast_posterior(self, root=0):
        """Collective: the posterior resident on ``root`` becomes resident here (RCCL broadcast)."""
        self._check(self._lib.gpso_broadcast_posterior(self._h, int(root)))
        n, d = C.c_int64(), C.c_int()
        self._check(self._lib.gpso_problem_shape(self._h, C.byref(n), C.byref(d)))
        self.n, self.d = int(n.value), int(d.value)

    def broadcast_posterior_rows(self, root=0):
        """Collective: after ``append`` on ``root`` only what the appends wrote travels (``gpso_broadcast_posterior_rows``);
        any rank that cannot take rows makes every rank take the whole range.  Returns True when rows sufficed;
        ``last_count(0)`` = the bytes that travelled."""
        rc = self._check(self._lib.gpso_broadcast_posterior_rows(self._h, int(root)))
        n, d = C.c_int64(), C.c_int()
        self._check(self._lib.gpso_problem_shape(self._h, C.byref(n), C.byref(d)))
        self.n, self.d = int(n.value), int(d.value)
        return rc == L.OK

    def posterior_dirty_ranges(self):
        """[(offset, nbytes), ...] of the posterior arena that a peer holding the posterior of the last hand-off lacks
        (``gpso_posterior_dirty_ranges``): [] = up to date; one range = the whole span when rows do not apply."""
        off = np.zeros(12, dtype=np.int64)
        nb = np.zeros(12, dtype=np.int64)
        cnt = self._check(self._lib.gpso_posterior_dirty_ranges(self._h, L.i64ptr(off), L.i64ptr(nb), 12))
        return [(int(off[i]), int(nb[i])) for i in range(cnt)]

    def posterior_mark_synced(self):
        """The peers now hold this posterior (after a hand-off by plain copies): the next ``posterior_dirty_ranges`` counts from here."""
        self._check(self._lib.gpso_posterior_mark_synced(self._h))

    def best_ucb_sharded(self, local_leaves, m_global, varsigma, seg_off=None):
        """Collective ``best_ucb``: ``local_leaves`` are this rank's rows
        ``distributed.shard_range(m_global, rank, world)`` of the batch; ``seg_off`` is global.
        Returns the global (idx, mean, var, ucb) per segment, identical on every rank."""
        ptr, dt, mem, m, keep = self._leaf_args(local_leaves)
        if seg_off is None:
            nseg, so_ptr = 1, None
        else:
            so = np.ascontiguousarray(seg_off, dtype=np.int64)
            nseg, so_ptr = int(so.shape[0] - 1), L.i64ptr(so)
        idx = np.empty(nseg, dtype=np.int64)
        mean = np.empty(nseg, dtype=np.float64)
        var = np.empty(nseg, dtype=np.float64)
        ucb = np.empty(nseg, dtype=np.float64)
        self._check(self._lib.gpso_best_ucb_sharded(self._h, ptr, dt, mem, m, int(m_global), so_ptr, nseg,
                                                    float(varsigma), L.i64ptr(idx), L.dptr(mean), L.dptr(var),
                                                    L.dptr(ucb)))
        return idx, mean, var, ucb

    def best_ucb_grow_sharded(self, bounds, depth, varsigma):
        """Collective ``best_ucb_grow``: every rank grows and scores its share of the reference rows."""
        b = L.as_f64(bounds)
        if b.ndim == 2:
            b = b[None]
        nseg, d, _ = b.shape
        if d != self.d:
            raise ValueError(f"bounds have D={d}, model has D={self.d}")
        idx = np.empty(nseg, dtype=np.int64)
        mean = np.empty(nseg, dtype=np.float64)
        var = np.empty(nseg, dtype=np.float64)
        ucb = np.empty(nseg, dtype=np.float64)
        self._check(self._lib.gpso_best_ucb_grow_sharded(self._h, L.dptr(b), nseg, int(depth), float(varsigma),
                                                         L.i64ptr(idx), L.dptr(mean), L.dptr(var), L.dptr(ucb)))
        return idx, mean, var, ucb

    def comm_abort(self):
        """Abort this engine's communicator (``ncclCommAbort``).  The one call that may come from ANOTHER thread
        than the one blocked inside a group call of this engine: that call then returns an error."""
        self._lib.gpso_comm_abort(self._h)

    # -- the two halves of the sharded calls, for an arbitrary (rank, world), no communicator needed ----
    def shard_winners(self, rank, world, local_leaves, m_global, varsigma, seg_off=None):
        """Rank ``rank``'s local half of ``best_ucb_sharded`` in a group of ``world``: its payload
        (``gpso_group_payload_doubles(nseg)`` float64: nseg x (mean, var, ucb, bit-cast index), spare, status)."""
        ptr, dt, mem, m, keep = self._leaf_args(local_leaves)
        if seg_off is None:
            nseg, so_ptr = 1, None
        else:
            so = np.ascontiguousarray(seg_off, dtype=np.int64)
            nseg, so_ptr = int(so.shape[0] - 1), L.i64ptr(so)
        payload = np.empty(self._lib.gpso_group_payload_doubles(nseg), dtype=np.float64)
        self._check(self._lib.gpso_shard_winners(self._h, int(rank), int(world), ptr, dt, mem, m, int(m_global),
                                                 so_ptr, nseg, float(varsigma), L.dptr(payload)))
        return payload

    def shard_winners_grow(self, rank, world, bounds, depth, varsigma):
        b = L.as_f64(bounds)
        if b.ndim == 2:
            b = b[None]
        nseg = b.shape[0]
        payload = np.empty(self._lib.gpso_group_payload_doubles(nseg), dtype=np.float64)
        self._check(self._lib.gpso_shard_winners_grow(self._h, int(rank), int(world), L.dptr(b), nseg, int(depth),
                                                      float(varsigma), L.dptr(payload)))
        return payload

    def fold_winners(self, payloads, nseg, m_global=-1, seg_off=None):
        """Payloads of all ranks (rank order) -> (idx, mean, var, ucb) per segment, as the group call returns
        them.  ``m_global`` >= 0 (with the global ``seg_off``): payloads of ``shard_winners``; < 0: of
        ``shard_winners_grow``."""
        g = np.ascontiguousarray(np.stack([np.asarray(q, dtype=np.float64) for q in payloads]))
        so_ptr = None
        if seg_off is not None:
            so = np.ascontiguousarray(seg_off, dtype=np.int64)
            so_ptr = L.i64ptr(so)
        idx = np.empty(nseg, dtype=np.int64)
        mean = np.empty(nseg, dtype=np.float64)
        var = np.empty(nseg, dtype=np.float64)
        ucb = np.empty(nseg, dtype=np.float64)
        self._check(self._lib.gpso_fold_winners(self._h, L.dptr(g), int(g.shape[0]), int(m_global), so_ptr, int(nseg),
                                                L.i64ptr(idx), L.dptr(mean), L.dptr(var), L.dptr(ucb)))
        return idx, mean, var, ucb

    # -- introspection -----------------------------------------------------------------------
    def get_matrix(self, which):
        out = np.empty((self.n, self.n), dtype=np.float64)
        self._check(self._lib.gpso_get_matrix(self._h, int(which), L.dptr(out)))
        return out

    def get_vector(self, which):
        out = np.empty(self.n, dtype=np.float64)
        self._check(self._lib.gpso_get_vector(self._h, int(which), L.dptr(out)))
        return out

    def posterior_buffers(self):
        """[(device_ptr, nbytes), ...] of what a peer rank needs to predict."""
        ptrs = (C.c_void_p * 8)()
        nb = (C.c_int64 * 8)()
        cnt = self._check(self._lib.gpso_posterior_buffers(self._h, ptrs, nb, 8))
        return [(int(ptrs[i] or 0), int(nb[i])) for i in range(cnt)]

    def posterior_span(self):
        """(device_ptr, offset, nbytes): the ONE contiguous range of this context's posterior arena the resident
        posterior uses -- what ``broadcast_posterior`` moves with a single ncclBroadcast."""
        ptr, off, nb = C.c_void_p(), C.c_int64(), C.c_int64()
        self._check(self._lib.gpso_posterior_span(self._h, C.byref(ptr), C.byref(off), C.byref(nb)))
        return int(ptr.value or 0), int(off.value), int(nb.value)

    def posterior_span_at(self, offset, nbytes):
        """Device pointer of (offset, nbytes) -- a peer's ``posterior_span`` -- inside THIS context's arena."""
        ptr = C.c_void_p()
        self._check(self._lib.gpso_posterior_span_at(self._h, int(offset), int(nbytes), C.byref(ptr)))
        return int(ptr.value or 0)

    def posterior_hash(self):
        """64-bit fingerprint of the resident predict-ready posterior (``gpso_posterior_hash``)."""
        out = C.c_uint64()
        self._check(self._lib.gpso_posterior_hash(self._h, C.byref(out)))
        return int(out.value)

    def alloc_posterior(self, n, d):
        self._check(self._lib.gpso_alloc_posterior(self._h, int(n), int(d)))
        self.n, self.d = int(n), int(d)

    def adopt_posterior(self):
        self._check(self._lib.gpso_adopt_posterior(self._h))
        n, d = C.c_int64(), C.c_int()  # (a posterior extended by gpso_append on the sender arrives with more rows)
        self._check(self._lib.gpso_problem_shape(self._h, C.byref(n), C.byref(d)))
        self.n, self.d = int(n.value), int(d.value)
