"""Host-side engine objects over the C-ABI (one context per GPU).

DenseEngine drives the dense (OnOffSVGP) path: data is made resident in HBM once (`set_data`), then every
`elbo` call is one ELBO value (+ gradient) = one optimiser step of the reference
(GPflow Model.optimize function evaluation / sess.run(train_op), scripts/onoff.py:379).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ZigpError, NotPositiveDefiniteError, as_f64, ptr

PARAM_KEYS = ('Zf', 'Zg', 'u_fm', 'u_gm', 'u_fs_sqrt', 'u_gs_sqrt', 'ell_f', 'ell_g', 'var_f', 'var_g', 'noise')


def _check(lib, ctx, rc):
    if rc == 0:
        return
    msg = lib.zigp_last_error(ctx)
    msg = msg.decode() if msg else 'unknown error'
    if rc == _lib.ZIGP_ENOTPD:
        raise NotPositiveDefiniteError(msg)
    if rc == _lib.ZIGP_EARG:
        raise ValueError(msg)
    raise ZigpError('libzigp error %d: %s' % (rc, msg))


class _Packed:
    """Keeps the numpy arrays behind a zigp_params struct alive."""

    def __init__(self, p):
        Zf, Zg = as_f64(p['Zf']), as_f64(p['Zg'])
        if Zf.ndim != 2 or Zg.ndim != 2 or Zf.shape[1] != Zg.shape[1]:
            raise ValueError('Zf and Zg must be (M,D) with equal D')
        D = Zf.shape[1]
        self.D, self.Mf, self.Mg = D, Zf.shape[0], Zg.shape[0]

        def ell(v):
            v = as_f64(v).reshape(-1)
            if v.size == 1:
                v = np.full(D, float(v[0]))
            if v.size != D:
                raise ValueError('lengthscales must be scalar or have D entries')
            return np.ascontiguousarray(v)

        self.arr = dict(Zf=Zf, Zg=Zg, u_fm=as_f64(p['u_fm']).reshape(-1), u_gm=as_f64(p['u_gm']).reshape(-1),
                        u_fs_sqrt=as_f64(p['u_fs_sqrt']).reshape(-1), u_gs_sqrt=as_f64(p['u_gs_sqrt']).reshape(-1),
                        ell_f=ell(p['ell_f']), ell_g=ell(p['ell_g']))
        for k, M in (('u_fm', self.Mf), ('u_fs_sqrt', self.Mf), ('u_gm', self.Mg), ('u_gs_sqrt', self.Mg)):
            if self.arr[k].size != M:
                raise ValueError('%s must have %d entries' % (k, M))
        s = _lib.zigp_params()
        s.Mf, s.Mg, s.D = self.Mf, self.Mg, D
        for k, a in self.arr.items():
            setattr(s, k, ptr(a))
        s.var_f, s.var_g, s.noise = float(np.squeeze(p['var_f'])), float(np.squeeze(p['var_g'])), float(np.squeeze(p['noise']))
        self.struct = s


def _scalar(v):
    """float of a Python number or a size-1 array (np.squeeze on a plain float costs 1.5 us)"""
    return float(v) if isinstance(v, (float, int)) else float(np.asarray(v).reshape(-1)[0])


class KronStepper:
    """see DenseEngine.kron_stepper"""
    _ARR = (('Z', 0), ('Z', 1), ('ell_', 0), ('ell_', 1))

    def __init__(self, engine, p):
        self.eng = engine
        s, keep, dims = engine._pack_kron(p)            # private copies below: the caller's arrays may be views that change or go away
        self.dims = dims
        self.par = {}
        self.s = _lib.zigp_kron_params()
        self.gs = _lib.zigp_kron_grads()
        sizes = []
        for tag in ('f', 'g'):
            Z0, Z1, l0, l1, um, us = keep[tag]
            self.par[tag] = [np.array(a, dtype=np.float64, order='C', copy=True) for a in (Z0, Z1, l0, l1, um, us)]
            sizes += [a.size for a in self.par[tag]]
        self.flat = np.zeros(sum(sizes))
        self.out = {}
        o = 0
        for tag in ('f', 'g'):
            Z0, Z1, l0, l1, um, us = self.par[tag]
            setattr(self.s, 'M0' + tag, Z0.shape[0]); setattr(self.s, 'M1' + tag, Z1.shape[0])
            for name, a in (('Z0', Z0), ('Z1', Z1), ('ell0', l0), ('ell1', l1)):
                setattr(self.s, name + tag, ptr(a))
            setattr(self.s, 'u_%sm' % tag, ptr(um)); setattr(self.s, 'u_%ss_sqrt' % tag, ptr(us))
            views = []
            for a in self.par[tag]:
                views.append(self.flat[o:o + a.size].reshape(a.shape)); o += a.size
            for name, v in zip(('Z0', 'Z1', 'ell0', 'ell1'), views[:4]):
                setattr(self.gs, name + tag, ptr(v))
            setattr(self.gs, 'u_%sm' % tag, ptr(views[4])); setattr(self.gs, 'u_%ss_sqrt' % tag, ptr(views[5]))
            self.out['Z' + tag] = [views[0], views[1]]
            self.out['ell_' + tag] = [views[2], views[3]]
            self.out['u_%sm' % tag] = views[4]
            self.out['u_%ss_sqrt' % tag] = views[5]
        self.s.D0, self.s.D1 = dims
        self._ed, self._kl, self._dmu = C.c_double(0), C.c_double(0), C.c_double(0)
        self._sref, self._gref = C.byref(self.s), C.byref(self.gs)
        self._edref, self._klref, self._dmuref = C.byref(self._ed), C.byref(self._kl), C.byref(self._dmu)

    def __call__(self, p, X=None, Y=None, rows=None, jitter=1e-5, scale=1.0, g_offset=0.0, include_kl=True, f_mu=None):
        s, copyto = self.s, np.copyto
        for tag in ('f', 'g'):
            b = self.par[tag]
            Z, ell, var = p['Z' + tag], p['ell_' + tag], p['var_' + tag]
            copyto(b[0], Z[0]); copyto(b[1], Z[1])
            copyto(b[2], np.reshape(ell[0], -1)); copyto(b[3], np.reshape(ell[1], -1))      # a scalar lengthscale broadcasts over the factor's columns
            copyto(b[4], np.reshape(p['u_%sm' % tag], -1)); copyto(b[5], np.reshape(p['u_%ss_sqrt' % tag], -1))
            setattr(s, 'var0' + tag, _scalar(var[0])); setattr(s, 'var1' + tag, _scalar(var[1]))
        s.noise = _scalar(p.get('noise', 1.0))
        fmu = 0.0 if f_mu is None else _scalar(f_mu)
        eng = self.eng
        if rows is None:
            X = as_f64(X)
            Y = as_f64(Y).reshape(-1)
            if X.ndim != 2 or X.shape[1] != self.dims[0] + self.dims[1] or Y.size != X.shape[0]:
                raise ValueError('X must be (N,%d) and Y have N entries' % (self.dims[0] + self.dims[1]))
            rc = eng.lib.zigp_kron_elbo(eng.ctx, self._sref, ptr(X), ptr(Y), X.shape[0], float(jitter), float(scale), float(g_offset), fmu,
                                        1 if include_kl else 0, self._edref, self._klref, self._gref, self._dmuref)
        else:
            rc = eng.lib.zigp_kron_elbo_rows(eng.ctx, self._sref, int(rows[0]), int(rows[1]), float(jitter), float(scale), float(g_offset), fmu,
                                             1 if include_kl else 0, self._edref, self._klref, self._gref, self._dmuref)
        _check(eng.lib, eng.ctx, rc)
        gs, out = self.gs, self.out
        out['noise'] = gs.noise
        out['var_f'] = [gs.var0f, gs.var1f]
        out['var_g'] = [gs.var0g, gs.var1g]
        if f_mu is not None:
            out['f_mu'] = self._dmu.value
        else:
            out.pop('f_mu', None)
        return self._ed.value, self._kl.value, out


class DenseEngine:
    def __init__(self, device=0, pivot_rtol=None):
        """pivot_rtol: smallest accepted Cholesky pivot as a multiple of eps * (variance + jitter).  None keeps the library default
        (8: an exactly singular Kuu is reported whichever way its rounding-noise pivot falls); 0 is tf.cholesky's bare `pivot > 0` test
        (onofftf/main.py:200,268,355) -- what the reference look-alikes use: reference_engine() below."""
        self.lib = _lib.load()
        self.ctx = C.c_void_p()
        rc = self.lib.zigp_create(C.byref(self.ctx), int(device))
        if rc != 0:
            raise ZigpError('zigp_create failed (rc=%d): no usable HIP device %d' % (rc, device))
        self.device = int(device)
        self.N = 0
        self._full_N = 0
        self.D = 0
        self._keep = None
        if pivot_rtol is not None:
            self.set_pivot_rtol(pivot_rtol)

    def close(self):
        if self.ctx:
            self.lib.zigp_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_chunk(self, rows):
        """rows per pass (a multiple of 1024); None or 0: the library's automatic rule (see get_chunk_rows)"""
        _check(self.lib, self.ctx, self.lib.zigp_set_chunk(self.ctx, int(rows or 0)))

    def get_chunk(self, M):
        """rows per pass of the dense path at M inducing points per latent (the set_chunk value, else the library's rule)"""
        r = int(self.lib.zigp_get_chunk(self.ctx, int(M)))
        if r < 0:
            raise ValueError('zigp_get_chunk: bad M')
        return r

    def get_chunk_rows(self, M, span):
        """rows per pass the dense path uses for a row range of `span` rows at M inducing points per latent (equal passes; short ranges
        go through in one pass while the panels stay within 9 GB)"""
        r = int(self.lib.zigp_get_chunk_rows(self.ctx, int(M), int(span)))
        if r < 0:
            raise ValueError('zigp_get_chunk_rows: bad M or span')
        return r

    def set_pivot_rtol(self, rtol):
        """smallest accepted Cholesky pivot = rtol * eps * (variance + jitter); default 8, 0 = tf.cholesky's bare pivot > 0 test"""
        _check(self.lib, self.ctx, self.lib.zigp_set_pivot_rtol(self.ctx, float(rtol)))

    # ---- data-parallel exchange inside the library (RCCL; include/zigp.h 'data-parallel exchange') ----
    def comm_unique_id(self):
        """rank 0: the 128-byte id every rank hands to comm_init"""
        buf = (C.c_char * _lib.COMM_ID_BYTES)()
        rc = self.lib.zigp_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise ZigpError('zigp_comm_unique_id failed (rc=%d): librccl.so.1 not loadable?' % rc)
        return bytes(buf.raw)

    def comm_init(self, rank, nranks, unique_id):
        """collective; afterwards elbo() / kron_elbo() / kron_head_elbo() return sums over ranks (pass include_kl on rank 0 only)"""
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError('unique_id must have %d bytes' % _lib.COMM_ID_BYTES)
        buf = C.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES)
        _check(self.lib, self.ctx, self.lib.zigp_comm_init(self.ctx, int(rank), int(nranks), C.cast(buf, C.c_void_p)))

    def comm_available(self):
        """True when RCCL can be bound in this process (zigp_comm_available): ranks agree on this before any of them enters comm_init"""
        v = C.c_int32(0)
        return self.lib.zigp_comm_available(C.byref(v)) == 0

    def comm_set_timeout(self, seconds):
        """how long comm_init waits for its peers before it gives up with a ZigpError (default 120 s)"""
        _check(self.lib, self.ctx, self.lib.zigp_comm_set_timeout(self.ctx, float(seconds)))

    def comm_destroy(self):
        _check(self.lib, self.ctx, self.lib.zigp_comm_destroy(self.ctx))

    def comm_allreduce(self, vec):
        """sum a small float64 host vector over the ranks of the library's communicator (returns a new array)"""
        v = np.array(vec, dtype=np.float64).reshape(-1)
        _check(self.lib, self.ctx, self.lib.zigp_comm_allreduce_host(self.ctx, ptr(v), v.size))
        return v

    def comm_info(self):
        r, n, k = C.c_int32(0), C.c_int32(0), C.c_int64(0)
        _check(self.lib, self.ctx, self.lib.zigp_comm_info(self.ctx, C.byref(r), C.byref(n), C.byref(k)))
        return dict(rank=r.value, nranks=n.value, allreduce_calls=k.value)

    def set_data(self, X, Y):
        X = as_f64(X)
        if X.ndim != 2:
            raise ValueError('X must be (N,D)')
        Y = as_f64(Y).reshape(-1)
        if Y.size != X.shape[0]:
            raise ValueError('Y must have N entries')
        _check(self.lib, self.ctx, self.lib.zigp_set_data(self.ctx, ptr(X), ptr(Y), X.shape[0], X.shape[1]))
        self.N, self.D = X.shape
        self._full_N = self.N

    def set_data_device(self, X_t, Y_t):
        """Adopt torch CUDA float64 tensors (no copy); they are kept alive by this object."""
        import torch
        N, D = X_t.shape
        if not (X_t.is_cuda and Y_t.is_cuda and X_t.device.index == self.device and Y_t.device.index == self.device):
            raise ValueError('set_data_device: tensors must live on cuda:%d, the engine\'s device' % self.device)
        if not (X_t.is_contiguous() and Y_t.is_contiguous() and X_t.dtype == torch.float64 and Y_t.dtype == torch.float64):
            raise ValueError('set_data_device: need contiguous float64 tensors')
        if Y_t.numel() != N:
            raise ValueError('Y must have N entries')
        # the engine's streams are non-blocking: nothing orders them after the torch stream that produced the tensors
        torch.cuda.current_stream(self.device).synchronize()
        self._keep = (X_t, Y_t)
        _check(self.lib, self.ctx, self.lib.zigp_set_data_device(self.ctx, C.c_void_p(X_t.data_ptr()), C.c_void_p(Y_t.data_ptr()), N, D))
        self.N, self.D = N, D
        self._full_N = N

    def select_rows(self, idx=None):
        """Minibatch by row indices (repeats allowed): the rows `idx` of the resident data set are gathered on the device and become
        the active data of the calls that follow (rows 0 .. len(idx)); None / empty: back to the whole resident set."""
        if idx is None or len(idx) == 0:
            _check(self.lib, self.ctx, self.lib.zigp_select_rows(self.ctx, None, 0))
            self.N = self._full_N
            return
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int64))
        _check(self.lib, self.ctx, self.lib.zigp_select_rows(self.ctx, idx.ctypes.data, idx.size))
        self.N = int(idx.size)

    def elbo(self, p, jitter=1e-6, scale=1.0, g_offset=0.0, rows=None, include_kl=True, need_grad=True):
        """Returns (elbo_data, kl, grads or None); ELBO = elbo_data - kl."""
        pk = _Packed(p)
        mean_D = self._set_mean_function(p, pk.D)
        r0, r1 = (0, self.N) if rows is None else rows
        ed, kl = C.c_double(0), C.c_double(0)
        gs, g = None, None
        if need_grad:
            g = dict(Zf=np.zeros((pk.Mf, pk.D)), Zg=np.zeros((pk.Mg, pk.D)), u_fm=np.zeros(pk.Mf), u_gm=np.zeros(pk.Mg),
                     u_fs_sqrt=np.zeros(pk.Mf), u_gs_sqrt=np.zeros(pk.Mg), ell_f=np.zeros(pk.D), ell_g=np.zeros(pk.D))
            gs = _lib.zigp_grads()
            for k, a in g.items():
                setattr(gs, k, ptr(a))
        rc = self.lib.zigp_elbo(self.ctx, C.byref(pk.struct), float(jitter), float(scale), float(g_offset), int(r0), int(r1),
                                1 if include_kl else 0, C.byref(ed), C.byref(kl), C.byref(gs) if gs is not None else None)
        _check(self.lib, self.ctx, rc)
        if need_grad:
            g['var_f'], g['var_g'], g['noise'] = gs.var_f, gs.var_g, gs.noise
            if mean_D is not None:
                da, db = np.zeros(max(mean_D, 1)), C.c_double(0)
                _check(self.lib, self.ctx, self.lib.zigp_get_mean_function_grad(self.ctx, ptr(da), mean_D, C.byref(db)))
                g['mean_a'], g['mean_b'] = da[:mean_D], db.value
        return ed.value, kl.value, g

    def _set_mean_function(self, p, D):
        """p may carry the mean function of f, m(x) = mean_b + mean_a . x (OnOffSVGP.py:29,134): 'mean_b' alone is GPflow's
        Constant, 'mean_a' (+ 'mean_b') its Linear; neither is Zero.  Returns len(mean_a) when one is set, else None."""
        a, b = p.get('mean_a'), p.get('mean_b')
        if a is None and b is None:
            _check(self.lib, self.ctx, self.lib.zigp_set_mean_function(self.ctx, None, -1, 0.0))
            return None
        a = np.zeros(0) if a is None else as_f64(a).reshape(-1)
        if a.size not in (0, D):
            raise ValueError('mean_a must have D entries')
        _check(self.lib, self.ctx, self.lib.zigp_set_mean_function(self.ctx, ptr(a) if a.size else None, a.size,
                                                                    float(np.squeeze(0.0 if b is None else b))))
        return a.size

    def predict(self, p, Xnew, jitter=1e-6, g_offset=0.0):
        """(9,N) array in the order of OnOffSVGP.build_predict (onoffgpf/OnOffSVGP.py:152)."""
        pk = _Packed(p)
        self._set_mean_function(p, pk.D)
        Xnew = as_f64(Xnew)
        if Xnew.ndim != 2 or Xnew.shape[1] != pk.D:
            raise ValueError('Xnew must be (N,%d)' % pk.D)
        out = np.zeros((9, Xnew.shape[0]))
        _check(self.lib, self.ctx, self.lib.zigp_predict(self.ctx, C.byref(pk.struct), ptr(Xnew), Xnew.shape[0], float(jitter),
                                                          float(g_offset), ptr(out)))
        return out

    def predict_device(self, p, X_t, jitter=1e-6, g_offset=0.0, out=None):
        """predict on a torch CUDA float64 tensor (N,D) of the engine's device; returns (or fills `out` with) a (9,N) CUDA tensor in the
        order of OnOffSVGP.build_predict -- nothing but the parameters crosses PCIe."""
        import torch
        pk = _Packed(p)
        self._set_mean_function(p, pk.D)
        if not (X_t.is_cuda and X_t.device.index == self.device and X_t.dtype == torch.float64 and X_t.is_contiguous() and X_t.dim() == 2 and X_t.shape[1] == pk.D):
            raise ValueError('predict_device: need a contiguous float64 (N,%d) tensor on cuda:%d' % (pk.D, self.device))
        N = int(X_t.shape[0])
        if out is None:
            out = torch.empty((9, N), dtype=torch.float64, device=X_t.device)
        elif not (out.is_cuda and out.device.index == self.device and out.dtype == torch.float64 and out.is_contiguous() and tuple(out.shape) == (9, N)):
            raise ValueError('predict_device: out must be a contiguous float64 (9,N) tensor on the same device')
        torch.cuda.current_stream(self.device).synchronize()      # the engine's streams are not ordered after torch's
        if N:
            _check(self.lib, self.ctx, self.lib.zigp_predict_device(self.ctx, C.byref(pk.struct), C.c_void_p(X_t.data_ptr()), N, float(jitter),
                                                                     float(g_offset), C.c_void_p(out.data_ptr())))
        return out

    def prior_kl(self, p, jitter=1e-6):
        pk = _Packed(p)
        out = np.zeros(2)
        _check(self.lib, self.ctx, self.lib.zigp_prior_kl(self.ctx, C.byref(pk.struct), float(jitter), ptr(out)))
        return out

    def rbf_K(self, X1, X2, ell, var):
        X1 = as_f64(X1)
        D = X1.shape[1]
        X2a = X1 if X2 is None else as_f64(X2)
        ell = as_f64(ell).reshape(-1)
        if ell.size == 1:
            ell = np.full(D, float(ell[0]))
        out = np.zeros((X1.shape[0], X2a.shape[0]))
        _check(self.lib, self.ctx, self.lib.zigp_rbf_K(self.ctx, ptr(X1), X1.shape[0], None if X2 is None else ptr(X2a), X2a.shape[0], D,
                                                        ptr(ell), float(np.squeeze(var)), ptr(out)))
        return out

    # ---- Kronecker (space x time) variant -------------------------------------------------------
    @staticmethod
    def _pack_kron(p, tags=('f', 'g')):
        """p: Zf/Zg = [Z0 (M0,D0), Z1 (M1,D1)], ell_f/ell_g = [l0, l1], var_f/var_g = [v0, v1], u_* (M0*M1), noise."""
        keep = {}
        s = _lib.zigp_kron_params()

        def ell(v, D):
            v = as_f64(v).reshape(-1)
            if v.size == 1:
                v = np.full(D, float(v[0]))
            if v.size != D:
                raise ValueError('lengthscales must be scalar or match the factor dimension')
            return np.ascontiguousarray(v)

        dims = None
        for tag in tags:
            Z0, Z1 = as_f64(p['Z' + tag][0]), as_f64(p['Z' + tag][1])
            if Z0.ndim != 2 or Z1.ndim != 2:
                raise ValueError('factor inducing inputs must be 2-D')
            if dims is None:
                dims = (Z0.shape[1], Z1.shape[1])
            elif dims != (Z0.shape[1], Z1.shape[1]):
                raise ValueError('f and g factors must act on the same input columns')
            M0, M1 = Z0.shape[0], Z1.shape[0]
            l0, l1 = ell(p['ell_' + tag][0], dims[0]), ell(p['ell_' + tag][1], dims[1])
            um = as_f64(p['u_%sm' % tag]).reshape(-1)
            us = as_f64(p['u_%ss_sqrt' % tag]).reshape(-1)
            if um.size != M0 * M1 or us.size != M0 * M1:
                raise ValueError('u_%sm / u_%ss_sqrt must have M0*M1 entries' % (tag, tag))
            keep[tag] = (Z0, Z1, l0, l1, um, us)
            setattr(s, 'M0' + tag, M0); setattr(s, 'M1' + tag, M1)
            setattr(s, 'Z0' + tag, ptr(Z0)); setattr(s, 'Z1' + tag, ptr(Z1))
            setattr(s, 'ell0' + tag, ptr(l0)); setattr(s, 'ell1' + tag, ptr(l1))
            setattr(s, 'var0' + tag, _scalar(p['var_' + tag][0])); setattr(s, 'var1' + tag, _scalar(p['var_' + tag][1]))
            setattr(s, 'u_%sm' % tag, ptr(um)); setattr(s, 'u_%ss_sqrt' % tag, ptr(us))
        s.D0, s.D1 = dims
        s.noise = _scalar(p.get('noise', 1.0))
        return s, keep, dims

    def kron_elbo(self, p, X=None, Y=None, jitter=1e-5, scale=1.0, g_offset=0.0, include_kl=True, need_grad=True, rows=None, f_mu=None):
        """One Kronecker ELBO (step); returns (elbo_data, kl, grads or None).  Either an explicit host minibatch (X, Y), as
        scripts/onoff.py:377-381 feeds it, or rows=(lo, hi) of the resident data set (set_data / set_data_device).
        f_mu: the optional constant added to fmean (scripts/onoff.py:161,168-169); when given, grads['f_mu'] is its gradient."""
        s, keep, dims = self._pack_kron(p)
        if rows is None:
            X = as_f64(X)
            if X.ndim != 2 or X.shape[1] != dims[0] + dims[1]:
                raise ValueError('X must be (N,%d)' % (dims[0] + dims[1]))
            Y = as_f64(Y).reshape(-1)
            if Y.size != X.shape[0]:
                raise ValueError('Y must have N entries')
        elif X is not None or Y is not None:
            raise ValueError('pass either (X, Y) or rows=(lo, hi), not both')
        ed, kl, dmu = C.c_double(0), C.c_double(0), C.c_double(0)
        fmu = 0.0 if f_mu is None else _scalar(f_mu)
        g, gs = None, None
        if need_grad:
            g = {}
            gs = _lib.zigp_kron_grads()
            # one allocation for the twelve gradient arrays (views into it): np.zeros_like twelve times is 6 us of a 190 us call
            sizes = [a.size for tag in ('f', 'g') for a in keep[tag]]
            flat = np.zeros(sum(sizes))
            o = 0
            for tag in ('f', 'g'):
                Z0, Z1, l0, l1, um, us = keep[tag]
                arrs = {}
                for name, a in (('Z0', Z0), ('Z1', Z1), ('ell0', l0), ('ell1', l1), ('um', um), ('us', us)):
                    arrs[name] = flat[o:o + a.size].reshape(a.shape); o += a.size
                g[tag] = arrs
                setattr(gs, 'Z0' + tag, ptr(arrs['Z0'])); setattr(gs, 'Z1' + tag, ptr(arrs['Z1']))
                setattr(gs, 'ell0' + tag, ptr(arrs['ell0'])); setattr(gs, 'ell1' + tag, ptr(arrs['ell1']))
                setattr(gs, 'u_%sm' % tag, ptr(arrs['um'])); setattr(gs, 'u_%ss_sqrt' % tag, ptr(arrs['us']))
        gref = C.byref(gs) if gs is not None else None
        if rows is None:
            rc = self.lib.zigp_kron_elbo(self.ctx, C.byref(s), ptr(X), ptr(Y), X.shape[0], float(jitter), float(scale), float(g_offset), fmu,
                                         1 if include_kl else 0, C.byref(ed), C.byref(kl), gref, C.byref(dmu))
        else:
            rc = self.lib.zigp_kron_elbo_rows(self.ctx, C.byref(s), int(rows[0]), int(rows[1]), float(jitter), float(scale), float(g_offset), fmu,
                                              1 if include_kl else 0, C.byref(ed), C.byref(kl), gref, C.byref(dmu))
        _check(self.lib, self.ctx, rc)
        out = None
        if need_grad:
            out = dict(noise=gs.noise)
            if f_mu is not None:
                out['f_mu'] = dmu.value
            for tag in ('f', 'g'):
                a = g[tag]
                out['Z' + tag] = [a['Z0'], a['Z1']]
                out['ell_' + tag] = [a['ell0'], a['ell1']]
                out['var_' + tag] = [getattr(gs, 'var0' + tag), getattr(gs, 'var1' + tag)]
                out['u_%sm' % tag] = a['um']
                out['u_%ss_sqrt' % tag] = a['us']
        return ed.value, kl.value, out

    def kron_fit_steps(self, shape, x, m, v, lr, positive, t0, row_begin, batch, jitter=1e-5, scale=1.0, Xw=None, Yw=None,
                       beta1=0.9, beta2=0.999, eps=1e-8, include_kl=True):
        """n = len(row_begin) Adam steps of the Kronecker on/off fit ON THE DEVICE (zigp_kron_fit_steps; the loop body of
        scripts/onoff.py:375-381): one synchronisation for the whole call.
        shape: dict(M0f, M1f, M0g, M1g, D0, D1); x, m, v: float64 [n_free] free state and Adam moments, UPDATED IN PLACE, in the block
        order of include/zigp.h (per latent Z0, Z1, u, s, ell0, ell1, var0, var1; then the noise variance); lr, positive: 17 per-block
        learning rates / Log1pe flags; t0: iterations done so far; row_begin[i] >= 0: rows of the resident data set, -(1 + k): batch k of
        the host arrays (Xw, Yw).  Returns (elbo_data[n], kl[n]) -- the history, each at the parameters before that step's update."""
        s = _lib.zigp_kron_params()
        for k in ('M0f', 'M1f', 'M0g', 'M1g', 'D0', 'D1'):
            setattr(s, k, int(shape[k]))
        o = _lib.zigp_kron_fit_opts()
        for b in range(_lib.FIT_BLOCKS):
            o.lr[b] = float(lr[b]); o.positive[b] = int(bool(positive[b]))
        o.beta1, o.beta2, o.eps = float(beta1), float(beta2), float(eps)
        for a in (x, m, v):
            if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.ndim == 1 and a.size == x.size):
                raise ValueError('x, m, v must be contiguous float64 vectors of one length')
        rb = np.ascontiguousarray(np.asarray(row_begin, dtype=np.int64))
        n = rb.size
        xw = yw = None
        if Xw is not None:
            xw, yw = as_f64(Xw), as_f64(Yw).reshape(-1)
            if xw.ndim != 2 or xw.shape[0] % int(batch) or yw.size != xw.shape[0]:
                raise ValueError('Xw must hold whole batches of `batch` rows, Yw one value per row')
            if rb.min() < -(xw.shape[0] // int(batch)):
                raise ValueError('row_begin refers to a host batch beyond Xw')
        elif n and rb.min() < 0:
            raise ValueError('negative row_begin needs Xw, Yw')
        ed, kl = np.zeros(n), np.zeros(n)
        rc = self.lib.zigp_kron_fit_steps(self.ctx, C.byref(s), C.byref(o), ptr(x), ptr(m), ptr(v), x.size, int(t0), n, rb.ctypes.data, int(batch),
                                          ptr(xw) if xw is not None else None, ptr(yw) if yw is not None else None, float(jitter), float(scale),
                                          1 if include_kl else 0, ptr(ed), ptr(kl))
        try:
            _check(self.lib, self.ctx, rc)
        except ZigpError as e:
            # include/zigp.h: after a failure x / m / v hold the state before the failing step and the history from that step on is NaN --
            # the caller learns how many updates WERE applied (its iteration count and Adam's bias correction depend on it) and gets the
            # history of those steps
            # (the count comes from the library -- zigp_kron_fit_steps_applied; the first non-finite history entry is NOT it: an applied
            # step can itself have a non-finite ELBO at extreme parameters)
            done = max(0, min(n, int(self.lib.zigp_kron_fit_steps_applied(self.ctx))))
            e.steps_applied = done
            e.elbo_data, e.kl = ed[:done].copy(), kl[:done].copy()
            raise
        return ed, kl

    def kron_stepper(self, p):
        """A prepared Kronecker step for a FIXED model shape (the training loop: scripts/onoff.py:375-431 calls sess.run on one graph 50 000
        times): parameter buffers, the ctypes structs and the gradient arrays are set up once; a call copies the new parameter values in
        and returns views of its own gradient buffer (valid until the next call).  Same entry points, same numbers as kron_elbo."""
        return KronStepper(self, p)

    def kron_predict(self, p, Xnew, jitter=1e-6, g_offset=0.0, f_mu=None):
        """(9,N) in the order of build_predict (scripts/onoff.py:184); onofftf/onoffpred.py uses jitter 1e-6, g_offset -1."""
        s, keep, dims = self._pack_kron(p)
        Xnew = as_f64(Xnew)
        if Xnew.ndim != 2 or Xnew.shape[1] != dims[0] + dims[1]:
            raise ValueError('Xnew must be (N,%d)' % (dims[0] + dims[1]))
        out = np.zeros((9, Xnew.shape[0]))
        _check(self.lib, self.ctx, self.lib.zigp_kron_predict(self.ctx, C.byref(s), ptr(Xnew), Xnew.shape[0], float(jitter),
                                                               float(g_offset), 0.0 if f_mu is None else _scalar(f_mu), ptr(out)))
        return out

    LIK = {'gaussian': _lib.LIK_GAUSSIAN, 'bernoulli': _lib.LIK_BERNOULLI}

    def kron_head_elbo(self, p, X, Y, lik, jitter=1e-5, scale=1.0, f_mu=0.0, include_kl=True, need_grad=True):
        """Single-latent Kronecker SVGP bound with a 'gaussian' (scripts/svgp.py, hurdle.py) or 'bernoulli'
        (scripts/classifier.py) head.  p holds the f fields only (Zf, ell_f, var_f, u_fm, u_fs_sqrt[, noise]).
        Returns (elbo_data, kl, grads or None); grads has the f keys, 'noise' and 'f_mu'."""
        s, keep, dims = self._pack_kron(p, tags=('f',))
        X = as_f64(X)
        if X.ndim != 2 or X.shape[1] != dims[0] + dims[1]:
            raise ValueError('X must be (N,%d)' % (dims[0] + dims[1]))
        Y = as_f64(Y).reshape(-1)
        if Y.size != X.shape[0]:
            raise ValueError('Y must have N entries')
        ed, kl, dmu = C.c_double(0), C.c_double(0), C.c_double(0)
        gs, a = None, None
        if need_grad:
            gs = _lib.zigp_kron_grads()
            Z0, Z1, l0, l1, um, us = keep['f']
            a = dict(Z0=np.zeros_like(Z0), Z1=np.zeros_like(Z1), ell0=np.zeros_like(l0), ell1=np.zeros_like(l1),
                     um=np.zeros_like(um), us=np.zeros_like(us))
            gs.Z0f, gs.Z1f, gs.ell0f, gs.ell1f = ptr(a['Z0']), ptr(a['Z1']), ptr(a['ell0']), ptr(a['ell1'])
            gs.u_fm, gs.u_fs_sqrt = ptr(a['um']), ptr(a['us'])
        rc = self.lib.zigp_kron_head_elbo(self.ctx, C.byref(s), self.LIK[lik], ptr(X), ptr(Y), X.shape[0], float(jitter), float(scale),
                                          float(f_mu), 1 if include_kl else 0, C.byref(ed), C.byref(kl),
                                          C.byref(gs) if gs is not None else None, C.byref(dmu))
        _check(self.lib, self.ctx, rc)
        out = None
        if need_grad:
            out = dict(noise=gs.noise, f_mu=dmu.value, Zf=[a['Z0'], a['Z1']], ell_f=[a['ell0'], a['ell1']], var_f=[gs.var0f, gs.var1f],
                       u_fm=a['um'], u_fs_sqrt=a['us'])
        return ed.value, kl.value, out

    def kron_head_predict(self, p, Xnew, lik, jitter=1e-6, f_mu=0.0):
        """(4,N): fmean, fvar, pfmean, pfvar (onofftf/svgppred.py:180-186, onofftf/svcppred.py 'pfmean'/'pfvar')."""
        s, keep, dims = self._pack_kron(p, tags=('f',))
        Xnew = as_f64(Xnew)
        if Xnew.ndim != 2 or Xnew.shape[1] != dims[0] + dims[1]:
            raise ValueError('Xnew must be (N,%d)' % (dims[0] + dims[1]))
        out = np.zeros((4, Xnew.shape[0]))
        _check(self.lib, self.ctx, self.lib.zigp_kron_head_predict(self.ctx, C.byref(s), self.LIK[lik], ptr(Xnew), Xnew.shape[0],
                                                                    float(jitter), float(f_mu), ptr(out)))
        return out

    # ---- measurement ----
    def set_overlap(self, on=True):
        """stream overlap inside elbo(): True/1 (the library's DEFAULT since round 3) runs the side kernels of a chunk (kgrad, the next
        chunk's Kuf panels) on a second stream under its rank-N updates; False/0 keeps every launch on one stream -- set this for
        per-kernel timing with an external tracer (rocprofv3), where overlapped kernels stretch each other.  Results are bit-identical
        either way."""
        _check(self.lib, self.ctx, self.lib.zigp_set_overlap(self.ctx, int(on)))

    def set_kron_panels(self, on=True):
        """diagnostic: route the Kronecker entry points through the GEMM-panel path also for grids the fused kernels cover"""
        _check(self.lib, self.ctx, self.lib.zigp_set_kron_panels(self.ctx, 1 if on else 0))

    def set_kron_range_tiles(self, tiles=1024):
        """diagnostic: rows per range (in 16-point tiles) of the larger-grid Kronecker gradient step; results do not depend on it"""
        _check(self.lib, self.ctx, self.lib.zigp_set_kron_range_tiles(self.ctx, int(tiles)))

    def profile_enable(self, on=True):
        _check(self.lib, self.ctx, self.lib.zigp_profile_enable(self.ctx, 1 if on else 0))

    def clock_stamp(self):
        """{XCC id: (shader-clock counter, 100 MHz counter)} taken in stream order now (zigp_clock_stamp); see clock_mhz"""
        out = np.zeros(24, dtype=np.int64)
        _check(self.lib, self.ctx, self.lib.zigp_clock_stamp(self.ctx, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return {int(out[3 * i]): (int(out[3 * i + 1]), int(out[3 * i + 2])) for i in range(8)}

    @staticmethod
    def clock_mhz(a, b):
        """sustained shader clock (MHz) between two clock_stamp() results: median over the XCDs both stamps saw; None if the counters
        do not behave as a core-clock / constant-clock pair on this device"""
        vals = []
        for x in a:
            # a region shorter than 20 ms of the constant 100 MHz counter says nothing about a SUSTAINED clock (a 4 ms region once read 3.8 GHz
            # on a chip that tops out at 2.4: the two counters are sampled by a one-wave kernel a few microseconds apart)
            if x in b and b[x][1] - a[x][1] >= 2000000:
                vals.append((b[x][0] - a[x][0]) / ((b[x][1] - a[x][1]) / 1e8) / 1e6)
        if not vals:
            return None
        v = float(np.median(vals))
        return v if 300.0 < v < 3000.0 else None

    def profile_sampling(self, every=8):
        """every=1 times every launch of the chunk loop (exact sums); n > 1 samples every n-th full-size chunk"""
        _check(self.lib, self.ctx, self.lib.zigp_profile_sampling(self.ctx, int(every)))

    def profile_reset(self):
        _check(self.lib, self.ctx, self.lib.zigp_profile_reset(self.ctx))

    def profile_get(self):
        ms = np.zeros(_lib.NCLASS)
        n = np.zeros(_lib.NCLASS, dtype=np.int64)
        fl = np.zeros(_lib.NCLASS)
        tot = np.zeros(_lib.NCLASS, dtype=np.int64)
        _check(self.lib, self.ctx, self.lib.zigp_profile_get(self.ctx, ptr(ms), n.ctypes.data_as(C.POINTER(C.c_int64)), ptr(fl)))
        _check(self.lib, self.ctx, self.lib.zigp_profile_totals(self.ctx, tot.ctypes.data_as(C.POINTER(C.c_int64))))
        # ms / launches / flops: the TIMED (sampled) launches; total_launches: all of them; est_total_ms = avg * total
        return {name: dict(ms=float(ms[i]), launches=int(n[i]), flops=float(fl[i]), total_launches=int(tot[i]),
                           est_total_ms=float(ms[i]) / max(int(n[i]), 1) * int(tot[i]))
                for i, name in enumerate(_lib.PROF_CLASSES)}

    # ---- diagnostics ----
    def test_gemm(self, A, B, transA=False, transB=False):
        A, B = as_f64(A), as_f64(B)
        m, k = (A.shape[1], A.shape[0]) if transA else A.shape
        n = B.shape[0] if transB else B.shape[1]
        out = np.zeros((m, n))
        _check(self.lib, self.ctx, self.lib.zigp_test_gemm(self.ctx, int(transA), int(transB), m, n, k, ptr(A), ptr(B), ptr(out)))
        return out

    def test_kuf(self, X, Z, ell, var):
        """K (M,N) = kern.K(Z, X) as the chunk loop's k_kuf_build writes a Kuf panel (hand-written exponential)."""
        X, Z = as_f64(X), as_f64(Z)
        if X.ndim == 1: X = X[:, None]
        if Z.ndim == 1: Z = Z[:, None]
        N, D = X.shape
        M = Z.shape[0]
        ell = as_f64(np.broadcast_to(np.asarray(ell, dtype=np.float64), (D,)))
        out = np.zeros((M, N))
        _check(self.lib, self.ctx, self.lib.zigp_test_kuf(self.ctx, N, M, D, ptr(X), ptr(Z), ptr(ell), float(var), ptr(out)))
        return out

    def test_potrf_trtri(self, A, split_k=False):
        A = as_f64(A)
        n = A.shape[0]
        L, W = np.zeros((n, n)), np.zeros((n, n))
        _check(self.lib, self.ctx, self.lib.zigp_test_potrf_trtri(self.ctx, n, ptr(A), ptr(L), ptr(W), int(bool(split_k))))
        return L, W


def reference_engine(device=0):
    """The engine of the reference look-alikes (onoffgpf.OnOffSVGP, onofftf.onoff / predict_onoff, the likelihood heads): results identical
    to the reference on the same inputs includes WHERE it fails -- tf.cholesky raises on a non-positive pivot only (onofftf/main.py:200,
    268,355), so these run with zigp_set_pivot_rtol(ctx, 0).  The stricter default of the bare engine (8 eps (variance + jitter)) stays
    one call away: engine.set_pivot_rtol(8)."""
    return DenseEngine(device, pivot_rtol=0.0)
