"""ctypes binding of libpcr.so (include/pcr.h) -- the only way Python reaches the device code.

There is NO CPU fallback: if the shared library is missing or no gfx950 device is visible the
calls raise.  torch is imported first on purpose: libpcr.so's DT_NEEDED `libamdhip64.so.7` then
resolves to the HIP runtime torch already loaded, so device pointers and streams are shared
between torch tensors and this library (one runtime per process).
"""
import ctypes as C
import weakref
import os

import numpy as np
import torch  # noqa: F401  (must precede loading libpcr.so, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCR_LIB: developer aid for A/B runs of differently built variants of the library (never set in tests or by the driver)
LIB_PATH = os.environ.get("PCR_LIB") or os.path.join(_HERE, "libpcr.so")

NO_INDEX = 0xFFFFFFFF
MAX_K = 32
ORDER_ASCENDING, ORDER_DESCENDING = 0, 1
INDEX_AUTO, INDEX_GRID, INDEX_BRUTE = 0, 1, 2
STEP_EXACT_INDICES = 1

_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint32)
_bp = C.POINTER(C.c_uint8)
_vp = C.c_void_p


class PcrError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        super().__init__(what)


class IndexParams(C.Structure):
    _fields_ = [("kind", C.c_int), ("target_occupancy", C.c_double), ("cell_size", C.c_double),
                ("max_cells", C.c_uint64), ("brute_max_points", C.c_uint32), ("max_levels", C.c_uint32),
                ("compact_tables", C.c_uint32)]


class IndexInfo(C.Structure):
    _fields_ = [("kind", C.c_int), ("n", C.c_uint32), ("dims", C.c_uint32 * 3), ("cells", C.c_uint64),
                ("occupied_cells", C.c_uint64), ("cell_size", C.c_double), ("mean_occupancy", C.c_double),
                ("max_occupancy", C.c_uint32), ("levels", C.c_uint32), ("device_bytes", C.c_uint64)]


# every symbol include/pcr.h declares: (restype, argtypes)
SYMBOLS = {
    "pcr_strerror": (C.c_char_p, [C.c_int]),
    "pcr_last_error": (C.c_char_p, []),
    "pcr_version": (C.c_char_p, []),
    "pcr_tune": (C.c_int, [C.c_int, C.c_int]),
    "pcr_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "pcr_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "pcr_ctx_set_profiling": (C.c_int, [_vp, C.c_int]),
    "pcr_ctx_read_kernel_times": (C.c_int, [_vp, _dp, _up]),
    "pcr_ctx_destroy": (C.c_int, [_vp]),
    "pcr_ctx_synchronize": (C.c_int, [_vp]),
    "pcr_cloud_upload": (C.c_int, [_vp, _dp, C.c_uint32, C.POINTER(_vp)]),
    "pcr_cloud_wrap_d": (C.c_int, [_vp, _vp, C.c_uint32, C.POINTER(_vp)]),
    "pcr_cloud_free": (C.c_int, [_vp]),
    "pcr_cloud_size": (C.c_uint32, [_vp]),
    "pcr_cloud_bounds": (C.c_int, [_vp, _dp, _dp]),
    "pcr_index_build": (C.c_int, [_vp, _vp, C.POINTER(IndexParams), C.POINTER(_vp)]),
    "pcr_index_free": (C.c_int, [_vp]),
    "pcr_index_get_info": (C.c_int, [_vp, C.POINTER(IndexInfo)]),
    "pcr_knn": (C.c_int, [_vp, _vp, _dp, C.c_uint32, C.c_uint32, C.c_int, C.c_double, _up, _dp]),
    "pcr_knn_d": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_int, C.c_double, _vp, _vp]),
    "pcr_unique_sq_dist": (C.c_int, [_vp, _vp, _dp]),
    "pcr_sample_random": (C.c_int, [_vp, C.c_uint32, C.c_double, C.c_uint64, C.c_uint32, _up, _up]),
    "pcr_sample_random_draws": (C.c_int, [_vp, C.c_uint32, C.c_double, C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, _up, _up]),
    "pcr_sample_grid": (C.c_int, [_vp, _vp, C.c_uint32, _up, _up]),
    "pcr_icp_step": (C.c_int, [_vp, _vp, _vp, _up, C.c_uint32, _dp, _dp, C.c_uint32, C.c_double, C.c_double,
                               C.c_uint32, _dp, _up, _up, _bp]),
    "pcr_icp_step_d": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint32, _dp, _dp, C.c_uint32, C.c_double,
                                 C.c_double, C.c_uint32, _vp, _vp, _vp, _vp]),
    "pcr_icp_step_noalloc_d": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint32, _dp, _dp, C.c_uint32, C.c_double,
                                         C.c_double, C.c_uint32, _vp, _vp, _vp, _vp]),
    "pcr_noalloc_debug_poison_tickets": (C.c_int, [_vp]),
    "pcr_last_step_kernel": (C.c_char_p, []),
    "pcr_icp_pivots": (C.c_int, [_vp, _vp, _dp, _dp, _dp, _dp]),
    "pcr_fit_create": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, _vp,
                                 C.c_double, C.c_uint32, C.POINTER(_vp)]),
    "pcr_fit_set_samples": (C.c_int, [_vp, _up, C.c_uint32]),
    "pcr_fit_set_debug_outputs": (C.c_int, [_vp, _vp, _vp, _vp]),
    "pcr_fit_reset": (C.c_int, [_vp, _dp, _dp]),
    "pcr_fit_enqueue_step": (C.c_int, [_vp]),
    "pcr_fit_partials_d": (_vp, [_vp]),
    "pcr_fit_enqueue_solve": (C.c_int, [_vp]),
    "pcr_ctx_read_kernel_launches": (C.c_int, [_vp, C.c_int, _dp, C.c_uint32, _up]),
    "pcr_fit_enqueue_iteration": (C.c_int, [_vp]),
    "pcr_fit_enqueue_iterations": (C.c_int, [_vp, C.c_uint32]),
    "pcr_fit_batch_create": (C.c_int, [_vp, C.POINTER(_vp), C.c_uint32, C.POINTER(_vp)]),
    "pcr_fit_batch_enqueue_iteration": (C.c_int, [_vp]),
    "pcr_fit_batch_destroy": (C.c_int, [_vp]),
    "pcr_fit_read": (C.c_int, [_vp, _dp, _dp, _up, C.POINTER(C.c_int), _dp, _dp, _dp, C.c_uint32]),
    "pcr_fit_destroy": (C.c_int, [_vp]),
    "pcr_comm_unique_id": (C.c_int, [C.c_char_p]),
    "pcr_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p, C.POINTER(_vp)]),
    "pcr_allreduce_partials": (C.c_int, [_vp, _vp]),
    "pcr_comm_destroy": (C.c_int, [_vp]),
    "pcr_reduce_pairs": (C.c_int, [_vp, _dp, _dp, C.c_uint32, _dp, _dp, _dp]),
    "pcr_solve_rigid": (C.c_int, [_dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "pcr_reject_centroid_dist": (C.c_int, [_vp, _vp, _vp, _dp, _dp, C.c_uint32, C.c_uint32, C.c_double, C.c_double,
                                           _bp, _dp, _dp]),
    "pcr_reject_centroid_dist_sharded": (C.c_int, [_vp, _vp, _vp, _dp, _dp, C.c_uint32, C.c_uint32, C.c_double,
                                                   C.c_double, _bp, _dp, _dp, _vp, _vp]),
    "pcr_reduce17_with_comm": (C.c_int, [_vp, _vp]),
    "pcr_cloud_set_normals": (C.c_int, [_vp, _dp]),
    "pcr_cloud_set_normals_d": (C.c_int, [_vp, _vp]),
    "pcr_icp_step_normal_fit": (C.c_int, [_vp, _vp, _vp, _up, C.c_uint32, _dp, _dp, C.c_uint32, C.c_double, C.c_double,
                                          C.c_int, _dp, _up, _bp]),
}
REDUCE17_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)  # pcr_reduce17_fn(user, buf17_d)

_lib = None


def lib():
    """Load libpcr.so (raises if it has not been built -- there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        L = lib()
        raise PcrError(rc, f"{what}: {L.pcr_strerror(rc).decode()} -- {L.pcr_last_error().decode()}")


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _u(a):
    return None if a is None else a.ctypes.data_as(_up)


def _b(a):
    return None if a is None else a.ctypes.data_as(_bp)


def _f64(a, cols=3):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(-1, cols) if cols else a


def _rt(R, T):
    return (np.ascontiguousarray(R, np.float64).reshape(9).copy(),
            np.ascontiguousarray(T, np.float64).reshape(3).copy())


class Context:
    """pcr_ctx: one device + one stream.  stream: None (the context's own stream) or a torch.cuda.Stream
    (the default stream included) on which the _d entry points are then ordered."""

    def __init__(self, device=0, stream=None):
        self.h = _vp()
        _check(lib().pcr_ctx_create(device, C.byref(self.h)), "pcr_ctx_create")
        self.device = device
        self._children = weakref.WeakSet()  # fits, indices, clouds: destroyed before the context they point into
        if stream is not None:
            _check(lib().pcr_ctx_set_stream(self.h, _vp(stream.cuda_stream)), "pcr_ctx_set_stream")

    def set_profiling(self, on, classes=None, stride=1):
        """classes: iterable of kernel classes (0 transform, 1 step, 2 fold, 3 knn) to bracket; None = all.
        stride > 1: only every stride-th launch of a class is bracketed (the first one included)."""
        mask = 0xf if classes is None else sum(1 << c for c in classes)
        v = 0 if not on else (1 if classes is None and stride <= 1 else (mask << 1) | ((stride & 0xff) << 8))
        _check(lib().pcr_ctx_set_profiling(self.h, v), "pcr_ctx_set_profiling")

    def read_kernel_times(self):
        """{class: (sum_ms, launches)} since the previous read; synchronises the stream."""
        ms = np.zeros(4)
        cnt = np.zeros(4, np.uint32)
        _check(lib().pcr_ctx_read_kernel_times(self.h, _d(ms), _u(cnt)), "pcr_ctx_read_kernel_times")
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(("transform", "step", "fold", "knn"))}

    def read_kernel_launches(self, cls):
        """ms of every bracketed launch of one class since the previous read_kernel_times, in launch order."""
        ms = np.zeros(1024)
        cnt = C.c_uint32(0)
        _check(lib().pcr_ctx_read_kernel_launches(self.h, cls, _d(ms), 1024, C.byref(cnt)), "pcr_ctx_read_kernel_launches")
        return ms[:min(cnt.value, 1024)].copy()

    def synchronize(self):
        _check(lib().pcr_ctx_synchronize(self.h), "pcr_ctx_synchronize")

    def close(self):
        if self.h:
            for cls in (Comm, FitBatch, Fit, Index, Cloud):  # dependants first, in dependency order
                for child in [c for c in self._children if isinstance(c, cls)]:
                    child.close()
            lib().pcr_ctx_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Cloud:
    """pcr_cloud.  `xyz` is a numpy (n,3) array (copied to the device) or a CUDA torch tensor of
    float64 (n,3) (borrowed: kept alive by this object)."""

    def __init__(self, ctx, xyz):
        self.ctx = ctx
        self.h = _vp()
        ctx._children.add(self)
        if isinstance(xyz, torch.Tensor):
            assert xyz.is_cuda and xyz.dtype == torch.float64 and xyz.is_contiguous()
            self._keep = xyz
            self.n = xyz.numel() // 3
            _check(lib().pcr_cloud_wrap_d(ctx.h, _vp(xyz.data_ptr()), self.n, C.byref(self.h)), "pcr_cloud_wrap_d")
        else:
            a = _f64(xyz)
            self.n = len(a)
            _check(lib().pcr_cloud_upload(ctx.h, _d(a), self.n, C.byref(self.h)), "pcr_cloud_upload")

    def set_normals(self, normals):
        """Attach per-point normals (numpy (n,3), copied; None detaches): what icp_step_normal_fit reads."""
        a = None if normals is None else _f64(normals)
        assert a is None or len(a) == self.n
        _check(lib().pcr_cloud_set_normals(self.h, _d(a)), "pcr_cloud_set_normals")

    def bounds(self):
        lo, hi = np.empty(3), np.empty(3)
        _check(lib().pcr_cloud_bounds(self.h, _d(lo), _d(hi)), "pcr_cloud_bounds")
        return lo, hi

    def close(self):
        if self.h:
            lib().pcr_cloud_free(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Index:
    """pcr_index over a Cloud."""

    def __init__(self, ctx, cloud, kind=INDEX_AUTO, target_occupancy=0.0, cell_size=0.0, max_cells=0,
                 brute_max_points=0, max_levels=0, compact_tables=0):
        self.ctx, self.cloud = ctx, cloud
        self.h = _vp()
        ctx._children.add(self)
        prm = IndexParams(kind, target_occupancy, cell_size, max_cells, brute_max_points, max_levels, compact_tables)
        _check(lib().pcr_index_build(ctx.h, cloud.h, C.byref(prm), C.byref(self.h)), "pcr_index_build")
        self.n = cloud.n

    def info(self):
        inf = IndexInfo()
        _check(lib().pcr_index_get_info(self.h, C.byref(inf)), "pcr_index_get_info")
        return {f: (list(getattr(inf, f)) if f == "dims" else getattr(inf, f)) for f, _ in IndexInfo._fields_}

    def knn(self, q, k, order=ORDER_ASCENDING, max_d2=-1.0, want_d2=False):
        q = _f64(q)
        idx = np.empty((len(q), k), np.uint32)
        d2 = np.empty((len(q), k), np.float64) if want_d2 else None
        _check(lib().pcr_knn(self.ctx.h, self.h, _d(q), len(q), k, order, max_d2, _u(idx), _d(d2)), "pcr_knn")
        return (idx, d2) if want_d2 else idx

    def knn_d(self, q_t, k, out_idx_t, out_d2_t=None, order=ORDER_ASCENDING, max_d2=-1.0):
        """Device-resident variant: q_t (nq,3) float64 CUDA tensor, out_idx_t (nq,k) int32/uint32 CUDA tensor."""
        nq = q_t.shape[0]
        _check(lib().pcr_knn_d(self.ctx.h, self.h, _vp(q_t.data_ptr()), nq, k, order, max_d2,
                               _vp(out_idx_t.data_ptr()), _vp(out_d2_t.data_ptr()) if out_d2_t is not None else None),
               "pcr_knn_d")

    def unique_sq_dist(self):
        out = C.c_double(0)
        _check(lib().pcr_unique_sq_dist(self.ctx.h, self.h, C.cast(C.byref(out), _dp)), "pcr_unique_sq_dist")
        return out.value

    def close(self):
        if self.h:
            lib().pcr_index_free(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def icp_step(ctx, src_index, tgt_index, R, T, k, thr_fwd_sq, thr_rev_sq, sample_idx=None, exact=True,
             want_outputs=True, flags=0):
    R, T = _rt(R, T)
    if sample_idx is not None:
        sample_idx = np.ascontiguousarray(sample_idx, np.uint32)
        m = len(sample_idx)
    else:
        m = src_index.n
    partials = np.empty(17)
    fwd = np.empty(m, np.uint32) if want_outputs else None
    rev = np.empty(m, np.uint32) if want_outputs else None
    keep = np.empty(m, np.uint8) if want_outputs else None
    _check(lib().pcr_icp_step(ctx.h, src_index.h, tgt_index.h, _u(sample_idx), m, _d(R), _d(T), k, thr_fwd_sq,
                              thr_rev_sq, (STEP_EXACT_INDICES if exact else 0) | flags, _d(partials), _u(fwd), _u(rev),
                              _b(keep)), "pcr_icp_step")
    return dict(partials=partials, fwd=fwd, rev=rev, keep=keep)


def icp_step_normal_fit(ctx, src_index, tgt_index, R, T, k, nvdt=0.0125, ndt=0.85, compare_normals=False, sample_idx=None):
    """f4 (opt-in): pcr_icp_step_normal_fit; the clouds of both indices carry normals (Cloud.set_normals)."""
    R, T = _rt(R, T)
    if sample_idx is not None:
        sample_idx = np.ascontiguousarray(sample_idx, np.uint32)
        m = len(sample_idx)
    else:
        m = src_index.n
    partials = np.empty(17)
    tgt = np.empty(m, np.uint32)
    keep = np.empty(m, np.uint8)
    _check(lib().pcr_icp_step_normal_fit(ctx.h, src_index.h, tgt_index.h, _u(sample_idx), m, _d(R), _d(T), k, nvdt, ndt,
                                         1 if compare_normals else 0, _d(partials), _u(tgt), _b(keep)),
           "pcr_icp_step_normal_fit")
    return dict(partials=partials, tgt=tgt, keep=keep)


NOALLOC_MAX_SAMPLES = 4194304


def icp_step_d(ctx, src_index, tgt_index, R, T, k, thr_fwd_sq, thr_rev_sq, partials_t, sample_begin=0, m=None,
               sample_idx_t=None, exact=False, out_fwd_t=None, out_rev_t=None, out_keep_t=None, flags=0, noalloc=False):
    """Asynchronous, device-resident step: partials_t is a float64 CUDA tensor of >= 17 elements.
    noalloc: go through pcr_icp_step_noalloc_d (fixed library scratch, no allocation inside the call)."""
    R, T = _rt(R, T)
    if m is None:
        m = sample_idx_t.numel() if sample_idx_t is not None else src_index.n - sample_begin

    def p(t):
        return _vp(t.data_ptr()) if t is not None else None

    fn = lib().pcr_icp_step_noalloc_d if noalloc else lib().pcr_icp_step_d
    _check(fn(ctx.h, src_index.h, tgt_index.h, p(sample_idx_t), sample_begin, m, _d(R), _d(T), k,
              thr_fwd_sq, thr_rev_sq, (STEP_EXACT_INDICES if exact else 0) | flags, p(partials_t),
              p(out_fwd_t), p(out_rev_t), p(out_keep_t)), "pcr_icp_step_noalloc_d" if noalloc else "pcr_icp_step_d")


def last_step_kernel():
    return lib().pcr_last_step_kernel().decode()


def icp_pivots(src_index, tgt_index, R, T):
    R, T = _rt(R, T)
    pp, pq = np.empty(3), np.empty(3)
    _check(lib().pcr_icp_pivots(src_index.h, tgt_index.h, _d(R), _d(T), _d(pp), _d(pq)), "pcr_icp_pivots")
    return pp, pq


def sample_random(ctx, n, ratio, seed=0):
    out = np.empty(max(n, 1), np.uint32)
    cnt = C.c_uint32(0)
    _check(lib().pcr_sample_random(ctx.h, n, ratio, seed, n, _u(out), C.byref(cnt)), "pcr_sample_random")
    return out[:cnt.value].copy()


def sample_random_draws(ctx, ratio, draws, rand_max=32767):
    """CSamplerRandom's rule on the caller's rand() stream (one draw per point)."""
    d = np.ascontiguousarray(draws, np.int32)
    out = np.empty(len(d), np.uint32)
    cnt = C.c_uint32(0)
    _check(lib().pcr_sample_random_draws(ctx.h, len(d), ratio, d.ctypes.data_as(C.POINTER(C.c_int32)), rand_max, len(d),
                                         _u(out), C.byref(cnt)), "pcr_sample_random_draws")
    return out[:cnt.value].copy()


def sample_grid(ctx, index):
    out = np.empty(max(index.n, 1), np.uint32)
    cnt = C.c_uint32(0)
    _check(lib().pcr_sample_grid(ctx.h, index.h, index.n, _u(out), C.byref(cnt)), "pcr_sample_grid")
    return out[:cnt.value].copy()


class ConvergenceParams(C.Structure):
    """pcr_convergence_params; regcfg() = RegCfg.txt:5-11 over the defaults of DefaultConvergenceCriterial.cpp:28-33."""
    _fields_ = [("max_iterations", C.c_uint32), ("failure_after_max_iterations", C.c_int),
                ("rotation_threshold", C.c_double), ("translation_threshold", C.c_double),
                ("abs_mse", C.c_double), ("rel_mse", C.c_double), ("max_similar", C.c_uint32)]

    @classmethod
    def regcfg(cls, max_iterations=30):
        return cls(max_iterations, 0, 1e-6, 1e-7, 0.01, 0.01, 0)


FIT_STATES = {0: "NotConverged", 1: "Criteria_Iterations", 2: "Criteria_Transform", 3: "Criteria_AbsMSE",
              4: "Criteria_RelMSE", 5: "Criteria_NoCorrespondeces", -2: "Degenerate"}


class Fit:
    """pcr_fit: the device-resident loop.  step() and solve() only enqueue; read() synchronises."""

    def __init__(self, ctx, src_index, tgt_index, k, thr_fwd_sq, thr_rev_sq, conv=None, unique_sq_dist=0.0,
                 sample_idx_t=None, sample_begin=0, m=None, traj_cap=64):
        self.ctx, self.src, self.tgt = ctx, src_index, tgt_index
        self._keep = sample_idx_t
        ctx._children.add(self)
        if m is None:
            m = sample_idx_t.numel() if sample_idx_t is not None else src_index.n - sample_begin
        self.traj_cap = traj_cap
        self.h = _vp()
        _check(lib().pcr_fit_create(ctx.h, src_index.h, tgt_index.h,
                                    _vp(sample_idx_t.data_ptr()) if sample_idx_t is not None else None, sample_begin, m, k,
                                    thr_fwd_sq, thr_rev_sq, C.cast(C.byref(conv), _vp) if conv is not None else None,
                                    unique_sq_dist, traj_cap, C.byref(self.h)), "pcr_fit_create")

    def set_samples(self, sample_idx):
        a = np.ascontiguousarray(sample_idx, np.uint32)
        _check(lib().pcr_fit_set_samples(self.h, _u(a), len(a)), "pcr_fit_set_samples")

    def set_debug_outputs(self, out_fwd_t=None, out_rev_t=None, out_keep_t=None):
        """parity runs: every later iteration also writes fwd/rev (uint32) and keep (uint8) per sample into these CUDA
        tensors (slot convention of icp_step_d); all None switches them off.  The tensors must outlive their use."""
        self._dbg = (out_fwd_t, out_rev_t, out_keep_t)

        def p(t):
            return _vp(t.data_ptr()) if t is not None else None
        _check(lib().pcr_fit_set_debug_outputs(self.h, p(out_fwd_t), p(out_rev_t), p(out_keep_t)),
               "pcr_fit_set_debug_outputs")

    def reset(self, R, T):
        R, T = _rt(R, T)
        _check(lib().pcr_fit_reset(self.h, _d(R), _d(T)), "pcr_fit_reset")

    def step(self):
        _check(lib().pcr_fit_enqueue_step(self.h), "pcr_fit_enqueue_step")

    def solve(self):
        _check(lib().pcr_fit_enqueue_solve(self.h), "pcr_fit_enqueue_solve")

    def iterate(self):
        """step + solve with nothing in between: one launch per iteration where the filtered kernel applies."""
        _check(lib().pcr_fit_enqueue_iteration(self.h), "pcr_fit_enqueue_iteration")

    def iterate_many(self, count):
        """`count` iterations as one HIP graph launch (captured on first use with that count)."""
        _check(lib().pcr_fit_enqueue_iterations(self.h, count), "pcr_fit_enqueue_iterations")

    def partials_tensor(self):
        """The 17 partial sums as a torch view of the fit's device buffer (for torch.distributed.all_reduce)."""
        import torch
        ptr = lib().pcr_fit_partials_d(self.h)

        class _Mem:
            __cuda_array_interface__ = {"shape": (17,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(_Mem(), device=f"cuda:{self.ctx.device}")

    def read(self):
        R, T = np.empty(9), np.empty(3)
        it, st = C.c_uint32(0), C.c_int(0)
        mse, pairs = C.c_double(0), C.c_double(0)
        traj = np.zeros((max(self.traj_cap, 1), 14))
        _check(lib().pcr_fit_read(self.h, _d(R), _d(T), C.byref(it), C.byref(st), C.cast(C.byref(mse), _dp),
                                  C.cast(C.byref(pairs), _dp), _d(traj), self.traj_cap), "pcr_fit_read")
        return dict(R=R, T=T, iterations=it.value, state=st.value, msg=FIT_STATES.get(st.value, "?"), mse=mse.value,
                    pairs=pairs.value, traj=traj[:min(it.value, self.traj_cap)])

    def run(self, R, T, max_steps, all_reduce=None):
        """Enqueue up to max_steps iterations from (R,T) and read the result (one synchronisation)."""
        self.reset(R, T)
        part = self.partials_tensor() if all_reduce is not None else None
        for _ in range(max_steps):
            if all_reduce is None:
                self.iterate()
                continue
            self.step()
            all_reduce(part)
            self.solve()
        return self.read()

    def close(self):
        if self.h:
            lib().pcr_fit_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FitBatch:
    """pcr_fit_batch: fits of one context advanced together, one kernel launch per iteration."""

    def __init__(self, ctx, fits):
        self.ctx, self.fits = ctx, list(fits)
        self.h = _vp()
        ctx._children.add(self)
        arr = (_vp * len(self.fits))(*[f.h for f in self.fits])
        _check(lib().pcr_fit_batch_create(ctx.h, arr, len(self.fits), C.byref(self.h)), "pcr_fit_batch_create")

    def iterate(self):
        _check(lib().pcr_fit_batch_enqueue_iteration(self.h), "pcr_fit_batch_enqueue_iteration")

    def close(self):
        if self.h:
            lib().pcr_fit_batch_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """pcr_comm: RCCL communicator bound to a context's stream (one process per GPU)."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().pcr_comm_unique_id(buf), "pcr_comm_unique_id")
        return buf.raw

    def __init__(self, ctx, rank, world, uid):
        self.ctx = ctx
        self.h = _vp()
        ctx._children.add(self)
        _check(lib().pcr_comm_create(ctx.h, rank, world, uid, C.byref(self.h)), "pcr_comm_create")

    def allreduce_partials(self, ptr):
        """ptr: device address (int) or a CUDA tensor of >= 17 float64."""
        addr = ptr.data_ptr() if hasattr(ptr, "data_ptr") else int(ptr)
        _check(lib().pcr_allreduce_partials(self.h, _vp(addr)), "pcr_allreduce_partials")

    def close(self):
        if self.h:
            lib().pcr_comm_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reduce_pairs(ctx, pair_p, pair_q, pivot_p=None, pivot_q=None):
    """The 17 partial sums of an explicit pair list, reduced on the device.  Returns (partials, pivot_p, pivot_q)."""
    pp, pq = _f64(pair_p), _f64(pair_q)
    assert pp.shape == pq.shape
    pivot_p = np.ascontiguousarray(pp[0] if pivot_p is None and len(pp) else (pivot_p if pivot_p is not None else np.zeros(3)), np.float64).reshape(3).copy()
    pivot_q = np.ascontiguousarray(pq[0] if pivot_q is None and len(pq) else (pivot_q if pivot_q is not None else np.zeros(3)), np.float64).reshape(3).copy()
    out = np.empty(17)
    _check(lib().pcr_reduce_pairs(ctx.h, _d(pp), _d(pq), len(pp), _d(pivot_p), _d(pivot_q), _d(out)), "pcr_reduce_pairs")
    return out, pivot_p, pivot_q


def solve_rigid(partials, pivot_p, pivot_q, R, T):
    """Returns (R', T', Rc, Tc, mse) -- host-only (3x3 SVD + composition)."""
    R, T = _rt(R, T)
    partials = np.ascontiguousarray(partials, np.float64).reshape(17)
    pivot_p = np.ascontiguousarray(pivot_p, np.float64).reshape(3)
    pivot_q = np.ascontiguousarray(pivot_q, np.float64).reshape(3)
    Rc, Tc = np.empty(9), np.empty(3)
    mse = C.c_double(0)
    _check(lib().pcr_solve_rigid(_d(partials), _d(pivot_p), _d(pivot_q), _d(R), _d(T), _d(Rc), _d(Tc),
                                 C.cast(C.byref(mse), _dp)), "pcr_solve_rigid")
    return R, T, Rc, Tc, mse.value


def reject_centroid_dist(ctx, src_index, tgt_index, pair_p, pair_q, msn=25, msd=8.0, cdf=1.0, reduce=None):
    """reduce: None (all pairs are here), a Comm (pairs sharded over its ranks: the library's in-stream RCCL
    all-reduce), or a callable(ptr) that sums the 17-double device buffer at `ptr` over all ranks in place, ordered
    on the context's stream (e.g. torch.distributed on a tensor wrapping it) and returns None/0 or a negative status."""
    pp, pq = _f64(pair_p), _f64(pair_q)
    P = len(pp)
    keep = np.empty(P, np.uint8)
    d = np.empty(P)
    stat = np.empty(2)
    if reduce is None:
        _check(lib().pcr_reject_centroid_dist(ctx.h, src_index.h, tgt_index.h, _d(pp), _d(pq), P, msn, msd, cdf,
                                              _b(keep), _d(d), _d(stat)), "pcr_reject_centroid_dist")
    else:
        if isinstance(reduce, Comm):
            fn, user = C.cast(lib().pcr_reduce17_with_comm, _vp), reduce.h
        else:
            def hook(_user, ptr, _f=reduce):
                try:
                    return int(_f(ptr) or 0)
                except Exception:  # an exception cannot cross the C frame: report a failure status instead
                    import traceback
                    traceback.print_exc()
                    return -1
            cb = REDUCE17_FN(hook)
            fn, user = C.cast(cb, _vp), None
        _check(lib().pcr_reject_centroid_dist_sharded(ctx.h, src_index.h, tgt_index.h, _d(pp), _d(pq), P, msn, msd, cdf,
                                                      _b(keep), _d(d), _d(stat), fn, user),
               "pcr_reject_centroid_dist_sharded")
    return dict(keep=keep, d=d, mean=stat[0], sigma=stat[1], kept=int(keep.sum()))
