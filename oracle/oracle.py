"""ctypes bindings of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: the checker, never the thing measured or shipped.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NO_INDEX = 0xFFFFFFFF

_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint32)
_bp = C.POINTER(C.c_uint8)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _u(a):
    return None if a is None else a.ctypes.data_as(_up)


def _b(a):
    return None if a is None else a.ctypes.data_as(_bp)


def _f64(a, cols=3):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if cols:
        a = a.reshape(-1, cols)
    return a


def build():
    """(Re)build liboracle.so."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


class ConvParams(C.Structure):
    _fields_ = [("max_iterations", C.c_uint32), ("failure_after_max_iter", C.c_int),
                ("rotation_threshold", C.c_double), ("translation_threshold", C.c_double),
                ("rel_mse", C.c_double), ("abs_mse", C.c_double), ("max_similar", C.c_uint32)]

    @classmethod
    def regcfg(cls, max_iterations=30):
        # RegCfg.txt:5-11 over the defaults of DefaultConvergenceCriterial.cpp:28-33
        return cls(max_iterations, 0, 1e-6, 1e-7, 0.01, 0.01, 0)


class FitParams(C.Structure):
    _fields_ = [("k", C.c_uint32), ("sdft", C.c_double), ("ndft", C.c_double), ("conv", ConvParams)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_kd_old_build.restype = C.c_void_p
        L.orc_kd_old_build.argtypes = [_dp, C.c_uint32, C.c_uint32]
        L.orc_kd_old_query.argtypes = [C.c_void_p, _dp, C.c_uint32, _up, _dp, C.c_int]
        L.orc_kd_old_free.argtypes = [C.c_void_p]
        L.orc_kd_new_build.restype = C.c_void_p
        L.orc_kd_new_build.argtypes = [_dp, C.c_uint32, C.c_uint32, C.c_int]
        L.orc_kd_new_query.argtypes = [C.c_void_p, C.c_uint32, _dp, C.c_uint32, _up]
        L.orc_kd_new_free.argtypes = [C.c_void_p]
        L.orc_transform.argtypes = [_dp, C.c_uint32, _dp, _dp]
        L.orc_uniq_sq_dist.restype = C.c_double
        L.orc_uniq_sq_dist.argtypes = [_dp, C.c_uint32]
        L.orc_icrp_correspondences.restype = C.c_uint32
        L.orc_icrp_correspondences.argtypes = [_dp, C.c_uint32, _dp, C.c_uint32, C.c_void_p, _up, C.c_uint32,
                                               _dp, _dp, C.c_uint32, C.c_double, C.c_double, _up, _up, _bp,
                                               _dp, _dp, C.c_int]
        L.orc_svd2.restype = C.c_int
        L.orc_svd2.argtypes = [_dp, _dp, C.c_uint32, _dp, _dp, _dp, _dp]
        L.orc_svd3.argtypes = [_dp, _dp, _dp, _dp]
        L.orc_conv_create.restype = C.c_void_p
        L.orc_conv_create.argtypes = [C.POINTER(ConvParams)]
        L.orc_conv_step.restype = C.c_int
        L.orc_conv_step.argtypes = [C.c_void_p, _dp, _dp, C.c_uint32, _dp, _dp, C.c_uint32, C.c_double,
                                    C.POINTER(C.c_int), _dp]
        L.orc_conv_free.argtypes = [C.c_void_p]
        L.orc_fit.restype = C.c_uint32
        L.orc_fit.argtypes = [_dp, C.c_uint32, _dp, C.c_uint32, _up, C.c_uint32, C.POINTER(FitParams), _dp, _dp,
                              C.POINTER(C.c_int), _dp, C.c_uint32, C.c_int]
        L.orc_reject_centroid_dist.restype = C.c_uint32
        L.orc_reject_centroid_dist.argtypes = [_dp, C.c_uint32, _dp, C.c_uint32, _dp, _dp, C.c_uint32, C.c_uint32,
                                               C.c_double, C.c_double, _dp, _bp, _dp]
        L.orc_knn_brute.argtypes = [_dp, C.c_uint32, _dp, C.c_uint32, C.c_uint32, _up, _dp]
        _lib = L
    return _lib


class KdOld:
    """hiveCommon::CKNNSearch restatement (ascending distance, first-visited wins)."""

    def __init__(self, xyz, k):
        self.xyz = _f64(xyz)
        self.k = int(k)
        self.h = lib().orc_kd_old_build(_d(self.xyz), len(self.xyz), self.k)
        if not self.h:
            raise ValueError("orc_kd_old_build failed")

    def query(self, q, want_d2=False, threads=1):
        q = _f64(q)
        idx = np.empty((len(q), self.k), np.uint32)
        d2 = np.empty((len(q), self.k), np.float64) if want_d2 else None
        lib().orc_kd_old_query(self.h, _d(q), len(q), _u(idx), _d(d2), threads)
        return (idx, d2) if want_d2 else idx

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kd_old_free(self.h)
            self.h = None


class KdNew:
    """hiveSearch::CKNNSearch restatement (descending distance per query)."""

    def __init__(self, data, dim=3, leaf_max=10):
        self.data = _f64(data, dim)
        self.dim = dim
        self.h = lib().orc_kd_new_build(_d(self.data), len(self.data), dim, leaf_max)
        if not self.h:
            raise ValueError("orc_kd_new_build failed (degenerate split)")

    def query(self, q, k):
        q = _f64(q, self.dim)
        idx = np.empty((len(q), k), np.uint32)
        lib().orc_kd_new_query(self.h, k, _d(q), len(q), _u(idx))
        return idx

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kd_new_free(self.h)
            self.h = None


def knn_brute(xyz, q, k):
    xyz, q = _f64(xyz), _f64(q)
    idx = np.empty((len(q), k), np.uint32)
    d2 = np.empty((len(q), k), np.float64)
    lib().orc_knn_brute(_d(xyz), len(xyz), _d(q), len(q), k, _u(idx), _d(d2))
    return idx, d2


def transform(xyz, R, T):
    out = _f64(xyz).copy()
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    T = np.ascontiguousarray(T, np.float64).reshape(3)
    lib().orc_transform(_d(out), len(out), _d(R), _d(T))
    return out


def uniq_sq_dist(xyz):
    xyz = _f64(xyz)
    return lib().orc_uniq_sq_dist(_d(xyz), len(xyz))


def icrp_correspondences(src, tgt, tgt_tree, R, T, k, thr_fwd_sq, thr_rev_sq, sample_idx=None, threads=1,
                         want_pairs=True):
    src, tgt = _f64(src), _f64(tgt)
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    T = np.ascontiguousarray(T, np.float64).reshape(3)
    if sample_idx is not None:
        sample_idx = np.ascontiguousarray(sample_idx, np.uint32)
        m = len(sample_idx)
    else:
        m = len(src)
    fwd = np.empty(m, np.uint32)
    rev = np.empty(m, np.uint32)
    keep = np.empty(m, np.uint8)
    pp = np.empty((m, 3), np.float64) if want_pairs else None
    pq = np.empty((m, 3), np.float64) if want_pairs else None
    P = lib().orc_icrp_correspondences(_d(src), len(src), _d(tgt), len(tgt), tgt_tree.h, _u(sample_idx), m, _d(R),
                                       _d(T), k, thr_fwd_sq, thr_rev_sq, _u(fwd), _u(rev), _b(keep), _d(pp), _d(pq),
                                       threads)
    out = dict(fwd=fwd, rev=rev, keep=keep, P=P)
    if want_pairs:
        out["pair_p"], out["pair_q"] = pp[:P], pq[:P]
    return out


def svd3(H):
    """H given as a 3x3 numpy matrix (row, col); returns U, S, V as numpy matrices."""
    Hc = np.ascontiguousarray(np.asarray(H, np.float64).T).reshape(9)  # column-major
    U = np.empty(9)
    S = np.empty(3)
    V = np.empty(9)
    lib().orc_svd3(_d(Hc), _d(U), _d(S), _d(V))
    return U.reshape(3, 3).T.copy(), S, V.reshape(3, 3).T.copy()


def svd2(pair_p, pair_q, R, T):
    """Returns (R', T', Rc, Tc); R/Rc as column-major 9-vectors."""
    pp, pq = _f64(pair_p), _f64(pair_q)
    R = np.array(R, np.float64).reshape(9).copy()
    T = np.array(T, np.float64).reshape(3).copy()
    Rc = np.empty(9)
    Tc = np.empty(3)
    rc = lib().orc_svd2(_d(pp), _d(pq), len(pp), _d(R), _d(T), _d(Rc), _d(Tc))
    if rc != 0:
        raise ZeroDivisionError("empty pair set")
    return R, T, Rc, Tc


class Convergence:
    STATES = ["NotConverged", "Criteria_Iterations", "Criteria_Transform", "Criteria_AbsMSE", "Criteria_RelMSE",
              "Criteria_NoCorrespondeces"]

    def __init__(self, params):
        self.h = lib().orc_conv_create(C.byref(params))

    def step(self, R, T, it, pair_p, pair_q, uniq):
        R = np.ascontiguousarray(R, np.float64).reshape(9)
        T = np.ascontiguousarray(T, np.float64).reshape(3)
        pp, pq = _f64(pair_p), _f64(pair_q)
        st = C.c_int(0)
        mse = C.c_double(0)
        c = lib().orc_conv_step(self.h, _d(R), _d(T), it, _d(pp), _d(pq), len(pp), uniq, C.byref(st),
                                C.cast(C.byref(mse), _dp))
        return bool(c), st.value, mse.value

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_conv_free(self.h)
            self.h = None


def sampler_random(draws, ratio, rand_max=32767):
    d = np.ascontiguousarray(draws, np.int32)
    out = np.empty(len(d), np.uint32)
    fn = lib().orc_sampler_random
    fn.restype = C.c_uint32
    fn.argtypes = [C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.c_double, _up]
    c = fn(d.ctypes.data_as(C.POINTER(C.c_int32)), len(d), rand_max, ratio, _u(out))
    return out[:c].copy()


def icrp_normal_fit(src, src_normals, tgt, tgt_normals, tree, R, T, k, nvdt=0.0125, ndt=0.85, compare_normals=False,
                    sample_idx=None, threads=1):
    """f4 (opt-in): __insertFitPairs as the pair decision; see icrp.cpp."""
    src, tgt, tn = _f64(src), _f64(tgt), _f64(tgt_normals)
    sn = _f64(src_normals) if src_normals is not None else None
    R, T = np.ascontiguousarray(R, np.float64).reshape(9), np.ascontiguousarray(T, np.float64).reshape(3)
    if sample_idx is not None:
        sample_idx = np.ascontiguousarray(sample_idx, np.uint32)
        m = len(sample_idx)
    else:
        m = len(src)
    out = np.empty(m, np.uint32)
    keep = np.empty(m, np.uint8)
    pp, pq = np.empty((m, 3)), np.empty((m, 3))
    fn = lib().orc_icrp_normal_fit
    fn.restype = C.c_uint32
    fn.argtypes = [_dp, _dp, C.c_uint32, _dp, _dp, C.c_uint32, C.c_void_p, _up, C.c_uint32, _dp, _dp, C.c_uint32,
                   C.c_double, C.c_double, C.c_int, _up, _bp, _dp, _dp, C.c_int]
    P = fn(_d(src), _d(sn), len(src), _d(tgt), _d(tn), len(tgt), tree.h, _u(sample_idx), m, _d(R), _d(T), k, nvdt, ndt,
           1 if compare_normals else 0, _u(out), _b(keep), _d(pp), _d(pq), threads)
    return dict(tgt=out, keep=keep, P=P, pair_p=pp[:P], pair_q=pq[:P])


def fit(src, tgt, k=4, sdft=36.0, ndft=4.0, conv=None, R0=None, T0=None, sample_idx=None, threads=1, traj_cap=1024):
    src, tgt = _f64(src), _f64(tgt)
    prm = FitParams(k, sdft, ndft, conv or ConvParams.regcfg())
    R = np.array(R0 if R0 is not None else np.eye(3).T.reshape(9), np.float64).reshape(9).copy()
    T = np.array(T0 if T0 is not None else np.zeros(3), np.float64).reshape(3).copy()
    if sample_idx is not None:
        sample_idx = np.ascontiguousarray(sample_idx, np.uint32)
        m = len(sample_idx)
    else:
        m = len(src)
    traj = np.zeros((traj_cap, 14))
    st = C.c_int(0)
    n = lib().orc_fit(_d(src), len(src), _d(tgt), len(tgt), _u(sample_idx), m, C.byref(prm), _d(R), _d(T),
                      C.byref(st), _d(traj), traj_cap, threads)
    return dict(R=R, T=T, iterations=n, state=st.value, traj=traj[:min(n, traj_cap)])


def reject_centroid_dist(src, tgt, pair_p, pair_q, msn=25, msd=8.0, cdf=1.0):
    src, tgt, pp, pq = _f64(src), _f64(tgt), _f64(pair_p), _f64(pair_q)
    P = len(pp)
    d = np.empty(P)
    keep = np.empty(P, np.uint8)
    stat = np.empty(2)
    n = lib().orc_reject_centroid_dist(_d(src), len(src), _d(tgt), len(tgt), _d(pp), _d(pq), P, msn, msd, cdf,
                                       _d(d), _b(keep), _d(stat))
    return dict(d=d, keep=keep, mean=stat[0], sigma=stat[1], kept=n)


