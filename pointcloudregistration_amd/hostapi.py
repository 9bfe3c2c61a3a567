"""ctypes binding of libpcr_host.so (include/pcr_host.h): the C++ plugin layer that mirrors the reference's
registration interfaces over the device library.  No fallback: a missing library raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpcr_host.so")

_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint32)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, _dp, C.c_void_p)

# name -> (restype, argtypes): must list every symbol include/pcr_host.h declares (tests/test_abi.py checks)
SYMBOLS = {
    "pcrh_reset": (None, []),
    "pcrh_set_device": (None, [C.c_int]),
    "pcrh_set_verbose": (None, [C.c_int]),
    "pcrh_set_compact_tables": (None, [C.c_uint32]),
    "pcrh_set_sharding": (None, [C.c_uint32, C.c_uint32, ALLREDUCE_FN, C.c_void_p]),
    "pcrh_config_parse_file": (C.c_int, [C.c_char_p]),
    "pcrh_config_parse_text": (None, [C.c_char_p]),
    "pcrh_config_set": (None, [C.c_char_p, C.c_char_p]),
    "pcrh_config_get": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32]),
    "pcrh_is_registered": (C.c_int, [C.c_char_p]),
    "pcrh_fit": (C.c_int, [_dp, _dp, C.c_uint32, _dp, _dp, C.c_uint32, _up, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p,
                           C.c_char_p, _dp, _dp, _dp, C.c_uint32, _up, C.c_char_p, C.c_uint32]),
    "pcrh_correspondences": (C.c_int, [_dp, C.c_uint32, _dp, C.c_uint32, _up, C.c_uint32, _dp, _dp, C.c_char_p, _dp, _dp,
                                       C.c_uint32, _up]),
    "pcrh_svd2": (C.c_int, [_dp, _dp, C.c_uint32, _dp, _dp]),
    "pcrh_search": (C.c_int, [_dp, C.c_uint32, C.c_uint32, _dp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, _up]),
    "pcrh_ply_load": (C.c_int, [C.c_char_p, C.c_int, C.c_uint32, _dp, _dp, _up]),
    "pcrh_ply_save": (C.c_int, [C.c_char_p, _dp, _dp, C.c_uint32]),
    "pcrh_register_sequence": (C.c_int, [C.POINTER(_dp), _up, C.c_uint32, _dp, _dp, _dp, _up, C.c_char_p, C.c_uint32]),
    "pcrh_conv_reset": (None, [C.c_double]),
    "pcrh_conv_step": (C.c_int, [_dp, _dp, C.c_uint32, _dp, _dp, C.c_uint32, C.POINTER(C.c_int), _dp, C.c_char_p, C.c_uint32]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _u(a):
    return None if a is None else a.ctypes.data_as(_up)


def _pts(a):
    return None if a is None else np.ascontiguousarray(a, np.float64).reshape(-1, 3)


def _s(x):
    return None if x is None else x.encode()


def reset():
    lib().pcrh_reset()


def set_compact_tables(mode):
    """pcr_index_params.compact_tables for every index the plugin layer builds from now on (0: dense, the default)."""
    lib().pcrh_set_compact_tables(int(mode))


def config_text(text):
    lib().pcrh_config_parse_text(text.encode())


def config_file(path):
    return lib().pcrh_config_parse_file(path.encode()) == 0


def config_set(key, value):
    lib().pcrh_config_set(key.encode(), str(value).encode())


def config_get(key):
    buf = C.create_string_buffer(512)
    return buf.value.decode() if lib().pcrh_config_get(key.encode(), buf, 512) == 0 else None


def is_registered(sig):
    return bool(lib().pcrh_is_registered(sig.encode()))


def fit(src, tgt, R0=None, T0=None, sample_idx=None, sampler=None, estimation=None, rejection=None, transformation=None,
        src_normals=None, tgt_normals=None, traj_cap=1024):
    """CIterativeFit::fit through the plugin layer.  Returns dict(R, T, iterations, msg, traj, status)."""
    src, tgt = _pts(src), _pts(tgt)
    sn, tn = _pts(src_normals), _pts(tgt_normals)
    R = np.eye(3).reshape(9).copy() if R0 is None else np.ascontiguousarray(R0, np.float64).reshape(9).copy()
    T = np.zeros(3) if T0 is None else np.ascontiguousarray(T0, np.float64).reshape(3).copy()
    smp = None if sample_idx is None else np.ascontiguousarray(sample_idx, np.uint32)
    traj = np.zeros((traj_cap, 14))
    iters = C.c_uint32(0)
    msg = C.create_string_buffer(256)
    rc = lib().pcrh_fit(_d(src), _d(sn), len(src), _d(tgt), _d(tn), len(tgt), _u(smp), 0 if smp is None else len(smp),
                        _s(sampler), _s(estimation), _s(rejection), _s(transformation), _d(R), _d(T), _d(traj), traj_cap,
                        C.byref(iters), msg, 256)
    return dict(status=rc, R=R, T=T, iterations=iters.value, msg=msg.value.decode(), traj=traj[:min(iters.value, traj_cap)])


def correspondences(src, tgt, R, T, sample_idx=None, rejection=None):
    """Pair lists (transformed source points, target points) after estimation + rejection."""
    src, tgt = _pts(src), _pts(tgt)
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    T = np.ascontiguousarray(T, np.float64).reshape(3)
    smp = None if sample_idx is None else np.ascontiguousarray(sample_idx, np.uint32)
    cap = len(src) if smp is None else len(smp)
    pp, pq = np.empty((max(cap, 1), 3)), np.empty((max(cap, 1), 3))
    n = C.c_uint32(0)
    rc = lib().pcrh_correspondences(_d(src), len(src), _d(tgt), len(tgt), _u(smp), 0 if smp is None else len(smp), _d(R), _d(T),
                                    _s(rejection), _d(pp), _d(pq), cap, C.byref(n))
    if rc != 0:
        raise RuntimeError(f"pcrh_correspondences failed ({rc})")
    return pp[:n.value].copy(), pq[:n.value].copy()


def svd2(pair_a, pair_b):
    a, b = _pts(pair_a), _pts(pair_b)
    R, T = np.empty(9), np.empty(3)
    rc = lib().pcrh_svd2(_d(a), _d(b), len(a), _d(R), _d(T))
    if rc != 0:
        raise RuntimeError(f"pcrh_svd2 failed ({rc})")
    return R, T


def search(points, queries, k, nearest_first=False, multi_threaded=False, dim=3):
    """CBaseSearch::executeSearch on CGridSearchGPU.  Returns [nq, k] uint32, or None where the reference returns NULL."""
    p = np.ascontiguousarray(points, np.float64).reshape(-1, dim)
    q = np.ascontiguousarray(queries, np.float64).reshape(-1, dim)
    out = np.empty((len(q), k), np.uint32)
    rc = lib().pcrh_search(_d(p), len(p), dim, _d(q), len(q), k, int(nearest_first), int(multi_threaded), _u(out))
    return out if rc == 0 else None


def ply_load(path, with_normals=True):
    """(xyz, normals or None) of an ASCII PLY in the reference's grammar; None if unreadable."""
    n = C.c_uint32(0)
    rc = lib().pcrh_ply_load(path.encode(), int(with_normals), 0, None, None, C.byref(n))
    if rc == -1:
        return None
    xyz = np.empty((n.value, 3))
    nrm = np.empty((n.value, 3)) if with_normals else None
    rc = lib().pcrh_ply_load(path.encode(), int(with_normals), n.value, _d(xyz), _d(nrm), C.byref(n))
    return (xyz, nrm) if rc == 0 else None


def ply_save(path, xyz, normals=None):
    xyz, normals = _pts(xyz), _pts(normals)
    return lib().pcrh_ply_save(path.encode(), _d(xyz), _d(normals), len(xyz)) == 0


def register_sequence(scans, init=None):
    """Sequential pairwise registration (scan i+1 -> scan i) through CRegistartion.  scans: list of (n_i,3) arrays;
    init: optional list of (R9, T3).  Returns dict(relative=[(R,T)], absolute=[(R,T)], iterations, msgs)."""
    scans = [_pts(s) for s in scans]
    k = len(scans)
    ptrs = (_dp * k)(*[_d(s) for s in scans])
    counts = np.array([len(s) for s in scans], np.uint32)
    ini = None
    if init is not None:
        ini = np.concatenate([np.concatenate([np.asarray(R, np.float64).reshape(9), np.asarray(T, np.float64).reshape(3)])
                              for R, T in init])
    rel, ab = np.zeros((k - 1, 12)), np.zeros((k - 1, 12))
    its = np.zeros(k - 1, np.uint32)
    msgs = C.create_string_buffer(64 * (k - 1))
    rc = lib().pcrh_register_sequence(ptrs, _u(counts), k, _d(ini), _d(rel), _d(ab), _u(its), msgs, 64)
    if rc != 0:
        raise RuntimeError(f"pcrh_register_sequence failed ({rc})")
    m = [msgs.raw[i * 64:(i + 1) * 64].split(b"\0")[0].decode() for i in range(k - 1)]
    return dict(relative=[(r[:9].copy(), r[9:].copy()) for r in rel], absolute=[(r[:9].copy(), r[9:].copy()) for r in ab],
                iterations=its, msgs=m)


def conv_reset(unique_sq_dist):
    lib().pcrh_conv_reset(float(unique_sq_dist))


def conv_step(R, T, iteration, pair_p, pair_q):
    """Returns (converged, state, mse, message)."""
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    T = np.ascontiguousarray(T, np.float64).reshape(3)
    pp, pq = _pts(pair_p), _pts(pair_q)
    st, mse = C.c_int(0), C.c_double(0)
    msg = C.create_string_buffer(128)
    r = lib().pcrh_conv_step(_d(R), _d(T), iteration, _d(pp), _d(pq), len(pp), C.byref(st), C.cast(C.byref(mse), _dp), msg, 128)
    return bool(r), st.value, mse.value, msg.value.decode()


def set_sharding(rank, world, fn):
    """fn(numpy view of the 17 partials) -> None, summing them over the ranks in place.  Keep the returned handle alive."""
    def tramp(ptr, _user):
        try:
            fn(np.ctypeslib.as_array(ptr, shape=(17,)))
            return 0
        except Exception:  # noqa: BLE001 -- reported as a failed all-reduce by the C++ layer
            return -1
    cb = ALLREDUCE_FN(tramp) if fn is not None else C.cast(None, ALLREDUCE_FN)
    lib().pcrh_set_sharding(rank, world, cb, None)
    return cb
