#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the build container only).

What these vectors are
  * The outputs of the ORACLE (oracle/liboracle.so: the CPU restatement of the reference's path) on fixed inputs,
    frozen so that neither the restatement nor the HIP path can drift unnoticed.  They are NOT outputs of the
    reference: its sources cannot be built in this image (oracle/Makefile), and it holds no expected outputs of its
    own -- "parity unpinned", DESIGN.md section 1.  (Round 1 generated the same files while also
    cross-checking every index against a build of Common/KNNSearch.cpp that needed a stand-in header; that build is
    gone, the values are unchanged -- `git diff` after a regeneration shows nothing.)
  * Rc/Tc of the first iteration are cross-checked against numpy.linalg.svd (the reference's Eigen
    JacobiSVD is absent; the polar factor is algorithm-independent).
Inputs: the two scan patches the reference ships (MyLog/2015.1.17/cur_test_data/{src,tgt}.ply,
coordinates only, re-encoded as fp64) and seeded synthetic clouds from
pointcloudregistration_amd/synth.py (regenerated from the seed at test time, not stored).

usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from pointcloudregistration_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/PointCloudRegistration"


def read_ply_xyz(path):
    """ASCII PLY: scan for `element vertex N` and `end_header`, take the first three columns
    (the convention of PLYLoad.cpp:76-111)."""
    with open(path, "r", errors="replace") as f:
        n = None
        for line in f:
            t = line.split()
            if len(t) >= 3 and t[0] == "element" and t[1] == "vertex":
                n = int(t[2])
            if t and t[0] == "end_header":
                break
        rows = []
        for _ in range(n):
            rows.append([float(v) for v in f.readline().split()[:3]])
    return np.asarray(rows, np.float64)


def sqd(a, b):
    d = a - b
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]


def ref_iteration(src, tgt, ref_tgt_tree, R, T, k, thrF, thrR, sample_idx=None):
    """One ICRP correspondence pass driven query by query from Python (a second route through the restated tree,
    independent of the oracle's own loop in icrp.cpp)."""
    smp = np.arange(len(src), dtype=np.uint32) if sample_idx is None else sample_idx
    tsrc = O.transform(src, R, T)  # PointCloud.cpp:368-370 arithmetic (compiled, no FMA)
    p = tsrc[smp]
    fwd = ref_tgt_tree.query(p)[:, k - 1]  # NeighborSet.back()
    q = tgt[fwd]
    rev = O.KdOld(tsrc, 1).query(q)[:, 0]
    pp = tsrc[rev]
    keep = ((sqd(p, q) < thrF) & (sqd(p, pp) < thrR)).astype(np.uint8)
    return fwd.astype(np.uint32), rev.astype(np.uint32), keep


def partial_sums(p, q, keep, pivot_p, pivot_q):
    """The 17 sums of include/pcr.h in extended precision (np.longdouble) -- a reference VALUE for the
    tolerance test, independent of any summation order."""
    m = keep.astype(bool)
    a = (p[m] - pivot_p).astype(np.longdouble)
    b = (q[m] - pivot_q).astype(np.longdouble)
    out = np.zeros(17, np.longdouble)
    out[0] = m.sum()
    out[1:4] = a.sum(0)
    out[4:7] = b.sum(0)
    out[7:16] = (a[:, :, None] * b[:, None, :]).sum(0).reshape(9)
    out[16] = sqd(p[m], q[m]).astype(np.longdouble).sum()
    return out.astype(np.float64)


def tie_gap(tgt, queries, k):
    """min over queries of d2_(K+1) - d2_K: > 0 means bit-exact index comparison is meaningful."""
    if len(tgt) <= k:
        return np.inf
    _, d2 = O.knn_brute(tgt, queries, k + 1)
    return float((d2[:, k] - d2[:, k - 1]).min())


def make_case(name, src, tgt, k, n_index_iters, R0=None, T0=None, store_inputs=False, extra=None):
    conv = O.ConvParams.regcfg(30)
    U = O.uniq_sq_dist(tgt)
    U_src = O.uniq_sq_dist(src)
    sdft, ndft = 36.0, 4.0  # RegCfg.txt:77-78
    fit = O.fit(src, tgt, k=k, sdft=sdft, ndft=ndft, conv=conv, R0=R0, T0=T0)
    n_it = fit["iterations"]
    traj = fit["traj"]
    R = np.array(R0 if R0 is not None else synth.colmajor(np.eye(3)))
    T = np.array(T0 if T0 is not None else np.zeros(3))
    ref_tree = O.KdOld(tgt, k)
    orc_tree = O.KdOld(tgt, k)
    it_R, it_T, it_fwd, it_rev, it_keep, it_part, it_gap = [], [], [], [], [], [], []
    lo_s, hi_s = src.min(0), src.max(0)
    lo_t, hi_t = tgt.min(0), tgt.max(0)
    c_src, pivot_q = 0.5 * (lo_s + hi_s), 0.5 * (lo_t + hi_t)
    for it in range(min(n_index_iters, n_it)):
        fwd, rev, keep = ref_iteration(src, tgt, ref_tree, R, T, k, U * sdft, U * ndft)
        oc = O.icrp_correspondences(src, tgt, orc_tree, R, T, k, U * sdft, U * ndft)
        assert (oc["fwd"] == fwd).all() and (oc["rev"] == rev).all() and (oc["keep"] == keep).all(), \
            f"{name}: oracle loop disagrees with the query-by-query pass at iteration {it}"
        tsrc = O.transform(src, R, T)
        Rm = synth.rowcol(R)
        pivot_p = Rm @ c_src + T
        it_R.append(R.copy())
        it_T.append(T.copy())
        it_fwd.append(fwd)
        it_rev.append(rev)
        it_keep.append(keep)
        it_part.append(partial_sums(tsrc, tgt[fwd], keep, pivot_p, pivot_q))
        it_gap.append(min(tie_gap(tgt, tsrc, k), tie_gap(tsrc, tgt[fwd], 1)))
        if it == 0:  # cross-check the first increment against numpy's SVD
            m = keep.astype(bool)
            a, b = tsrc[m], tgt[fwd][m]
            ca, cb = a.mean(0), b.mean(0)
            H = (a - ca).T @ (b - cb)
            Uu, _, Vt = np.linalg.svd(H)
            V = Vt.T
            if np.linalg.det(Uu) * np.linalg.det(V) < 0:
                V[:, 2] *= -1
            Rc_np = V @ Uu.T
            _, _, Rc_o, Tc_o = O.svd2(a, b, R, T)
            assert np.abs(synth.rowcol(Rc_o) - Rc_np).max() < 1e-12, f"{name}: oracle SVD2 vs numpy"
            assert np.abs(Tc_o - (cb - Rc_np @ ca)).max() < 1e-12
        R, T = traj[it, :9].copy(), traj[it, 9:12].copy()
    data = dict(k=k, sdft=sdft, ndft=ndft, uniq_tgt=U, uniq_src=U_src, iterations=n_it, state=fit["state"],
                traj=traj, final_R=fit["R"], final_T=fit["T"], it_R=np.array(it_R), it_T=np.array(it_T),
                it_fwd=np.array(it_fwd), it_rev=np.array(it_rev), it_keep=np.array(it_keep),
                it_partials=np.array(it_part), it_tie_gap=np.array(it_gap),
                R0=np.array(R0 if R0 is not None else synth.colmajor(np.eye(3))),
                T0=np.array(T0 if T0 is not None else np.zeros(3)))
    if store_inputs:
        data.update(src=src, tgt=tgt)
    if extra:
        data.update(extra)
    path = os.path.join(OUT, f"{name}_k{k}.npz")
    np.savez_compressed(path, **data)
    print(f"{name} k={k}: {n_it} iterations, state {O.Convergence.STATES[fit['state']]}, "
          f"kept {int(it_keep[0].sum())}/{len(src)} at it 1, min tie gap {min(it_gap):.3e} -> {os.path.basename(path)}"
          f" ({os.path.getsize(path) / 1024:.0f} KiB)")
    return data


def main():
    O.build()

    # ---- the two scan patches the reference ships --------------------------------------
    psrc = read_ply_xyz(os.path.join(REF, "MyLog/2015.1.17/cur_test_data/src.ply"))
    ptgt = read_ply_xyz(os.path.join(REF, "MyLog/2015.1.17/cur_test_data/tgt.ply"))
    assert psrc.shape == (1353, 3) and ptgt.shape == (1090, 3)
    # The raw patches are 0.03 apart (>> sqrt(36 U) = 0.0035): with the identity as the initial transform no
    # pair survives and the reference divides by zero.  A coarse initial translation (difference of the
    # centroids, rounded to 1e-4) stands in for the coarse registration that precedes fineFit.
    T0 = np.array([-0.0183, -0.0009, -0.0309])
    for k in (1, 4):
        d = make_case("patch", psrc, ptgt, k, n_index_iters=30, T0=T0, store_inputs=True)
    # K-NN answers on the patch target
    rng = np.random.default_rng(7)
    q = ptgt[rng.integers(0, len(ptgt), 400)] + rng.standard_normal((400, 3)) * 2e-3
    q = np.concatenate([q, (ptgt.min(0) - 0.05)[None], (ptgt.max(0) + 0.07)[None], ptgt[:16]])
    knn = {f"idx_k{k}": O.KdOld(ptgt, k).query(q) for k in (1, 2, 4, 8, 25, 32)}
    np.savez_compressed(os.path.join(OUT, "patch_knn.npz"), q=q, **knn)
    # CRejectionCentroidDist on the first-iteration pairs (oracle restatement).  MSD is in the cloud's units: 8.0 (default) keeps all 25; 1e-5 cuts lists.
    m = d["it_keep"][0].astype(bool)
    pp, pq = O.transform(psrc, d["it_R"][0], d["it_T"][0])[m], ptgt[d["it_fwd"][0]][m]
    rej = {}
    for tag, msd in (("default", 8.0), ("tight", 1.0e-5)):
        r = O.reject_centroid_dist(psrc, ptgt, pp, pq, msn=25, msd=msd, cdf=1.0)
        rej.update({f"{tag}_d": r["d"], f"{tag}_keep": r["keep"], f"{tag}_mean": r["mean"], f"{tag}_sigma": r["sigma"],
                    f"{tag}_msd": msd})
    np.savez_compressed(os.path.join(OUT, "patch_reject.npz"), pair_p=pp, pair_q=pq, **rej)

    # ---- seeded synthetic clouds (inputs are regenerated from the seed by the tests) ----
    n, seed = 4096, 20250101
    tgt = synth.cube_surface(n, seed)
    s = synth.cube_surface(n, seed + 1)
    rng = np.random.default_rng(seed + 2)
    s += rng.standard_normal(s.shape) * 0.2 * np.sqrt(6.0 / n)
    Rm, t = synth.euler_zyx(-3.0, 5.0, 10.0), np.array([0.02, -0.01, 0.03])
    src = np.ascontiguousarray((s - t) @ Rm)
    for k in (1, 4):
        make_case("cube4096", src, tgt, k, n_index_iters=30, extra=dict(seed=seed, n=n))

    src, tgt, Rm, t = synth.registration_pair(40000, 40000, 20250001)
    for k in (1, 4):
        make_case("spheres40k", src, tgt, k, n_index_iters=3, extra=dict(seed=20250001, n=40000, R_true=Rm, t_true=t))


if __name__ == "__main__":
    main()
