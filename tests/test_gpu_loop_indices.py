"""The TIMED path is the CHECKED path: per-iteration correspondences out of the device-resident loop.

`bench.py` times `pcr_fit_enqueue_iteration`: the filtered step kernel pruned from the start by the temporal bound
(last iteration's K-th distance + how far a point can have moved), the in-kernel fold and the in-kernel solve.  That
path's indices used to be invisible (Missing 1 / Weak 3): only trajectories were compared, and one wrongly
pruned neighbour in a million moves T by ~1e-9 -- inside the tolerance.

Here every iteration of the loop also writes what the reference's loop hands from stage to stage
(IterativeFit.cpp:138-158; CorrespondenceEstimationClosestPoints.cpp:84-106): forward index, reverse index, keep flag
of EVERY sample (`pcr_fit_set_debug_outputs`).  They must equal, bit for bit, the STATELESS step
(`pcr_icp_step_d`, no temporal bound, host-side solve) at the same (R,T) -- the entry point that
tests/test_gpu_full_size.py checks sample by sample against the oracle at 1 M -- and the 17 sums and the next transform
must be bit-identical too.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _loop_vs_stateless(pcr, ctx, six, tix, R0, T0, k, iters, U, m_slots, expect_kernel, conv=None, batch_with=None,
                       oracle_check=None, sample_idx_t=None):
    dev = f"cuda:{ctx.device}"
    thr_f, thr_r = U * 36.0, U * 4.0
    fit = pcr.Fit(ctx, six, tix, k, thr_f, thr_r, conv=conv, unique_sq_dist=U, traj_cap=iters + 1,
                  sample_idx_t=sample_idx_t)
    fwd = torch.full((m_slots,), 0xDEAD, dtype=torch.int32, device=dev)
    rev = torch.full((m_slots,), 0xDEAD, dtype=torch.int32, device=dev)
    keep = torch.full((m_slots,), 7, dtype=torch.uint8, device=dev)
    s_fwd, s_rev = torch.empty_like(fwd), torch.empty_like(rev)
    s_keep = torch.empty_like(keep)
    part = torch.zeros(17, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()  # (torch filled these on ITS stream; the library's kernels run on the context's own)
    fit.set_debug_outputs(fwd, rev, keep)
    fit.reset(R0, T0)
    batch = pcr.FitBatch(ctx, [fit] + list(batch_with or [])) if batch_with is not None else None
    R, T = np.asarray(R0, np.float64).reshape(9).copy(), np.asarray(T0, np.float64).reshape(3).copy()
    checked = 0
    for it in range(1, iters + 1):
        if batch is not None:
            batch.iterate()
        else:
            fit.iterate()
        out = fit.read()   # synchronises
        if out["iterations"] < it:
            break          # the criteria fired at the previous iteration: this launch was a no-op
        assert pcr.last_step_kernel() == expect_kernel if batch is None else True
        lf, lr, lk = fwd.cpu().numpy().view(np.uint32), rev.cpu().numpy().view(np.uint32), keep.cpu().numpy()
        # the stateless step at the transform this iteration started from
        pcr.icp_step_d(ctx, six, tix, R, T, k, thr_f, thr_r, part, exact=False, out_fwd_t=s_fwd, out_rev_t=s_rev,
                       out_keep_t=s_keep, sample_idx_t=sample_idx_t)
        ctx.synchronize()
        sf, sr, sk = s_fwd.cpu().numpy().view(np.uint32), s_rev.cpu().numpy().view(np.uint32), s_keep.cpu().numpy()
        bad = np.flatnonzero((lf != sf) | (lr != sr) | (lk != sk))
        assert bad.size == 0, (f"iteration {it}: {bad.size} samples differ between the device-resident loop and the stateless "
                               f"step, first {bad[:5]}: fwd {lf[bad[:5]]} vs {sf[bad[:5]]}, rev {lr[bad[:5]]} vs {sr[bad[:5]]}")
        assert int(lk.sum()) == out["traj"][it - 1, 13] == part[0].item()
        if oracle_check is not None and it in oracle_check[1]:
            oracle_check[0](it, R, T, lf, lr, lk)
        # next transform: host solve of the stateless sums == the in-kernel solve of the loop's own sums, bit for bit
        pp, pq = pcr.icp_pivots(six, tix, R, T)
        R, T, _, _, mse = pcr.solve_rigid(part.cpu().numpy(), pp, pq, R, T)
        assert (out["traj"][it - 1, :9] == R).all() and (out["traj"][it - 1, 9:12] == T).all()
        assert out["traj"][it - 1, 12] == mse
        checked += 1
    fit.set_debug_outputs(None, None, None)
    if batch is not None:
        batch.close()
    fit.close()
    return checked


@pytest.mark.timeout(900)
def test_loop_indices_equal_the_stateless_step_at_1m_for_30_iterations(pcr, oracle):
    """C3 (the bench workload): 1 M x 1 M, K = 4, every source point a sample, ALL 30 iterations of the loop bench.py
    times (VERDICT r2 asked for 12; the late ones are where the temporal bound is tightest -- the images move by 2 % of a
    point spacing at iteration 30 -- so a bound that is too small would show there).  Iteration 3 (temporal bound active,
    far from converged) is additionally compared with the ORACLE sample by sample."""
    from pointcloudregistration_amd import synth
    N = 1_000_000
    src, tgt, Rm, t = synth.registration_pair(N, N, 20250001)
    ctx = pcr.Context(0)
    six = pcr.Index(ctx, pcr.Cloud(ctx, src), kind=pcr.INDEX_GRID)
    tix = pcr.Index(ctx, pcr.Cloud(ctx, tgt), kind=pcr.INDEX_GRID)
    U = tix.unique_sq_dist()
    R0, T0 = synth.coarse_initial_transform(Rm, t, N)
    tree = oracle.KdOld(tgt, 4)

    def vs_oracle(it, R, T, lf, lr, lk):
        want = oracle.icrp_correspondences(src, tgt, tree, R, T, 4, U * 36.0, U * 4.0, threads=16)
        # production mode reports a forward index only where the first gate can still pass, a reverse index only where
        # the reverse search ran; the oracle always reports both (exact mode is compared in test_gpu_full_size.py)
        found = lf != pcr.NO_INDEX
        assert (lf[found] == want["fwd"][found]).all()
        ran = lr != pcr.NO_INDEX
        assert (lr[ran] == want["rev"][ran]).all()
        assert (lk == want["keep"]).all() and ran.sum() >= lk.sum() > 100000

    n = _loop_vs_stateless(pcr, ctx, six, tix, R0, T0, 4, 30, U, N, "icp_step_fast_kernel<4>",
                           oracle_check=(vs_oracle, {3}))
    assert n == 30
    ctx.close()


def test_loop_indices_k1_and_sample_list(pcr):
    """K = 1 (RegCfg_4_1's NN), a strided sample list (slot = position in the list), 300 k points."""
    from pointcloudregistration_amd import synth
    N = 300_000
    src, tgt, Rm, t = synth.registration_pair(N, N, 4242)
    ctx = pcr.Context(0)
    six, tix = pcr.Index(ctx, pcr.Cloud(ctx, src)), pcr.Index(ctx, pcr.Cloud(ctx, tgt))
    U = tix.unique_sq_dist()
    R0, T0 = synth.coarse_initial_transform(Rm, t, N)
    samples = torch.arange(5, N, 3, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    assert _loop_vs_stateless(pcr, ctx, six, tix, R0, T0, 1, 8, U, samples.numel(), "icp_step_fast_kernel<1>",
                              sample_idx_t=samples) == 8
    assert _loop_vs_stateless(pcr, ctx, six, tix, R0, T0, 1, 4, U, N, "icp_step_fast_kernel<1>") == 4
    ctx.close()


def test_loop_indices_on_a_partial_overlap_pair(pcr):
    """65 k-point partial scans: a tenth of the lanes leave the filtered walk for the fp64 shells, seeded with the filtered
    walk's exact list (fallback_forward) -- the route the full-overlap pair hardly exercises.  RegCfg criteria on: the
    loop stops by itself, and every iteration it ran is compared."""
    from pointcloudregistration_amd import synth
    scans, _ = synth.scan_sequence(2, 65000, 99, step_deg=1.0)
    src, tgt = scans[1], scans[0]
    ctx = pcr.Context(0)
    six, tix = pcr.Index(ctx, pcr.Cloud(ctx, src)), pcr.Index(ctx, pcr.Cloud(ctx, tgt))
    U = tix.unique_sq_dist()
    n = _loop_vs_stateless(pcr, ctx, six, tix, np.eye(3).reshape(9), np.zeros(3), 4, 30, U, len(src),
                           "icp_step_fast_kernel<4>", conv=pcr.ConvergenceParams.regcfg(30))
    assert n >= 5
    ctx.close()


def test_loop_indices_through_the_one_launch_batch(pcr):
    """pcr_fit_batch (C4): the observed fit advances inside icp_step_batch_kernel together with two other fits."""
    from pointcloudregistration_amd import synth
    scans, _ = synth.scan_sequence(4, 50000, 31, step_deg=1.0)
    ctx = pcr.Context(0)
    idx = [pcr.Index(ctx, pcr.Cloud(ctx, s)) for s in scans]
    U = [i.unique_sq_dist() for i in idx]
    others = []
    for a in (2, 3):
        f = pcr.Fit(ctx, idx[a], idx[a - 1], 4, U[a - 1] * 36.0, U[a - 1] * 4.0, conv=pcr.ConvergenceParams.regcfg(30),
                    unique_sq_dist=U[a - 1])
        f.reset(np.eye(3).reshape(9), np.zeros(3))
        others.append(f)
    n = _loop_vs_stateless(pcr, ctx, idx[1], idx[0], np.eye(3).reshape(9), np.zeros(3), 4, 30, U[0], len(scans[1]),
                           None, conv=pcr.ConvergenceParams.regcfg(30), batch_with=others)
    assert n >= 5
    for f in others:
        f.close()
    ctx.close()
