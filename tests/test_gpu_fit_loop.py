"""GPU parity of the device-resident fit loop (pcr_fit_*): solve, composition and convergence criteria run on the
device, nothing leaves it between iterations.  Checked against the golden trajectories (the oracle's restatement of the
reference's tree and loop, tests/golden/make_golden.py) and, bit for bit, against the host path (pcr_icp_step_d + pcr_solve_rigid), which shares its source."""
import numpy as np
import pytest

from conftest import case_inputs, golden

pytestmark = pytest.mark.gpu


def _setup(pcr, ctx, name):
    g = golden(name + ".npz")
    src, tgt = case_inputs(name, g)
    six, tix = pcr.Index(ctx, pcr.Cloud(ctx, src)), pcr.Index(ctx, pcr.Cloud(ctx, tgt))
    return g, src, tgt, six, tix


@pytest.mark.parametrize("name", ["patch_k1", "patch_k4", "cube4096_k1", "cube4096_k4", "spheres40k_k4"])
def test_device_loop_matches_golden(pcr, ctx, name):
    g, src, tgt, six, tix = _setup(pcr, ctx, name)
    k, U = int(g["k"]), float(g["uniq_tgt"])
    fit = pcr.Fit(ctx, six, tix, k, U * g["sdft"], U * g["ndft"], conv=pcr.ConvergenceParams.regcfg(30),
                  unique_sq_dist=U, traj_cap=64)
    out = fit.run(g["R0"], g["T0"], 40)  # more steps enqueued than needed: the surplus must be no-ops
    assert out["iterations"] == int(g["iterations"])
    assert out["state"] == int(g["state"])
    scale = max(1.0, np.abs(tgt).max())
    np.testing.assert_allclose(out["R"], g["final_R"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(out["T"], g["final_T"], rtol=0, atol=1e-9 * scale)
    n = len(g["traj"])
    assert len(out["traj"]) == n
    np.testing.assert_allclose(out["traj"][:, :12], g["traj"][:, :12], rtol=0, atol=1e-9 * scale)
    np.testing.assert_allclose(out["traj"][:, 12], g["traj"][:, 12], rtol=1e-9)
    assert (out["traj"][:, 13] == g["traj"][:, 13]).all()
    # a second run on the same object starts from scratch (criteria state reset) and repeats itself bit for bit
    again = fit.run(g["R0"], g["T0"], 40)
    assert again["iterations"] == out["iterations"] and (again["R"] == out["R"]).all() and (again["T"] == out["T"]).all()


def test_device_loop_is_bit_identical_to_host_loop(pcr, ctx):
    """Same kernels, same reduction order, same solve source: the two loops must agree exactly."""
    import torch
    g, src, tgt, six, tix = _setup(pcr, ctx, "cube4096_k4")
    U = float(g["uniq_tgt"])
    part = torch.zeros(17, dtype=torch.float64, device="cuda")
    R, T = g["R0"].copy(), g["T0"].copy()
    host = []
    for _ in range(6):
        pcr.icp_step_d(ctx, six, tix, R, T, 4, U * 36, U * 4, part)
        ctx.synchronize()
        p = part.cpu().numpy()
        pp, pq = pcr.icp_pivots(six, tix, R, T)
        R, T, _, _, mse = pcr.solve_rigid(p, pp, pq, R, T)
        host.append(np.concatenate([R, T, [mse, p[0]]]))
    fit = pcr.Fit(ctx, six, tix, 4, U * 36, U * 4, conv=None, traj_cap=16)
    out = fit.run(g["R0"], g["T0"], 6)
    assert out["iterations"] == 6 and out["state"] == 0
    assert (out["traj"] == np.array(host)).all()
    assert (out["R"] == R).all() and (out["T"] == T).all()


def test_device_loop_stops_without_correspondences(pcr, ctx):
    g, src, tgt, six, tix = _setup(pcr, ctx, "patch_k1")
    U = float(g["uniq_tgt"])
    fit = pcr.Fit(ctx, six, tix, 1, U * 36, U * 4, conv=pcr.ConvergenceParams.regcfg(30), unique_sq_dist=U)
    out = fit.run(np.eye(3).reshape(9), np.zeros(3), 5)  # the patches are 0.03 apart: nothing passes the gates
    assert out["iterations"] == 1 and out["state"] == 5 and out["msg"] == "Criteria_NoCorrespondeces"
    assert (out["R"] == np.eye(3).reshape(9)).all() and (out["T"] == 0).all()


def test_device_loop_with_external_reduction(pcr, ctx):
    """The multi-GPU hook: the caller reduces the 17 partials in place between step and solve.  Two half-range
    fits whose partials are summed by hand must reproduce the full-range fit (pair counts exactly)."""
    g, src, tgt, six, tix = _setup(pcr, ctx, "cube4096_k4")
    U = float(g["uniq_tgt"])
    n = len(src)
    whole = pcr.Fit(ctx, six, tix, 4, U * 36, U * 4, traj_cap=8).run(g["R0"], g["T0"], 3)
    a = pcr.Fit(ctx, six, tix, 4, U * 36, U * 4, sample_begin=0, m=n // 2, traj_cap=8)
    b = pcr.Fit(ctx, six, tix, 4, U * 36, U * 4, sample_begin=n // 2, m=n - n // 2, traj_cap=8)
    a.reset(g["R0"], g["T0"])
    b.reset(g["R0"], g["T0"])
    pa, pb = a.partials_tensor(), b.partials_tensor()
    import torch
    for _ in range(3):
        a.step()               # (an index holds no per-iteration state: the two fits need no serialising)
        b.step()
        ctx.synchronize()      # the fixture's context has its own stream; torch works on another one
        tot = pa + pb          # what an all-reduce leaves on every rank
        pa.copy_(tot)
        pb.copy_(tot)
        torch.cuda.synchronize()
        a.solve()
        b.solve()
    ra, rb = a.read(), b.read()
    assert (ra["R"] == rb["R"]).all() and (ra["T"] == rb["T"]).all()
    assert (ra["traj"][:, 13] == whole["traj"][:, 13]).all()
    np.testing.assert_allclose(ra["R"], whole["R"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(ra["T"], whole["T"], rtol=0, atol=1e-12)


def test_rccl_allreduce_entry_point_single_rank(pcr, ctx):
    """pcr_comm_* / pcr_allreduce_partials with a one-rank communicator (the box has one GPU): the in-place
    all-reduce between step and solve must leave the fit bit-identical to the loop without it.  N > 1 runs the
    same call with more ranks (bench.py uses torch.distributed's RCCL communicator for that)."""
    g, src, tgt, six, tix = _setup(pcr, ctx, "cube4096_k4")
    U = float(g["uniq_tgt"])
    comm = pcr.Comm(ctx, 0, 1, pcr.Comm.unique_id())
    fit = pcr.Fit(ctx, six, tix, 4, U * 36, U * 4, traj_cap=8)
    plain = fit.run(g["R0"], g["T0"], 4)
    ptr = pcr.lib().pcr_fit_partials_d(fit.h)
    out = fit.run(g["R0"], g["T0"], 4, all_reduce=lambda _t: comm.allreduce_partials(ptr))
    assert (out["traj"] == plain["traj"]).all() and (out["R"] == plain["R"]).all()
    comm.close()


def test_concurrent_fits_on_two_contexts_share_one_pair_of_indices(pcr, ctx):
    """Indices are read-only: two contexts (two streams) fit on the SAME source and target index at the same time, from
    different initial transforms, and each must reproduce what it gives when it runs alone."""
    from pointcloudregistration_amd import synth
    n = 120000
    src, tgt, Rm, t = synth.registration_pair(n, n, 77)
    six, tix = pcr.Index(ctx, pcr.Cloud(ctx, src), kind=pcr.INDEX_GRID), pcr.Index(ctx, pcr.Cloud(ctx, tgt), kind=pcr.INDEX_GRID)
    U = tix.unique_sq_dist()
    other = pcr.Context(ctx.device)
    starts = [synth.coarse_initial_transform(Rm, t, n), synth.coarse_initial_transform(Rm, t, n, rot_spacings=0.7, trans_spacings=1.6)]
    conv = pcr.ConvergenceParams.regcfg(12)
    alone = [pcr.Fit(c, six, tix, 4, U * 36, U * 4, conv=conv, unique_sq_dist=U, traj_cap=16).run(R, T, 12)
             for c, (R, T) in zip((ctx, other), starts)]
    fits = [pcr.Fit(c, six, tix, 4, U * 36, U * 4, conv=conv, unique_sq_dist=U, traj_cap=16) for c in (ctx, other)]
    for f, (R, T) in zip(fits, starts):
        f.reset(R, T)
    for _ in range(12):        # interleaved enqueues on two streams: the GPU overlaps them as it likes
        for f in fits:
            f.iterate()
    both = [f.read() for f in fits]
    for a, b in zip(alone, both):
        assert a["iterations"] == b["iterations"] and a["state"] == b["state"]
        assert (a["traj"] == b["traj"]).all() and (a["R"] == b["R"]).all() and (a["T"] == b["T"]).all()
    assert alone[0]["pairs"] > 10000
    for f in fits:
        f.close()
    other.close()


def test_graph_replay_of_many_iterations_equals_plain_launches(pcr, ctx):
    """pcr_fit_enqueue_iterations: N iterations captured once as a HIP graph and replayed -- the whole trajectory must
    equal N single launches bit for bit, on the first (capturing) call, on a replay, with a different count (recapture),
    with criteria on (surplus iterations are no-ops), and with kernel timing on (falls back to plain launches)."""
    g = golden("spheres40k_k4.npz")
    src, tgt = case_inputs("spheres40k_k4", g)
    six, tix = pcr.Index(ctx, pcr.Cloud(ctx, src), kind=pcr.INDEX_GRID), pcr.Index(ctx, pcr.Cloud(ctx, tgt), kind=pcr.INDEX_GRID)
    U = tix.unique_sq_dist()
    for conv in (None, pcr.ConvergenceParams.regcfg(30)):
        fit = pcr.Fit(ctx, six, tix, 4, U * 36, U * 4, conv=conv, unique_sq_dist=U, traj_cap=16)
        want = fit.run(g["R0"], g["T0"], 12)
        for count, timing in ((12, False), (12, False), (7, False), (12, True)):
            fit.reset(g["R0"], g["T0"])
            ctx.set_profiling(timing)
            fit.iterate_many(count)
            if count < 12:
                fit.iterate_many(12 - count)
            ctx.set_profiling(False)
            got = fit.read()
            assert got["iterations"] == want["iterations"] and got["state"] == want["state"]
            assert (got["traj"] == want["traj"]).all() and (got["R"] == want["R"]).all() and (got["T"] == want["T"]).all()
        fit.close()
