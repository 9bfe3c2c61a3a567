"""The entry point that allocates nothing (pcr_icp_step_noalloc_d, include/pcr.h) against the entry point it is a
variant of (pcr_icp_step_d), through the C ABI: bit-identical partial sums and per-sample outputs on two calls in
a row, the second with a smaller input -- the fixed scratch keeps the first call's contents, and nothing of them may
leak into the second.  Collected last (file name), so a failure here hides no other test."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(pcr, ctx, six, tix, R, T, k, thr_f, thr_r, begin, m, n, exact):
    """(partials, fwd, rev, keep) of the plain and of the no-allocation entry point for one sample range."""
    dev = f"cuda:{ctx.device}"
    out = []
    for noalloc in (False, True):
        part = torch.full((17,), float("nan"), dtype=torch.float64, device=dev)
        fwd = torch.full((n,), 0x7fffffff, dtype=torch.int32, device=dev)
        rev = torch.full((n,), 0x7fffffff, dtype=torch.int32, device=dev)
        keep = torch.full((n,), 7, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        pcr.icp_step_d(ctx, six, tix, R, T, k, thr_f, thr_r, part, sample_begin=begin, m=m, exact=exact,
                       out_fwd_t=fwd, out_rev_t=rev, out_keep_t=keep, noalloc=noalloc)
        ctx.synchronize()
        out.append((part.cpu().numpy(), fwd.cpu().numpy(), rev.cpu().numpy(), keep.cpu().numpy()))
    return out


def test_noalloc_step_equals_step_d_twice_second_call_smaller(pcr, ctx):
    from pointcloudregistration_amd import synth
    R0, T0 = None, None
    sizes = [(60000, 60000, 4), (9000, 60000, 1)]  # (source points, target points, K): the second call is smaller
    for call, (ns, nt, k) in enumerate(sizes):
        src, tgt, Rm, t = synth.registration_pair(ns, nt, 71 + call)
        six = pcr.Index(ctx, pcr.Cloud(ctx, src), kind=pcr.INDEX_GRID)
        tix = pcr.Index(ctx, pcr.Cloud(ctx, tgt), kind=pcr.INDEX_GRID)
        U = tix.unique_sq_dist()
        R0, T0 = synth.coarse_initial_transform(Rm, t, ns)
        # the whole cloud, then a sub-range of the processing order (what a rank of a sharded fit runs)
        for begin, m in ((0, ns), (ns // 3, ns // 2)):
            for exact in (True, False):
                plain, noalloc = _both(pcr, ctx, six, tix, R0, T0, k, U * 36.0, U * 4.0, begin, m, ns, exact)
                assert plain[0][0] > 100, "the case must keep pairs to mean anything"
                for a, b, what in zip(plain, noalloc, ("partials", "fwd", "rev", "keep")):
                    assert (a == b).all(), (call, begin, m, exact, what)
                if m < ns:  # range mode writes only the entries of its own points (indexed by original index)
                    untouched = plain[3] == 7
                    assert untouched.sum() == ns - m
        six.close()
        tix.close()


def test_noalloc_step_with_a_sample_list(pcr, ctx):
    from pointcloudregistration_amd import synth
    src, tgt, Rm, t = synth.registration_pair(30000, 30000, 5)
    six = pcr.Index(ctx, pcr.Cloud(ctx, src), kind=pcr.INDEX_GRID)
    tix = pcr.Index(ctx, pcr.Cloud(ctx, tgt), kind=pcr.INDEX_GRID)
    U = tix.unique_sq_dist()
    R, T = synth.coarse_initial_transform(Rm, t, 30000)
    dev = f"cuda:{ctx.device}"
    smp = torch.from_numpy(np.random.default_rng(2).choice(30000, 5000, replace=False).astype(np.int32)).to(dev)
    res = []
    for noalloc in (False, True):
        part = torch.zeros(17, dtype=torch.float64, device=dev)
        fwd = torch.zeros(5000, dtype=torch.int32, device=dev)
        pcr.icp_step_d(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, part, sample_idx_t=smp, exact=True, out_fwd_t=fwd,
                       noalloc=noalloc)
        ctx.synchronize()
        res.append((part.cpu().numpy(), fwd.cpu().numpy()))
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all()
    assert res[0][0][0] > 100


def test_noalloc_step_rejects_sizes_beyond_its_scratch(pcr, ctx):
    n = pcr.NOALLOC_MAX_SAMPLES + 4096
    rng = np.random.default_rng(9)
    pts = rng.random((n, 3))
    six = pcr.Index(ctx, pcr.Cloud(ctx, pts), kind=pcr.INDEX_GRID)
    part = torch.zeros(17, dtype=torch.float64, device=f"cuda:{ctx.device}")
    R, T = np.eye(3).reshape(9), np.zeros(3)
    with pytest.raises(pcr.PcrError) as e:
        pcr.icp_step_d(ctx, six, six, R, T, 1, 1e-4, 1e-4, part, m=n, noalloc=True)
    assert e.value.status == -1 and "PCR_NOALLOC_MAX_SAMPLES" in str(e.value)
    # the largest size it accepts runs (a range of the same cloud), and agrees with the plain entry point
    got = []
    for noalloc in (False, True):
        pcr.icp_step_d(ctx, six, six, R, T, 1, 1e-4, 1e-4, part, sample_begin=100, m=pcr.NOALLOC_MAX_SAMPLES,
                       noalloc=noalloc)
        ctx.synchronize()
        got.append(part.cpu().numpy().copy())
    assert (got[0] == got[1]).all() and got[0][0] == pcr.NOALLOC_MAX_SAMPLES  # every point is its own nearest neighbour
    six.close()


def test_noalloc_step_runs_under_stream_capture_and_clears_stale_tickets(pcr):
    """The proof that the call allocates nothing: it is captured into a HIP graph on the context's stream, in GLOBAL
    capture mode, where hipMalloc/hipFree (from any thread) are errors that invalidate the capture -- a capture that
    ends well and a replay that reproduces the plain entry point's results bit for bit is that proof.  Before it, the
    fixed scratch's tickets are deliberately left NON-ZERO (what a launch that died mid-fold would leave): the entry
    point clears them itself, every call, so later calls are not poisoned for the life of the process."""
    from pointcloudregistration_amd import synth
    stream = torch.cuda.Stream()
    ctx = pcr.Context(0, stream=stream)
    n = 50000
    src, tgt, Rm, t = synth.registration_pair(n, n, 1234)
    six = pcr.Index(ctx, pcr.Cloud(ctx, src), kind=pcr.INDEX_GRID)
    tix = pcr.Index(ctx, pcr.Cloud(ctx, tgt), kind=pcr.INDEX_GRID)
    U = tix.unique_sq_dist()
    R, T = synth.coarse_initial_transform(Rm, t, n)
    dev = f"cuda:{ctx.device}"

    def buffers():
        return (torch.full((17,), float("nan"), dtype=torch.float64, device=dev),
                torch.full((n,), 0x7fffffff, dtype=torch.int32, device=dev),
                torch.full((n,), 0x7fffffff, dtype=torch.int32, device=dev),
                torch.full((n,), 7, dtype=torch.uint8, device=dev))
    want = buffers()
    got = buffers()
    torch.cuda.synchronize()
    pcr.icp_step_d(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, want[0], exact=True, out_fwd_t=want[1], out_rev_t=want[2],
                   out_keep_t=want[3])
    ctx.synchronize()
    # poison the tickets of the fixed scratch, as an aborted launch would; a plain (uncaptured) call first
    assert pcr.lib().pcr_noalloc_debug_poison_tickets(ctx.h) == 0
    pcr.icp_step_d(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, got[0], exact=True, out_fwd_t=got[1], out_rev_t=got[2],
                   out_keep_t=got[3], noalloc=True)
    ctx.synchronize()
    for a, b, what in zip(want, got, ("partials", "fwd", "rev", "keep")):
        assert torch.equal(a, b), ("uncaptured call after poisoned tickets", what)
    got = buffers()
    torch.cuda.synchronize()
    assert pcr.lib().pcr_noalloc_debug_poison_tickets(ctx.h) == 0
    ctx.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream, capture_error_mode="global"):
        pcr.icp_step_d(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, got[0], exact=True, out_fwd_t=got[1], out_rev_t=got[2],
                       out_keep_t=got[3], noalloc=True)
    torch.cuda.synchronize()
    assert torch.isnan(got[0]).all() and (got[3] == 7).all()   # captured, not run
    for _ in range(2):                                          # replayed twice: the second run starts from the first's tickets
        got[0].fill_(float("nan"))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert want[0][0].item() > 1000
        for a, b, what in zip(reversed(want), reversed(got), ("keep", "rev", "fwd", "partials")):
            assert torch.equal(a, b), ("graph replay", what)
    del graph
    six.close()
    tix.close()
    ctx.close()
