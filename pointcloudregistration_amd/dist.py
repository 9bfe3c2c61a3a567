"""Source-sharded multi-GPU ICP loop.

One process per GPU.  Both clouds and both indices are replicated; the SAMPLES are split into
contiguous ranges (contiguous in the source index's cell order, so every rank's queries are spatially
coherent).  The only exchange per iteration is the sum of the 17 partials -- one all-reduce of 136
bytes (torch.distributed backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) --
after which every rank runs the same deterministic 3x3 solve, so (R, T) never needs a broadcast.
"""
import numpy as np
import torch


def shard_range(m, rank, world):
    """Contiguous split of [0, m) into `world` ranges whose sizes differ by at most one."""
    base, extra = divmod(m, world)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def all_reduce_sum(t):
    """Sum a tensor over all ranks in place (no-op when torch.distributed is not initialised)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def run_iterations(step, solve, R, T, iterations, all_reduce=all_reduce_sum, on_iteration=None):
    """The loop of CIterativeFit::__fitLoop (IterativeFit.cpp:119-129) for a fixed iteration count.

    step(R, T)      -> torch tensor of 17 float64 partial sums of THIS rank's samples (any device)
    solve(p, R, T)  -> (R', T', mse) from the GLOBAL partial sums (numpy 17-vector)
    Returns (R, T, list of (mse, n)).
    """
    R = np.array(R, np.float64).reshape(9).copy()
    T = np.array(T, np.float64).reshape(3).copy()
    hist = []
    for it in range(iterations):
        part = step(R, T)
        part = all_reduce(part)
        p = part.detach().cpu().numpy() if isinstance(part, torch.Tensor) else np.asarray(part)
        R, T, mse = solve(p, R, T)
        hist.append((mse, p[0]))
        if on_iteration is not None and on_iteration(it + 1, R, T, mse, p):
            break
    return R, T, hist
