"""GPU parity: batched exact K-NN through the C ABI vs the oracle and the golden vectors."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["grid", "brute"])
def test_patch_knn_golden(pcr, ctx, kind):
    """Known answers produced by the reference's own KD-tree (tests/golden/patch_knn.npz)."""
    g = golden("patch_knn.npz")
    tgt = golden("patch_k1.npz")["tgt"]
    cloud = pcr.Cloud(ctx, tgt)
    ix = pcr.Index(ctx, cloud, kind=pcr.INDEX_GRID if kind == "grid" else pcr.INDEX_BRUTE)
    for k in (1, 2, 4, 8, 25, 32):
        got = ix.knn(g["q"], k)
        assert (got == g[f"idx_k{k}"]).all(), f"k={k}: {(got != g[f'idx_k{k}']).sum()} mismatches"
        desc = ix.knn(g["q"], k, order=pcr.ORDER_DESCENDING)
        assert (desc == g[f"idx_k{k}"][:, ::-1]).all()


@pytest.mark.parametrize("n,nq,k", [(5000, 3000, 1), (5000, 3000, 4), (200000, 50000, 4), (200000, 20000, 25)])
def test_random_volume_vs_oracle(pcr, ctx, oracle, n, nq, k):
    rng = np.random.default_rng(n + k)
    X = rng.random((n, 3)) * [1.0, 2.0, 0.5] + [10.0, -3.0, 0.0]
    Q = rng.random((nq, 3)) * [1.4, 2.4, 0.9] + [9.8, -3.2, -0.2]  # some queries outside the bounds
    want, wd = oracle.KdOld(X, k).query(Q, want_d2=True, threads=8)
    ix = pcr.Index(ctx, pcr.Cloud(ctx, X), kind=pcr.INDEX_GRID)
    got, gd = ix.knn(Q, k, want_d2=True)
    assert (got == want).all()
    assert (gd == wd).all()  # squared distances bit-identical (same operation order, no FMA)
