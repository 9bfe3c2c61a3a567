import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def case_inputs(name, g):
    """(src, tgt) of a golden case: stored for the scan patches, regenerated from the seed otherwise."""
    from pointcloudregistration_amd import synth
    if "src" in g:
        return g["src"], g["tgt"]
    seed, n = int(g["seed"]), int(g["n"])
    if name.startswith("cube4096"):
        tgt = synth.cube_surface(n, seed)
        s = synth.cube_surface(n, seed + 1)
        rng = np.random.default_rng(seed + 2)
        s += rng.standard_normal(s.shape) * 0.2 * np.sqrt(6.0 / n)
        Rm, t = synth.euler_zyx(-3.0, 5.0, 10.0), np.array([0.02, -0.01, 0.03])
        return np.ascontiguousarray((s - t) @ Rm), tgt
    if name.startswith("spheres40k"):
        src, tgt, _, _ = synth.registration_pair(n, n, seed)
        return src, tgt
    raise KeyError(name)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def pcr():
    """The product binding.  GPU tests only: fails loudly if libpcr.so or the device is missing."""
    from pointcloudregistration_amd import capi
    capi.lib()
    return capi


@pytest.fixture(scope="session")
def ctx(pcr):
    c = pcr.Context(0)
    yield c
    c.close()
