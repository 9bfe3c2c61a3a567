"""CPU tests of the boundary: libpcr.so loads without a GPU, exports every symbol include/pcr.h declares,
and its host-only entry points behave (no device compute is attempted here)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pcr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pcr):
    import ctypes
    L = ctypes.CDLL(pcr.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/pcr.h but not exported by libpcr.so"
    # and the Python binding covers the same set
    assert set(pcr.SYMBOLS) == set(names)


def test_no_cpu_fallback_without_device(pcr):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pcr.PcrError) as e:
        pcr.Context(0)
    assert e.value.status == -6  # PCR_E_NO_DEVICE: the product path fails loudly, it never falls back


def test_strerror_and_version(pcr):
    L = pcr.lib()
    assert L.pcr_strerror(0) == b"ok" and L.pcr_strerror(-5).startswith(b"no correspondence")
    assert b"gfx950" in L.pcr_version()


def test_solve_rigid_host_only(pcr):
    """pcr_solve_rigid is pure host code: recover a known transform from exact partial sums."""
    from pointcloudregistration_amd import synth
    rng = np.random.default_rng(0)
    a = rng.standard_normal((300, 3)) + [5.0, -2.0, 1.0]
    Rm, t = synth.euler_zyx(20.0, 10.0, -30.0), np.array([1.0, 2.0, -0.5])
    b = a @ Rm.T + t
    pp, pq = a.mean(0) + 0.1, b.mean(0) - 0.2
    A, B = a - pp, b - pq
    partials = np.concatenate([[len(a)], A.sum(0), B.sum(0), (A[:, :, None] * B[:, None, :]).sum(0).reshape(9),
                               [((a - b) ** 2).sum()]])
    R, T, Rc, Tc, mse = pcr.solve_rigid(partials, pp, pq, np.eye(3).reshape(9), np.zeros(3))
    assert np.abs(synth.rowcol(R) - Rm).max() < 1e-12 and np.abs(T - t).max() < 1e-11
    assert abs(mse - ((a - b) ** 2).sum() / len(a)) < 1e-12
    # composition on the left: R <- Rc*R, T <- Rc*T + Tc (BaseTransformationEstimation.h:28-32)
    R2, T2, _, _, _ = pcr.solve_rigid(partials, pp, pq, synth.colmajor(Rm), t)
    assert np.abs(synth.rowcol(R2) - Rm @ Rm).max() < 1e-12 and np.abs(T2 - (Rm @ t + t)).max() < 1e-11
    # degenerate (planar, mirrored) input still yields a proper rotation
    a2 = a.copy()
    a2[:, 2] = 0.0
    b2 = a2 * [1, 1, -1]
    A, B = a2 - pp, b2 - pq
    partials = np.concatenate([[len(a)], A.sum(0), B.sum(0), (A[:, :, None] * B[:, None, :]).sum(0).reshape(9), [0.0]])
    R3, _, _, _, _ = pcr.solve_rigid(partials, pp, pq, np.eye(3).reshape(9), np.zeros(3))
    assert abs(np.linalg.det(synth.rowcol(R3)) - 1.0) < 1e-12
    with pytest.raises(pcr.PcrError) as e:
        pcr.solve_rigid(np.zeros(17), pp, pq, np.eye(3).reshape(9), np.zeros(3))
    assert e.value.status == -5


def test_struct_layouts_of_the_binding_match_the_header(pcr, tmp_path):
    """The ctypes mirrors of the structs that cross the ABI are laid out exactly as a C compiler lays out the header's:
    a small C program prints sizeof/offsetof, compared field by field (catches a field added on one side only)."""
    import ctypes
    import subprocess
    structs = {"pcr_index_params": pcr.IndexParams, "pcr_index_info": pcr.IndexInfo,
               "pcr_convergence_params": pcr.ConvergenceParams}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pcr.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
