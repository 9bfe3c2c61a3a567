"""Seeded synthetic clouds for the parity tests and the benchmark.

Everything is numpy `default_rng(seed)` so a fixture can be regenerated from its seed on the GPU
box (which has no dataset and no network).  The reference's own synthetic recipe is "known rigid
transform + uniform noise" (unit test/TestRegManDirAndClosestPoints/TestRegManDirAndClosestPoints.cpp:
101-107); the clouds here follow it with Gaussian noise, as BASELINE.json's configs ask.
"""
import numpy as np


def euler_zyx(rx_deg, ry_deg, rz_deg):
    """R = Rz * Ry * Rx (row, col) -- the ZYX Euler matrix of unit test/TestUnitity.cpp:132-163."""
    rx, ry, rz = np.deg2rad([rx_deg, ry_deg, rz_deg])
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def colmajor(Rm):
    """3x3 (row, col) matrix -> the 9-vector layout of Eigen::Matrix3d::data()."""
    return np.ascontiguousarray(np.asarray(Rm, np.float64).T).reshape(9)


def rowcol(R9):
    return np.asarray(R9, np.float64).reshape(3, 3).T.copy()


_SPHERES = ((np.array([0.0, 0.0, 0.0]), 1.0), (np.array([1.3, 0.2, 0.1]), 0.6), (np.array([-0.3, 1.2, 0.4]), 0.4))


def three_spheres(n, seed):
    """n points, area-uniform on the union of three sphere surfaces (radii 1.0/0.6/0.4): a closed,
    scan-like 2-D manifold in 3-D."""
    rng = np.random.default_rng(seed)
    area = np.array([r * r for _, r in _SPHERES])
    which = rng.choice(len(_SPHERES), size=n, p=area / area.sum())
    v = rng.standard_normal((n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    c = np.stack([_SPHERES[w][0] for w in range(len(_SPHERES))])[which]
    r = np.array([s[1] for s in _SPHERES])[which][:, None]
    return np.ascontiguousarray(c + r * v)


def mean_spacing(n):
    """Typical nearest-neighbour spacing of n area-uniform points on the three-sphere surface."""
    area = 4 * np.pi * sum(r * r for _, r in _SPHERES)
    return float(np.sqrt(area / n))


def registration_pair(n_src, n_tgt, seed, rot_deg=(3.0, -5.0, 8.0), trans=(0.05, -0.03, 0.02), noise=0.25):
    """(src, tgt, R_true 3x3, t_true): tgt sampled on the surface; src = an independent sampling moved
    by the INVERSE of the known transform plus Gaussian noise (sigma = noise * mean spacing), so that
    R_true * src + t_true lies on the target surface."""
    tgt = three_spheres(n_tgt, seed)
    s = three_spheres(n_src, seed + 1)
    rng = np.random.default_rng(seed + 2)
    s = s + rng.standard_normal(s.shape) * (noise * mean_spacing(n_src))
    Rm = euler_zyx(*rot_deg)
    t = np.asarray(trans, np.float64)
    src = (s - t) @ Rm  # = R^T (s - t), row-vector form
    return np.ascontiguousarray(src), tgt, Rm, t


def cube_surface(n, seed):
    """n points uniform on the surface of the unit cube (used by the 4 096-point fixture)."""
    rng = np.random.default_rng(seed)
    face = rng.integers(0, 6, n)
    uv = rng.random((n, 2))
    p = np.empty((n, 3))
    ax = face // 2
    side = (face % 2).astype(np.float64)
    for a in range(3):
        m = ax == a
        o = [i for i in range(3) if i != a]
        p[m, a] = side[m]
        p[m, o[0]] = uv[m, 0]
        p[m, o[1]] = uv[m, 1]
    return p


def lidar_like(n, seed, outlier_frac=0.0, extent=(200.0, 200.0, 30.0)):
    """LiDAR-like scene: ground plane + boxes, density falling off with range
    from a sensor at the origin; a fraction of points replaced by uniform-in-volume outliers."""
    rng = np.random.default_rng(seed)
    ex, ey, ez = extent
    # range-dependent density ~ 1/r^2 from the sensor: r log-uniform between 1 m and the half extent
    rmax = 0.5 * min(ex, ey)
    r = np.exp(rng.random(n) * np.log(rmax))
    th = rng.random(n) * 2 * np.pi
    p = np.stack([r * np.cos(th), r * np.sin(th), np.zeros(n)], 1)
    # a third of the points climb the walls of axis-aligned boxes on a 25 m lattice
    wall = rng.random(n) < 0.33
    bx = np.round(p[wall, 0] / 25.0) * 25.0
    p[wall, 0] = bx + 4.0 * np.sign(p[wall, 0] - bx + 1e-9)
    p[wall, 2] = rng.random(int(wall.sum())) * 0.6 * ez
    p[:, 2] += 0.02 * np.sin(p[:, 0] * 0.3) + 0.02 * np.cos(p[:, 1] * 0.2)
    if outlier_frac > 0:
        m = rng.random(n) < outlier_frac
        k = int(m.sum())
        p[m] = (rng.random((k, 3)) - [0.5, 0.5, 0.0]) * [ex, ey, ez]
    return np.ascontiguousarray(p)


def lidar_pair(n, seed, outlier_frac=0.3, rot_deg=(0.4, -0.3, 1.5), trans=(0.35, -0.2, 0.05), noise=0.01):
    """C5: two LiDAR-like scans of the same scene, the source moved by the inverse of a known
    small rigid motion (a vehicle's step between two sweeps) + Gaussian range noise; `outlier_frac` of the SOURCE
    points are uniform-in-volume outliers.  Returns (src, tgt, R_true, t_true) with tgt ~= R_true*src + t_true."""
    tgt = lidar_like(n, seed, 0.0)
    s = lidar_like(n, seed + 1, 0.0)
    rng = np.random.default_rng(seed + 2)
    s += rng.standard_normal(s.shape) * noise
    Rm, t = euler_zyx(*rot_deg), np.asarray(trans, np.float64)
    src = (s - t) @ Rm
    if outlier_frac > 0:
        m = rng.random(n) < outlier_frac
        src[m] = (rng.random((int(m.sum()), 3)) - [0.5, 0.5, 0.0]) * [200.0, 200.0, 30.0]
    return np.ascontiguousarray(src), tgt, Rm, t


def bumpy_surface(n, seed):
    """Area-ish-uniform samples of a closed bumpy surface r(u) = 1 + bumps(u): the stand-in for a statue scan."""
    rng = np.random.default_rng(seed)
    u = rng.standard_normal((n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    r = 1.0 + 0.12 * np.sin(3.0 * u[:, 0]) * np.cos(2.0 * u[:, 1]) + 0.08 * np.sin(5.0 * u[:, 2] + 1.0) + 0.05 * np.cos(7.0 * u[:, 0] * u[:, 1])
    return u * r[:, None]


def scan_sequence(num_scans, n, seed, step_deg=2.0, crop_deg=100.0, noise=0.0005):
    """C2/C4 stand-in: `num_scans` partial views of one bumpy object.  Scan i sees the part of the
    surface within crop_deg of its viewing direction (which turns by 24 degrees per scan, so neighbours overlap
    partially), is sampled independently, and is expressed in its own frame, which differs from the previous
    scan's by a small known motion.  Returns (scans, poses) with x_world = R_i x_scan + t_i."""
    scans, poses = [], []
    Rw, tw = np.eye(3), np.zeros(3)
    for i in range(num_scans):
        if i:
            dR = euler_zyx(step_deg * np.cos(i), step_deg * np.sin(1.3 * i), step_deg * 0.7)
            Rw, tw = Rw @ dR, tw + Rw @ np.array([0.01 * np.cos(i), 0.008 * np.sin(i), 0.004])
        w = bumpy_surface(int(n * 360.0 / crop_deg / 2.0), seed + i)
        view = np.array([np.cos(np.deg2rad(24.0 * i)), np.sin(np.deg2rad(24.0 * i)), 0.2])
        view /= np.linalg.norm(view)
        cosang = (w / np.linalg.norm(w, axis=1, keepdims=True)) @ view
        w = w[cosang > np.cos(np.deg2rad(crop_deg))][:n]
        w = w + np.random.default_rng(seed + 100 + i).standard_normal(w.shape) * noise
        scans.append(np.ascontiguousarray((w - tw) @ Rw))   # x_scan = R^T (x_world - t)
        poses.append((Rw.copy(), tw.copy()))
    return scans, poses


def coarse_initial_transform(R_true, t_true, n, rot_spacings=1.5, trans_spacings=1.0):
    """(R0 col-major 9, T0): the known transform composed with a small rigid perturbation, sized in units of
    the cloud's mean point spacing.  ICRP is a FINE registration: its gates are a few spacings wide
    (sqrt(36 U)), so -- as in the reference, where coarseFit precedes fineFit (Registration.cpp:120-158) --
    the loop is started from a coarse alignment, not from the identity."""
    h = mean_spacing(n)
    ang = np.rad2deg(rot_spacings * h / 1.0)  # displacement of ~rot_spacings*h at unit radius
    dR = euler_zyx(0.3 * ang, -0.5 * ang, 0.8 * ang)
    dt = np.array([0.6, -0.5, 0.4]) * trans_spacings * h
    R0 = dR @ R_true
    T0 = dR @ t_true + dt
    return colmajor(R0), T0
