"""Shape-only constant tables of the hot path, built once on the host (numpy) and uploaded.

The reference rebuilds all of these on the host at EVERY forward (rotation matrices and grids in
GroupConv.trans_filter src/models.py:178-195; offsets / rotation prior models.py:361-379; translation
grid and its prior train_mnist.py:209-218,258-262).  They depend only on shapes, so they are hoisted
to construction time here.
"""
from __future__ import annotations

import math
from functools import lru_cache

import numpy as np


@lru_cache(maxsize=None)
def rotation_taps(k: int, R: int):
    """Bilinear taps of the R fixed filter rotations (2-D form of affine_grid + grid_sample,
    align_corners=False, zeros padding; src/models.py:174-197).

    Returns idx int32 [R][k*k][4] (-1 = outside) and w float32 [R][k*k][4].  Output pixel centres are
    (2i+1)/k - 1; source = R(theta_r) * target with theta_r = r*2pi/R accumulated in float64 like the
    reference loop (models.py:195); pixel coordinate ((v+1)*k - 1)/2.  Arithmetic in float32 as ATen does.
    """
    f32 = np.float32
    lin = (np.linspace(-1.0, 1.0, k).astype(f32) * f32((k - 1) / k)).astype(f32)
    xt = np.broadcast_to(lin[None, :], (k, k))
    yt = np.broadcast_to(lin[:, None], (k, k))
    idx = np.full((R, k * k, 4), -1, dtype=np.int32)
    wgt = np.zeros((R, k * k, 4), dtype=f32)
    theta = 0.0
    for r in range(R):
        c, s = f32(math.cos(theta)), f32(math.sin(theta))
        gx = (c * xt + s * yt).astype(f32)
        gy = (f32(-math.sin(theta)) * xt + c * yt).astype(f32)
        px = ((gx + f32(1)) * f32(k) - f32(1)) / f32(2)
        py = ((gy + f32(1)) * f32(k) - f32(1)) / f32(2)
        x0 = np.floor(px)
        y0 = np.floor(py)
        fx = (px - x0).astype(f32)
        fy = (py - y0).astype(f32)
        x0 = x0.astype(np.int64)
        y0 = y0.astype(np.int64)
        t = 0
        for dy in (0, 1):
            wy = fy if dy else (f32(1) - fy)
            for dx in (0, 1):
                wx = fx if dx else (f32(1) - fx)
                yy, xx = y0 + dy, x0 + dx
                ok = (yy >= 0) & (yy < k) & (xx >= 0) & (xx < k)
                idx[r, :, t] = np.where(ok, yy * k + xx, -1).reshape(-1)
                wgt[r, :, t] = np.where(ok, (wy * wx).astype(f32), f32(0)).reshape(-1)
                t += 1
        theta += 2 * math.pi / R
    return idx, wgt


@lru_cache(maxsize=None)
def rotation_taps_csr(k: int, R: int):
    """Transposed tap table in CSR form keyed by SOURCE pixel, for the deterministic gather backward.

    Returns ptr int32 [k*k+1], ent_r int32 [nnz], ent_dst int32 [nnz], ent_w float32 [nnz].
    """
    idx, wgt = rotation_taps(k, R)
    r_i, d_i, t_i = np.nonzero((idx >= 0) & (wgt != 0))
    src = idx[r_i, d_i, t_i].astype(np.int64)
    order = np.lexsort((t_i, d_i, r_i, src))      # fixed order inside every source row
    src, r_i, d_i, t_i = src[order], r_i[order], d_i[order], t_i[order]
    ptr = np.zeros(k * k + 1, dtype=np.int32)
    np.add.at(ptr, src + 1, 1)
    ptr = np.cumsum(ptr).astype(np.int32)
    return ptr, r_i.astype(np.int32), d_i.astype(np.int32), wgt[r_i, d_i, t_i].astype(np.float32)


def rotation_offsets(R: int, rot_refinement: bool) -> np.ndarray:
    """Angle offsets per rotation (src/models.py:361-366); zeros without refinement (models.py:401)."""
    if not rot_refinement:
        return np.zeros(R, dtype=np.float32)
    half = R // 2
    return np.asarray([(r * np.pi / half) if r <= half else ((r - R) * np.pi / half) for r in range(R)],
                      dtype=np.float64).astype(np.float32)


def rotation_log_prior(R: int, rot_refinement: bool, theta_prior: float, normal_prior_over_r: bool) -> np.ndarray:
    """log p(r) (src/models.py:368-379): Normal(0, theta_prior) or Uniform(-2pi, 2pi) at the offsets when
    refining, else -log R.  float32 (R,), evaluated in float32 like torch.distributions does."""
    f32 = np.float32
    if not rot_refinement:
        return (np.zeros(R, dtype=f32) - f32(np.log(R))).astype(f32)
    if normal_prior_over_r:
        off = rotation_offsets(R, True)
        sd = f32(theta_prior)
        var = sd * sd
        return (-((off - f32(0)) ** 2) / (f32(2) * var) - f32(math.log(float(sd))) -
                f32(math.log(math.sqrt(2 * math.pi)))).astype(f32)
    return np.full(R, -np.log(np.float32(4 * np.pi)), dtype=f32)


def translation_grid(Ho: int, spacing: float) -> np.ndarray:
    """float64 (Ho*Ho, 2): candidate translations (train_mnist.py:209-218): x ascending along w, y descending
    along h, step = float32 pixel spacing of the image coordinates (train_mnist.py:30).  Built from integer
    index * step."""
    s = np.float64(np.float32(spacing))
    g = (np.arange(Ho, dtype=np.float64) - (Ho // 2)) * s
    x0, x1 = np.meshgrid(g, g[::-1])
    return np.stack([x0.ravel(), x1.ravel()], 1)


def joint_log_prior(Ho: int, spacing: float, p_r: np.ndarray, dx_std: float = 0.1) -> np.ndarray:
    """log-softmax over (r,h,w) of log N(t; 0, 0.1) + log p(r) (train_mnist.py:258-262), float64 -> float32."""
    G = translation_grid(Ho, spacing)
    sd = np.float64(np.float32(dx_std))
    logn = -(G ** 2) / (2 * sd * sd) - np.log(sd) - math.log(math.sqrt(2 * math.pi))
    p_t = logn.sum(1)                                           # (Ho*Ho,)
    joint = (p_t[None, :] + p_r.astype(np.float64)[:, None]).reshape(-1)
    m = joint.max()
    lse = m + np.log(np.exp(joint - m).sum())
    return (joint - lse).astype(np.float32)


def image_coords(n: int) -> np.ndarray:
    """x_coord (n*n, 2) float32 (train_mnist.py:475-479)."""
    xg = np.linspace(-1, 1, n)
    yg = np.linspace(1, -1, n)
    x0, x1 = np.meshgrid(xg, yg)
    return np.stack([x0.ravel(), x1.ravel()], 1).astype(np.float32)
