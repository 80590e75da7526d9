"""Seeded synthetic MIMC3 inputs (image pair + xyuvav grid) -- SURVEY.md section 8(d).

There is no network and the reference ships no sample data (README.md:33 links a Google Drive),
so every test / bench input is generated here: a Gaussian-filtered PCG64 noise texture quantised
to integer DN 1..255 (what GMA_float_load_tiff produces for an 8-bit TIFF, GMA.c:288-310), the
second image being the first translated by a known shift, optionally with a sub-pixel (bilinear)
component, +-noise_dn DN noise and zeroed "null" blobs (DN < 1e-10 is null, MIMC_module.c:21).

The named configurations follow BASELINE.json / SURVEY.md section 8 table:
  C1  512^2,    1,024 pts (32x32, spacing 12, margin 70), ocw 16, shift (+4,-4), 45 deg 1806 m/yr
  C2  4096^2, 200,000 pts (500x400, spacing 8x10, margin 52), ocw 16, same flow
  C4  8192^2, 1,000,000 pts (1000x1000, spacing 8, margin 100), ocw 32, shift (+12,-12), 6101 m/yr
"""
from dataclasses import dataclass, field

import numpy as np
from scipy import ndimage

MPP = 15.0      # metres per pixel        (MIMC_main.c:154, re-derived from xyuvav rows 0-1 at :221)
DT_DAYS = 16.0  # temporal baseline, days (MIMC_misc.c:121-131 in the real CLI)
AW_CRE = 10.0   # MIMC_main.c:152
AW_SF = 1.8     # MIMC_main.c:153


@dataclass
class Case:
    name: str
    i0: np.ndarray            # [H][W] float32
    i1: np.ndarray            # [H][W] float32
    xyuvav: np.ndarray        # [N][6] float64: x, y, u, v, vx, vy
    dimx: int
    dimy: int
    ocw: int
    offset: np.ndarray        # int32[2], CP offset added to the search centre only
    shift: tuple              # true (du, dv) of i1 relative to i0, pixels
    dt: float = DT_DAYS
    mpp: float = MPP
    meta: dict = field(default_factory=dict)

    @property
    def n(self):
        return self.xyuvav.shape[0]


def texture(h, w, seed, sigma=2.0, bits=8):
    """Integer-valued float32 texture in [1, 2**bits-1]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = rng.standard_normal((h, w), dtype=np.float32)
    g = ndimage.gaussian_filter(g, sigma=sigma, mode="wrap")
    g -= g.mean()
    g /= g.std()
    np.clip(g, -3.0, 3.0, out=g)
    top = float(2 ** bits - 2)
    out = np.rint(1.0 + top * (g + 3.0) / 6.0).astype(np.float32)
    return out


def _blobs(img, frac, rng, radius=9):
    """Zero ~frac of the area in discs (null pixels)."""
    h, w = img.shape
    if frac <= 0:
        return
    nblob = max(1, int(frac * h * w / (np.pi * radius * radius)))
    yy, xx = np.mgrid[-radius:radius + 1, -radius:radius + 1]
    disc = (yy * yy + xx * xx) <= radius * radius
    cy = rng.integers(radius, h - radius, nblob)
    cx = rng.integers(radius, w - radius, nblob)
    for y, x in zip(cy, cx):
        sub = img[y - radius:y + radius + 1, x - radius:x + radius + 1]
        sub[disc] = 0.0


def make_pair(h, w, shift, seed, subpixel=(0.0, 0.0), noise_dn=0, null_frac=0.0, bits=8, pad=64, sigma=2.0):
    """i1[v,u] = i0[v-dv, u-du]: a feature at (u,v) in i0 sits at (u+du, v+dv) in i1."""
    du, dv = int(shift[0]), int(shift[1])
    assert abs(du) < pad - 1 and abs(dv) < pad - 1
    base = texture(h + 2 * pad, w + 2 * pad, seed, sigma=sigma, bits=bits)
    i0 = base[pad:pad + h, pad:pad + w].copy()
    fu, fv = float(subpixel[0]), float(subpixel[1])
    if fu == 0.0 and fv == 0.0:
        i1 = base[pad - dv:pad - dv + h, pad - du:pad - du + w].copy()
    else:
        assert 0.0 <= fu < 1.0 and 0.0 <= fv < 1.0
        # sample base at (v - dv - fv, u - du - fu): bilinear between the 4 integer-shift copies
        def crop(sy, sx):
            return base[pad - dv - sy:pad - dv - sy + h, pad - du - sx:pad - du - sx + w]
        i1 = ((1 - fv) * (1 - fu) * crop(0, 0) + (1 - fv) * fu * crop(0, 1)
              + fv * (1 - fu) * crop(1, 0) + fv * fu * crop(1, 1))
        i1 = np.rint(i1).astype(np.float32)
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    if noise_dn > 0:
        i1 = i1 + rng.integers(-noise_dn, noise_dn + 1, i1.shape).astype(np.float32)
        np.clip(i1, 1.0, float(2 ** bits - 1), out=i1)
    if null_frac > 0:
        _blobs(i0, null_frac, rng)
        _blobs(i1, null_frac, rng)
    return np.ascontiguousarray(i0, np.float32), np.ascontiguousarray(i1, np.float32)


def make_grid(dimx, dimy, u0, v0, su, sv, speed, angle_deg=45.0, perturb=0.1, mpp=MPP):
    """xyuvav [N][6], x fastest (MIMC_main.c:211-219 derives dimx from the first repeat of u)."""
    gu = u0 + su * np.arange(dimx, dtype=np.float64)
    gv = v0 + sv * np.arange(dimy, dtype=np.float64)
    uu, vv = np.meshgrid(gu, gv)                       # [dimy][dimx]
    ix, iy = np.meshgrid(np.arange(dimx), np.arange(dimy))
    mod = 1.0 + perturb * np.sin(2 * np.pi * ix / max(dimx, 2)) * np.cos(2 * np.pi * iy / max(dimy, 2))
    ang = np.deg2rad(angle_deg)
    vx = speed * np.cos(ang) * mod
    vy = speed * np.sin(ang) * mod
    xy = np.stack([uu * mpp, -vv * mpp, uu, vv, vx, vy], axis=-1).reshape(-1, 6)
    return np.ascontiguousarray(xy, np.float64)


_CONFIGS = {
    #        H     W     dimx  dimy  u0   v0   su  sv  ocw shift      speed
    "C1": (512, 512, 32, 32, 70, 70, 12, 12, 16, (4, -4), 1806.0),
    "C2": (4096, 4096, 500, 400, 52, 52, 8, 10, 16, (4, -4), 1806.0),
    "C4": (8192, 8192, 1000, 1000, 100, 100, 8, 8, 32, (12, -12), 6101.0),
}


def make_case(name, seed=None, noise_dn=2, null_frac=0.02, subpixel=(0.0, 0.0), perturb=0.1):
    """Named configuration of BASELINE.json (C3/C5 share C2's inputs)."""
    key = {"C3": "C2", "C5": "C2"}.get(name, name)
    h, w, dimx, dimy, u0, v0, su, sv, ocw, shift, speed = _CONFIGS[key]
    if seed is None:
        seed = 20260101 + int(key[1:])
    i0, i1 = make_pair(h, w, shift, seed, subpixel=subpixel, noise_dn=noise_dn, null_frac=null_frac)
    xy = make_grid(dimx, dimy, u0, v0, su, sv, speed, perturb=perturb)
    return Case(name, i0, i1, xy, dimx, dimy, ocw, np.zeros(2, np.int32), shift,
                meta=dict(seed=seed, noise_dn=noise_dn, null_frac=null_frac, subpixel=subpixel, speed=speed))


def make_small(h=160, w=176, dimx=9, dimy=8, ocw=7, shift=(3, -2), speed=1200.0, angle_deg=30.0, seed=11,
               offset=(0, 0), margin=None, **pair_kw):
    """Small free-form case for parity tests."""
    if margin is None:
        margin = ocw + 30
    su = max(1, (w - 2 * margin) // max(dimx - 1, 1))
    sv = max(1, (h - 2 * margin) // max(dimy - 1, 1))
    i0, i1 = make_pair(h, w, shift, seed, **pair_kw)
    xy = make_grid(dimx, dimy, margin, margin, su, sv, speed, angle_deg=angle_deg)
    return Case(f"small{seed}", i0, i1, xy, dimx, dimy, ocw, np.asarray(offset, np.int32), tuple(shift),
                meta=dict(seed=seed, speed=speed, angle_deg=angle_deg, **{k: v for k, v in pair_kw.items()}))


def synth_candidates(dimx, dimy, seed=5, k=8, p_out=0.45, sigma_in=0.06, out_amp=5.0):
    """Directly synthesised matcher outputs dp[k][N][3] around a smooth true field (survey probe,
    SURVEY.md section 8d): inliers N(true, sigma_in), outliers uniform +-out_amp, ncc in (0.2,1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = dimx * dimy
    ix, iy = np.meshgrid(np.arange(dimx), np.arange(dimy))
    tu = (4.0 + 1.5 * np.sin(2 * np.pi * ix / dimx) * np.cos(2 * np.pi * iy / dimy)).reshape(-1)
    tv = (-4.0 + 1.5 * np.cos(2 * np.pi * ix / dimx)).reshape(-1)
    dp = np.empty((k, n, 3), np.float32)
    for j in range(k):
        outl = rng.random(n) < p_out
        du = np.where(outl, tu + rng.uniform(-out_amp, out_amp, n), tu + rng.normal(0, sigma_in, n))
        dv = np.where(outl, tv + rng.uniform(-out_amp, out_amp, n), tv + rng.normal(0, sigma_in, n))
        dp[j, :, 0] = du
        dp[j, :, 1] = dv
        dp[j, :, 2] = rng.uniform(0.2, 1.0, n)
    return dp


def synth_qm_state(dimx, dimy, seed=5, k=8, p_out=0.45, sigma_in=0.06, out_amp=5.0, p_wrong=0.3):
    """Vectorised synthetic QM input for large grids (BASELINE config C5): per grid point one inlier
    cluster (mean of the inlier candidates around a smooth true field) plus one singleton cluster
    per outlier candidate, padded to [N][k][5]; the initial pick `dpf` is the inlier cluster when its
    fraction exceeds 0.6 (get_dpf0's rule, MIMC_module.c:913) and otherwise the inlier cluster with
    probability 1-p_wrong or a random outlier (a wrong snap that the pseudo-smoothing must repair).
    Returns (mvn [N][k][5] f32, nclus [N] i32, dpf [dimy][dimx] i32, dx, dy [dimy][dimx] f32)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = dimx * dimy
    ix, iy = np.meshgrid(np.arange(dimx), np.arange(dimy))
    tu = (4.0 + 1.5 * np.sin(2 * np.pi * ix / dimx) * np.cos(2 * np.pi * iy / dimy)).reshape(-1)
    tv = (-4.0 + 1.5 * np.cos(2 * np.pi * ix / dimx)).reshape(-1)
    outl = rng.random((n, k)) < p_out
    n_out = outl.sum(1)
    n_in = k - n_out
    mvn = np.zeros((n, k, 5), np.float32)
    has_in = n_in > 0
    sd = sigma_in / np.sqrt(np.maximum(n_in, 1))
    mvn[:, 0, 0] = np.where(has_in, tu + rng.normal(0, 1, n) * sd, 0)
    mvn[:, 0, 1] = np.where(has_in, tv + rng.normal(0, 1, n) * sd, 0)
    mvn[:, 0, 2] = np.where(has_in, sigma_in ** 2 * (1 - 1 / np.maximum(n_in, 1)), 0)
    mvn[:, 0, 3] = mvn[:, 0, 2]
    mvn[:, 0, 4] = n_in / k
    # outlier singletons occupy slots first_out .. first_out + n_out - 1
    first_out = has_in.astype(np.int64)
    rank = np.cumsum(outl, 1) - 1                                  # index of each outlier among the point's outliers
    slot = first_out[:, None] + rank
    pi, pj = np.nonzero(outl)
    ou = tu[pi] + rng.uniform(-out_amp, out_amp, pi.size)
    ov = tv[pi] + rng.uniform(-out_amp, out_amp, pi.size)
    mvn[pi, slot[pi, pj], 0] = ou
    mvn[pi, slot[pi, pj], 1] = ov
    mvn[pi, slot[pi, pj], 4] = 1.0 / k
    nclus = (first_out + n_out).astype(np.int32)
    dpf = np.zeros(n, np.int32)
    wrong = (mvn[:, 0, 4] <= 0.6) & (n_out > 0) & ((rng.random(n) < p_wrong) | ~has_in)
    pick = first_out + (rng.integers(0, 1 << 30, n) % np.maximum(n_out, 1))
    dpf[wrong] = pick[wrong]
    dx = mvn[np.arange(n), dpf, 0].reshape(dimy, dimx).copy()
    dy = mvn[np.arange(n), dpf, 1].reshape(dimy, dimx).copy()
    return mvn, nclus, dpf.reshape(dimy, dimx), dx, dy
