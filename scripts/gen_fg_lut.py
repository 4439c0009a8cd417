#!/usr/bin/env python3
"""Generate the split-sum FG (scale/bias) lookup table shipped with geosplatting_amd.

The reference loads an asset `rfstudio/assets/geometry/pbr/bsdf_256_256.bin`
(rfstudio/graphics/shaders.py:22-26).  That file is reference data and is NOT copied;
this script integrates the standard split-sum BRDF term itself:
    GGX half-vector distribution, height-correlated Smith visibility, Schlick Fresnel
    split  ->  (A, B) with  spec = F0*A + B
on a 256x256 grid, column = N.V at (x+0.5)/256, row = roughness at (y+0.5)/256, and
writes float32 [256,256,2] to geosplatting_amd/assets/fg_lut_256.bin.

Round 4: DETERMINISTIC float64 quadrature instead of 16 384 Hammersley samples per texel.
scripts/fg_lut_study.py showed that the sampled table was the outlier (3e-3 off at N.V -> 0 and
along a band of sampling artefacts) while the reference asset agrees with a converged integral to
2.4e-4.  Per texel: midpoint rule over (phi, t), xi = sin^2(pi t / 2) (tan theta_h = a tan(pi t / 2):
smooth map, Jacobian vanishing at both ends), resolution doubled until two successive levels agree to
2e-5 (the N.V -> 0 columns, where the visibility term peaks within |N.L| < N.V of the horizon, need
up to 4096 x 16384 points).  tests/test_oracle_cpu.py checks the result against a 16x16 sub-sample
of the reference asset (tests/golden/ref_fg_lut_sub16.npz).
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

TOL = 2e-5
MAX_NPHI = 4096


def texel_at(nv: float, rough: float, nphi: int, nt: int):
    """(A, B) at (N.V, roughness) with an nphi x nt midpoint rule (phi over [0, pi] by symmetry), float64."""
    a = rough * rough
    a2 = a * a
    vx, vz = np.sqrt(max(1.0 - nv * nv, 0.0)), nv
    cosp = np.cos((np.arange(nphi) + 0.5) / nphi * np.pi)[:, None]
    accA = accB = 0.0
    chunk = max(1, (1 << 19) // nphi)
    k1 = nv * nv * (1.0 - a2) + a2
    for c0 in range(0, nt, chunk):
        t = ((np.arange(c0, min(nt, c0 + chunk)) + 0.5) / nt)[None, :]
        jac = 0.5 * np.pi * np.sin(np.pi * t)
        theta = np.arctan(a * np.tan(0.5 * np.pi * t))
        hz = np.cos(theta)
        hx = np.sin(theta) * cosp
        VoH = vx * hx + vz * hz
        Lz = 2.0 * VoH * hz - vz
        NoL = np.clip(Lz, 0.0, 1.0)
        VoHc = np.clip(VoH, 0.0, 1.0)
        vis = 0.5 / np.maximum(NoL * np.sqrt(k1) + nv * np.sqrt(NoL * NoL * (1.0 - a2) + a2), 1e-300)
        gv = np.where(Lz > 0.0, vis * 4.0 * NoL * VoHc / np.maximum(hz, 1e-300), 0.0) * jac
        fc = (1.0 - VoHc) ** 5
        accB += (fc * gv).sum()
        accA += gv.sum()
    n = nphi * nt
    return np.array([(accA - accB) / n, accB / n])


def texel(nv: float, rough: float):
    """Resolution doubled until two successive levels agree to TOL; returns (value, levels used, last difference)."""
    nphi = 128                                  # (first comparison: 128 x 512 against 256 x 1024)
    prev = texel_at(nv, rough, nphi, 4 * nphi)
    while True:
        nphi *= 2
        cur = texel_at(nv, rough, nphi, 4 * nphi)
        diff = float(np.abs(cur - prev).max())
        if diff < TOL or nphi >= MAX_NPHI:
            return cur, nphi, diff
        prev = cur


def _row(yi, res=256):
    out = np.zeros((res, 2))
    worst, deepest = 0.0, 0
    for xi in range(res):
        v, nphi, diff = texel((xi + 0.5) / res, (yi + 0.5) / res)
        out[xi] = v
        worst, deepest = max(worst, diff), max(deepest, nphi)
    return yi, out, worst, deepest


def generate(res=256, procs=None):
    out = np.zeros((res, res, 2))
    worst, deepest = 0.0, 0
    with Pool(procs or os.cpu_count()) as pool:
        for yi, row, w, d in pool.imap_unordered(_row, range(res)):
            out[yi] = row
            worst, deepest = max(worst, w), max(deepest, d)
            print(f"row {yi:3d}: last refinement step <= {w:.1e}, finest rule {d} x {4 * d}", file=sys.stderr, flush=True)
    print(f"largest last refinement step {worst:.2e} (error of the accepted level ~ a third of it), finest rule {deepest} x {4 * deepest}",
          file=sys.stderr)
    return out.astype(np.float32)


if __name__ == "__main__":
    lut = generate()
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "geosplatting_amd", "assets", "fg_lut_256.bin")
    lut.tofile(dst)
    print("wrote", os.path.normpath(dst), lut.shape, lut.min(), lut.max())
