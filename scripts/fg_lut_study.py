#!/usr/bin/env python3
"""Which FG table is the outlier?  (container-only diagnostic)

The packaged split-sum table (scripts/gen_fg_lut.py, 16 384 Hammersley samples per texel) and the reference's asset
(rfstudio/assets/geometry/pbr/bsdf_256_256.bin, read by rfstudio/graphics/shaders.py:22-26) differ by up to 3e-3 at
grazing angles.  This script integrates the same split-sum term in float64 with a DETERMINISTIC tensor-product midpoint
rule over (phi, t), xi = sin^2(pi t / 2) -- >= 1 M points per texel, no random or low-discrepancy sequence involved -- on a 16x16
sub-sample of the grid, and prints the error of BOTH tables against it.

    python scripts/fg_lut_study.py [reference_asset.bin] > profiles/r04_fg_lut_study.txt

Only numbers are written; the asset itself is never copied.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def exact_texel(nv: float, rough: float, nphi: int = 1024, nt: int = 4096):
    """(A, B) of the split-sum BRDF term at (N.V, roughness): GGX importance sampling written as an integral over the unit square
    (xi1 -> phi, xi2 -> theta_h: tan^2 theta_h = a^2 xi2 / (1 - xi2)), height-correlated Smith visibility, Schlick split, in
    float64.  The plain midpoint rule in xi2 converges slowly (cos theta_h has an infinite slope at xi2 -> 1: 2e-3 off at 4 M
    points for N.V -> 0), so xi2 = sin^2(pi t / 2), i.e. tan theta_h = a tan(pi t / 2): the map is smooth, its Jacobian
    (pi / 2) sin(pi t) vanishes at both ends, and the midpoint rule in t is limited only by the kink where L leaves the
    hemisphere.  By symmetry phi only needs [0, pi]."""
    a = rough * rough
    a2 = a * a
    V = np.array([np.sqrt(max(1.0 - nv * nv, 0.0)), 0.0, nv])
    phi = (np.arange(nphi) + 0.5) / nphi * np.pi
    accA = accB = 0.0
    cosp = np.cos(phi)[:, None]
    chunk = 512
    for c0 in range(0, nt, chunk):
        t = ((np.arange(c0, min(nt, c0 + chunk)) + 0.5) / nt)[None, :]
        jac = 0.5 * np.pi * np.sin(np.pi * t)
        theta = np.arctan(a * np.tan(0.5 * np.pi * t))
        cos_t, sin_t = np.cos(theta), np.sin(theta)
        hx = sin_t * cosp
        hz = np.broadcast_to(cos_t, hx.shape)
        VoH = V[0] * hx + V[2] * hz
        Lz = 2.0 * VoH * hz - V[2]
        NoL = np.clip(Lz, 0.0, 1.0)
        VoHc = np.clip(VoH, 0.0, 1.0)
        vis = 0.5 / np.maximum(NoL * np.sqrt(nv * nv * (1.0 - a2) + a2) + nv * np.sqrt(NoL * NoL * (1.0 - a2) + a2), 1e-300)
        gv = vis * 4.0 * NoL * VoHc / np.maximum(hz, 1e-300)
        fc = (1.0 - VoHc) ** 5
        m = Lz > 0.0
        accA += (np.where(m, (1.0 - fc) * gv, 0.0) * jac).sum()
        accB += (np.where(m, fc * gv, 0.0) * jac).sum()
    n = nphi * nt
    return accA / n, accB / n


def main():
    own = np.fromfile(os.path.join(ROOT, "geosplatting_amd", "assets", "fg_lut_256.bin"), dtype=np.float32).reshape(256, 256, 2)
    ref_path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/rfstudio/assets/geometry/pbr/bsdf_256_256.bin"
    ref = np.fromfile(ref_path, dtype=np.float32).reshape(256, 256, 2) if os.path.exists(ref_path) else None
    rows = cols = np.array([0, 1, 2, 4, 8, 16, 24, 32, 43, 64, 96, 128, 160, 192, 224, 255])
    exact = np.zeros((len(rows), len(cols), 2))
    for i, r in enumerate(rows):
        for j, c in enumerate(cols):
            exact[i, j] = exact_texel((c + 0.5) / 256.0, (r + 0.5) / 256.0)
    # convergence of the rule itself: the same texels at half the resolution in both directions
    coarse = np.zeros_like(exact)
    for i, r in enumerate(rows):
        for j, c in enumerate(cols):
            coarse[i, j] = exact_texel((c + 0.5) / 256.0, (r + 0.5) / 256.0, 512, 2048)
    print(f"midpoint rule 1024 x 4096 (4.2 M points per texel) against 512 x 2048: max |diff| {np.abs(exact - coarse).max():.2e}")
    np.set_printoptions(linewidth=220, precision=5, suppress=True)
    for name, tab in (("packaged table (scripts/gen_fg_lut.py)", own), ("reference asset", ref)):
        if tab is None:
            print(f"{name}: not available here")
            continue
        d = tab[np.ix_(rows, cols)].astype(np.float64) - exact
        print(f"{name}: max |err| A {np.abs(d[..., 0]).max():.2e}  B {np.abs(d[..., 1]).max():.2e}   mean |err| A {np.abs(d[..., 0]).mean():.2e}  "
              f"B {np.abs(d[..., 1]).mean():.2e}")
        print("  err A, rows = roughness texel", rows.tolist(), "cols = N.V texel", cols.tolist())
        print(d[..., 0])
    if ref is not None:
        d = own[np.ix_(rows, cols)].astype(np.float64) - ref[np.ix_(rows, cols)]
        print(f"packaged - reference on the same texels: max {np.abs(d).max():.2e} mean {np.abs(d).mean():.2e}")


if __name__ == "__main__":
    main()
