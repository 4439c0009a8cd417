#!/usr/bin/env python3
"""Generate the split-sum FG (scale/bias) lookup table shipped with geosplatting_amd.

The reference loads an asset `rfstudio/assets/geometry/pbr/bsdf_256_256.bin`
(rfstudio/graphics/shaders.py:22-26).  That file is reference data and is NOT copied;
this script integrates the standard split-sum BRDF term itself:
    GGX importance sampling (Hammersley), height-correlated Smith visibility,
    Schlick Fresnel split  ->  (A, B) with  spec = F0*A + B
on a 256x256 grid, column = N.V at (x+0.5)/256, row = roughness at (y+0.5)/256, and
writes float32 [256,256,2] to geosplatting_amd/assets/fg_lut_256.bin.
tests/test_fg_lut.py checks it against a 16x16 sub-sample of the reference asset
(tests/golden/ref_fg_lut_sub16.npz) to 2e-3.
"""
import os
import sys

import numpy as np


def radical_inverse(i):
    i = i.astype(np.uint32)
    i = (i << 16) | (i >> 16)
    i = ((i & 0x55555555) << 1) | ((i & 0xAAAAAAAA) >> 1)
    i = ((i & 0x33333333) << 2) | ((i & 0xCCCCCCCC) >> 2)
    i = ((i & 0x0F0F0F0F) << 4) | ((i & 0xF0F0F0F0) >> 4)
    i = ((i & 0x00FF00FF) << 8) | ((i & 0xFF00FF00) >> 8)
    return i.astype(np.float64) * 2.3283064365386963e-10


def generate(res=256, ns=16384, chunk=2048):
    nv = (np.arange(res) + 0.5) / res
    V = np.stack([np.sqrt(1 - nv ** 2), np.zeros_like(nv), nv], -1)
    out = np.zeros((res, res, 2))
    idx = np.arange(ns)
    xi1 = (idx + 0.5) / ns
    xi2 = radical_inverse(idx)
    for yi in range(res):
        r = (yi + 0.5) / res
        a = r * r
        a2 = a * a
        acc = np.zeros((res, 2))
        for c0 in range(0, ns, chunk):
            p = 2 * np.pi * xi1[c0:c0 + chunk]
            x2 = xi2[c0:c0 + chunk]
            cos_t = np.sqrt((1 - x2) / (1 + (a2 - 1) * x2))
            sin_t = np.sqrt(np.maximum(1 - cos_t ** 2, 0))
            H = np.stack([sin_t * np.cos(p), sin_t * np.sin(p), cos_t], -1)
            VoH = V @ H.T
            Lz = 2 * VoH * H[None, :, 2] - V[:, 2:3]
            NoL = np.clip(Lz, 0, 1)
            NoH = np.clip(H[None, :, 2], 0, 1)
            VoHc = np.clip(VoH, 0, 1)
            NoV = nv[:, None]
            vis = 0.5 / np.maximum(NoL * np.sqrt(NoV * NoV * (1 - a2) + a2) + NoV * np.sqrt(NoL * NoL * (1 - a2) + a2), 1e-12)
            gv = vis * 4 * NoL * VoHc / np.maximum(NoH, 1e-12)
            fc = (1 - VoHc) ** 5
            m = Lz > 0
            acc[:, 0] += np.where(m, (1 - fc) * gv, 0).sum(1)
            acc[:, 1] += np.where(m, fc * gv, 0).sum(1)
        out[yi] = acc / ns
    return out.astype(np.float32)


if __name__ == "__main__":
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    lut = generate(ns=ns)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "geosplatting_amd", "assets", "fg_lut_256.bin")
    lut.tofile(dst)
    print("wrote", os.path.normpath(dst), lut.shape, lut.min(), lut.max())
