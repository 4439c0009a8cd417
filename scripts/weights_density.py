"""Diagnostic: fraction of non-zero entries in the cached prefilter weight tables, per pyramid level."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd.splitsum as ss
dev = torch.device("cuda:0")
for i, res in enumerate([512, 256, 128, 64, 32, 16]):
    rough = (i / 5) * (0.5 - 0.08) + 0.08
    e = ss.specular_weights(res, rough, 0.99, dev)
    w = e["fwd"]
    nz = 0
    for c in range(0, w.numel(), 1 << 28):
        nz += int((w[c:c + (1 << 28)] != 0).sum().item())
    b = e["bounds"].view(-1, 6, 4)
    wid = (b[..., 1] - b[..., 0] + 1).clamp(min=0); hei = (b[..., 3] - b[..., 2] + 1).clamp(min=0)
    box = float((wid * hei).sum().item())
    print(f"R={res} rough={rough:.3f} patches/texel={e['total'] / (6 * res * res):.1f} nonzero={nz / w.numel():.3f} "
          f"nonzero/texel={nz / (6 * res * res):.1f} box/texel={box / (6 * res * res):.1f} dense/texel={e['total'] * 64 / (6 * res * res):.1f}")
    ss._weights_cache.clear()
