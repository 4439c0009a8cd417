"""Diagnostic: how many weights would a finer patch granularity (8x4 halves, 8x2 quarters, 8x1 rows) drop from the cached
prefilter tables (after the all-zero 8x8 patches are gone)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd.splitsum as ss
dev = torch.device("cuda:0")
for i, res in enumerate([512, 256, 128]):
    rough = (i / 5) * (0.5 - 0.08) + 0.08
    e = ss.specular_weights(res, rough, 0.99, dev)
    W = e["fwd"].view(e["total"], 8, 8)
    nz = W != 0
    out = [f"R={res} patches={e['total']} dense={nz.float().mean().item():.3f}"]
    for rows in (4, 2, 1):
        k = nz.view(e["total"], 8 // rows, rows * 8).any(-1).float().mean().item()
        out.append(f"8x{rows} kept {k:.3f}")
    print("  ".join(out))
    ss._weights_cache.clear()
