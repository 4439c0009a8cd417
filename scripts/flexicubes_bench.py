#!/usr/bin/env python3
"""Time the FlexiCubes extraction (forward + backward) at the reference's stage-1 resolutions on the GPU.
    python scripts/flexicubes_bench.py [R ...]            (rocprofv3 --kernel-trace --stats -- python scripts/flexicubes_bench.py 96)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from geosplatting_amd.flexicubes import FlexiCubes          # noqa: E402

for R in [int(a) for a in sys.argv[1:]] or [72, 96, 128]:
    torch.manual_seed(0)
    fc = FlexiCubes.from_resolution(R, device="cuda", random_sdf=False)
    p = fc.vertices
    leaf = lambda t: t.detach().contiguous().requires_grad_(True)
    sdf = leaf((p * torch.tensor([1.0, 1.3, 0.8], device="cuda")).norm(dim=-1, keepdim=True) - 0.613
               + 0.05 * torch.sin(7 * p[:, 0:1]) * torch.cos(5 * p[:, 1:2]))
    w = torch.zeros(R ** 3, 21, device="cuda").normal_(0, 0.3)
    f = fc.replace(vertices=leaf(p + torch.zeros_like(p).normal_(0, 0.5).tanh() * (0.5 / R)), sdf_values=sdf,
                   alpha=leaf(w[:, :8]), beta=leaf(w[:, 8:20]), gamma=leaf(w[:, 20:]))
    for _ in range(3):
        (v, fa), L = f.dual_marching_cubes(); (v.sum() + L.sum()).backward()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 20
    e[0].record()
    for _ in range(n):
        (v, fa), L = f.dual_marching_cubes()
    e[1].record()
    for _ in range(n):
        (v, fa), L = f.dual_marching_cubes(); (v.sum() + L.sum()).backward()
    e[2].record(); torch.cuda.synchronize()
    fwd = e[0].elapsed_time(e[1]) / n; both = e[1].elapsed_time(e[2]) / n
    print(f"R={R}: V={v.shape[0]} F={fa.shape[0]} K={L.shape[0]}  fwd {fwd:.3f} ms  fwd+bwd {both:.3f} ms")
