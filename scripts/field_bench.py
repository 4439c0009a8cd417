"""Timing of the hash-grid encoder at the GaussianField configuration (16 levels, 16..4096, 2^18 entries, F = 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geosplatting_amd.field import hash_encode, level_scalings
import geosplatting_amd.synthetic as syn
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1966080
sc = level_scalings(16, 16, 4096)
table = ((torch.rand(16 * 2 ** 18, 2) * 2 - 1) * 1e-3).to(dev).requires_grad_(True)
v, f = syn.icosphere(7, radius=0.8)
from geosplatting_amd.mesh import mesh_to_splats, vertex_normals
v, f = v.to(dev), f.to(dev)
sp, _ = mesh_to_splats(v, f, vertex_normals(v, f))
x = sp.means[:N].clamp(-1, 1).to(dev).requires_grad_(True)       # surface points in mesh order (spatially coherent), as in training
gy = torch.randn(x.shape[0], 32, device=dev)
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(); y = hash_encode(x, table, sc, 18, grad_scaling=16.0); e1.record(); y.backward(gy); e2.record(); torch.cuda.synchronize()
    print(f"N={x.shape[0]} hash-grid fwd {e0.elapsed_time(e1):.3f} ms  bwd {e1.elapsed_time(e2):.3f} ms")
