"""Timing experiment: hash-grid backward with pieces of the slab kernel removed (-DGS_HG_EXP=...; diagnostic builds in /tmp)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import geosplatting_amd.build as B
tag = sys.argv[1] if len(sys.argv) > 1 else "base"
defs = sys.argv[2:]
so = f"/tmp/libgeosplat_hg_{tag}.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, *defs, "-shared", "-o", so, *[os.path.join(B.CSRC, s) for s in B.SOURCES]])
import geosplatting_amd._lib as L
L.LIB_PATH = so
import torch
from geosplatting_amd.field import hash_encode, level_scalings
import geosplatting_amd.synthetic as syn
dev = torch.device("cuda:0")
sc = level_scalings(16, 16, 4096)
table = ((torch.rand(16 * 2 ** 18, 2) * 2 - 1) * 1e-3).to(dev).requires_grad_(True)
v, f = syn.icosphere(7, radius=0.8)
from geosplatting_amd.mesh import mesh_to_splats, vertex_normals
v, f = v.to(dev), f.to(dev)
sp, _ = mesh_to_splats(v, f, vertex_normals(v, f))
x = sp.means.clamp(-1, 1).to(dev)
gy = torch.randn(x.shape[0], 32, device=dev)
for rep in range(3):
    y = hash_encode(x, table, sc, 18, grad_scaling=16.0)
    torch.cuda.synchronize()
    e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e1.record(); y.backward(gy); e2.record(); torch.cuda.synchronize()
print(f"{tag}: N={x.shape[0]} table-gradient backward (transpose + slab kernel) {e1.elapsed_time(e2):.3f} ms")
