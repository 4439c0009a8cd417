"""Timing of the field's 32-wide MLP (torch linear = rocBLAS) forward + backward at N points: how much of a stage-1 iteration
at 2 M Gaussians goes into the skinny weight-gradient GEMMs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geosplatting_amd.field import HashEncoding
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1966080
enc = HashEncoding([-1, 32, 32, 3], activation="sigmoid", max_res=4096, log2_hashmap_size=18, grad_scaling=16.0, device=dev)
x = (torch.rand(N, 3, device=dev) * 1.6 - 0.8)
feats = torch.randn(N, 32, device=dev, requires_grad=True)
ws = [w.detach().clone().requires_grad_(True) for w in enc.weights]
def mlp(f):
    h = f
    for i, w in enumerate(ws):
        h = torch.nn.functional.linear(h, w)
        if i < len(ws) - 1: h = torch.relu(h)
    return torch.sigmoid(h)
g = torch.randn(N, 3, device=dev)
for rep in range(3):
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    e[0].record(); y = mlp(feats); e[1].record(); y.backward(g); e[2].record()
    out = enc(x); e[3].record(); out.backward(g); e[4].record(); torch.cuda.synchronize()
    print(f"N={N}: MLP alone fwd {e[0].elapsed_time(e[1]):.2f} ms bwd {e[1].elapsed_time(e[2]):.2f} ms | full HashEncoding fwd {e[2].elapsed_time(e[3]):.2f} ms bwd {e[3].elapsed_time(e[4]):.2f} ms")
