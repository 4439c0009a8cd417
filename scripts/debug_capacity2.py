import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_parallel import _engine_steps
dev = torch.device("cuda", 0)
runs = {}
for name, n, cap in (("e1", 1, False), ("e2", 2, False), ("e3", 3, False), ("e3b", 3, False), ("c3", 3, True), ("c3b", 3, True)):
    runs[name] = _engine_steps(dev, n, use_capacity=cap)
def d(a, b):
    return max((x - y).abs().max().item() for x, y in zip(runs[a][1], runs[b][1]))
for a, b in (("e1", "e2"), ("e2", "e3"), ("e3", "e3b"), ("e3", "c3"), ("c3", "c3b"), ("e2", "c3")):
    print(a, b, d(a, b))
