"""CPU restatement of the hash-grid field encoder -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; geosplatting_amd never does.

Restates `HashEncoding` with `backend='torch'` (rfstudio/model/components/encoding.py:124-241: __setup__ scalings,
hash_fn, pytorch_fwd, the grad-scaling trick of __call__) and `MLP.__call__` (rfstudio/nn/mlp.py:126-145) in plain
torch, any dtype (float64 for gradient checks).  PINNED by tests/golden/ref_hashgrid.npz, which
scripts/make_golden_field.py produced by executing those reference functions themselves.
(The reference's default `backend='tcnn'` is tinycudann -- CUDA only, absent, different cell convention; it is NOT what
is restated here.)
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch


def level_scalings(num_levels: int = 16, min_res: int = 16, max_res: int = 1024) -> torch.Tensor:
    """encoding.py:124-132 (same torch expression, so the same float32 values)"""
    levels = torch.arange(num_levels)
    growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1
    return torch.floor(min_res * growth ** levels)


def hash_fn(coords: torch.Tensor, table_size: int, num_levels: int) -> torch.Tensor:
    """encoding.py:168-185; coords [..., L, 3] int32 -> rows of the [L*T, F] table"""
    c = coords * torch.tensor([1, 2654435761, 805459861])
    x = torch.bitwise_xor(torch.bitwise_xor(c[..., 0], c[..., 1]), c[..., 2])
    x = x % table_size
    return x + torch.arange(num_levels) * table_size


def encode(x: torch.Tensor, table: torch.Tensor, scalings: torch.Tensor, log2_T: int) -> torch.Tensor:
    """encoding.py:187-229; x [N,3] in [-1,1] -> [N, L*F]"""
    L = scalings.shape[0]
    T = 2 ** log2_T
    p = x[..., None, :] * 0.5 + 0.5
    scaled = p * scalings.view(-1, 1).to(p.dtype)
    sc = torch.ceil(scaled).to(torch.int32); sf = torch.floor(scaled).to(torch.int32)
    o = scaled - sf
    pick = lambda bx, by, bz: table[hash_fn(torch.stack([(sc if bx else sf)[..., 0], (sc if by else sf)[..., 1],
                                                           (sc if bz else sf)[..., 2]], -1), T, L)]
    ox, oy, oz = o[..., 0:1], o[..., 1:2], o[..., 2:3]
    f03 = pick(1, 1, 1) * ox + pick(0, 1, 1) * (1 - ox)
    f12 = pick(1, 0, 1) * ox + pick(0, 0, 1) * (1 - ox)
    f56 = pick(1, 0, 0) * ox + pick(0, 0, 0) * (1 - ox)
    f47 = pick(1, 1, 0) * ox + pick(0, 1, 0) * (1 - ox)
    f0312 = f03 * oy + f12 * (1 - oy)
    f4756 = f47 * oy + f56 * (1 - oy)
    return torch.flatten(f0312 * oz + f4756 * (1 - oz), start_dim=-2)


def mlp(feats: torch.Tensor, weights: Sequence[torch.Tensor], activation: str = "none") -> torch.Tensor:
    """mlp.py:126-145 without bias / skip connections: ReLU between layers, `activation` after the last"""
    x = feats
    for i, w in enumerate(weights):
        x = x @ w.t()
        if i < len(weights) - 1:
            x = torch.relu(x)
        elif activation == "sigmoid":
            x = x.sigmoid()
    return x


def hash_encoding(x, table, scalings, log2_T, weights: Optional[List[torch.Tensor]] = None, activation="none",
                  grad_scaling: Optional[float] = None):
    """encoding.py:231-241: the full HashEncoding.__call__"""
    if grad_scaling is not None:
        x = x * (1 / grad_scaling) + x.detach() * (1 - 1 / grad_scaling)
    f = encode(x, table, scalings, log2_T)
    if grad_scaling is not None:
        f = f * grad_scaling + f.detach() * (1 - grad_scaling)
    return f if weights is None else mlp(f, weights, activation)
