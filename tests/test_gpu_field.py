"""Hash-grid encoder HIP kernels (csrc/gs_hashgrid.hip) vs the reference golden vectors and the float64 restatement."""
import os

import numpy as np
import pytest
import torch

from oracle import field_ref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gold():
    g = np.load(os.path.join(GOLD, "ref_hashgrid.npz"))
    torch.manual_seed(int(g["b_table_seed"]))
    return g, torch.rand(16 * 2 ** int(g["b_log2"]), 2) * 2 - 1


@pytest.mark.parametrize("tag", ["a", "b"])
def test_hashgrid_golden(tag):
    """features, position gradient and the scattered table gradient of the reference's torch backend"""
    from geosplatting_amd.field import hash_encode
    g, b_table = _gold()
    table = (torch.tensor(g["a_table"]) if tag == "a" else b_table).cuda().requires_grad_(True)
    x = torch.tensor(g[f"{tag}_x"]).cuda().requires_grad_(True)
    f = hash_encode(x, table, torch.tensor(g[f"{tag}_scalings"]), int(g[f"{tag}_log2"]))
    assert np.allclose(f.detach().cpu().numpy(), g[f"{tag}_feats"], atol=1e-6)
    (f * torch.tensor(g[f"{tag}_g"]).cuda()).sum().backward()
    vx = g[f"{tag}_v_x"]
    assert np.abs(x.grad.cpu().numpy() - vx).max() < 1e-4 * np.abs(vx).max()
    rows = torch.tensor(g[f"{tag}_touched"])
    tg = table.grad.cpu()
    assert np.abs(tg[rows].numpy() - g[f"{tag}_v_table_touched"]).max() < 1e-4 * np.abs(g[f"{tag}_v_table_touched"]).max()
    mask = torch.ones(tg.shape[0], dtype=torch.bool); mask[rows] = False
    assert tg[mask].abs().max().item() == 0.0                          # bit-exact cell indices: nothing else is touched


def test_hashgrid_full_call_golden():
    """HashEncoding.__call__ of GaussianField.kd_enc: grad-scaling trick + 3-layer MLP with sigmoid"""
    from geosplatting_amd.field import hash_encode
    g, _ = _gold()
    table = torch.tensor(g["a_table"]).cuda().requires_grad_(True)
    x = torch.tensor(g["a_x"][:256]).cuda().requires_grad_(True)
    f = hash_encode(x, table, torch.tensor(g["a_scalings"]), int(g["a_log2"]), grad_scaling=16.0)
    y = torch.relu(torch.relu(f @ torch.tensor(g["c_w0"]).cuda().t()) @ torch.tensor(g["c_w1"]).cuda().t())
    y = (y @ torch.tensor(g["c_w2"]).cuda().t()).sigmoid()
    assert np.allclose(y.detach().cpu().numpy(), g["c_y"], atol=2e-6)
    (y * torch.tensor(g["c_gy"]).cuda()).sum().backward()
    assert np.abs(x.grad.cpu().numpy() - g["c_v_x"]).max() < 1e-4 * np.abs(g["c_v_x"]).max()
    got = table.grad.cpu()[torch.tensor(g["c_touched"])].numpy()
    assert np.abs(got - g["c_v_table_touched"]).max() < 1e-4 * np.abs(g["c_v_table_touched"]).max()


def test_hashgrid_vs_float64_and_properties():
    """GaussianField configuration (16 levels, 16..4096, 2^18 entries) on 200k points: fp32 restatement (the reference
    computes cell coordinates in fp32: at resolution 4096 one ulp of `scaled` is 5e-4 of a cell, so a float64 evaluation
    is a DIFFERENT function at the 1e-3 level -- parity is against the reference's arithmetic), linearity in the table,
    determinism of the forward, errors"""
    from geosplatting_amd import _lib
    from geosplatting_amd.field import HashEncoding, hash_encode, level_scalings
    gen = torch.Generator().manual_seed(3)
    L, log2 = 16, 18
    sc = level_scalings(L, 16, 4096)
    table = (torch.rand(L * 2 ** log2, 2, generator=gen) * 2 - 1)
    x = (torch.rand(200_000, 3, generator=gen) * 2 - 1)
    gy = torch.randn(200_000, 32, generator=gen)
    tc = table.cuda().requires_grad_(True); xc = x.cuda().requires_grad_(True)
    f = hash_encode(xc, tc, sc, log2)
    (f * gy.cuda()).sum().backward()
    sub = slice(0, 20_000)
    td = table.clone().requires_grad_(True); xd = x[sub].clone().requires_grad_(True)
    fr = field_ref.encode(xd, td, sc, log2)
    assert (f[sub].detach().cpu() - fr.detach()).abs().max().item() < 2e-6
    # gradient of the subset alone, through a second GPU call
    tc2 = table.cuda().requires_grad_(True); xc2 = x[sub].cuda().requires_grad_(True)
    (hash_encode(xc2, tc2, sc, log2) * gy[sub].cuda()).sum().backward()
    (fr * gy[sub]).sum().backward()
    assert (xc2.grad.cpu() - xd.grad).abs().max().item() < 1e-4 * xd.grad.abs().max().item()
    assert (tc2.grad.cpu() - td.grad).abs().max().item() < 1e-4 * td.grad.abs().max().item()
    assert torch.equal(xc.grad[sub], xc2.grad)                           # per-point work is independent of the batch
    f2 = hash_encode(xc.detach(), (2.5 * tc).detach(), sc, log2)
    assert torch.allclose(f2, 2.5 * f.detach(), rtol=1e-5, atol=1e-6)    # linear in the table
    assert torch.equal(hash_encode(xc.detach(), tc.detach(), sc, log2), f.detach())
    enc = HashEncoding([-1, 32, 32, 3], activation="sigmoid", max_res=4096, log2_hashmap_size=18, grad_scaling=16.0)
    y = enc(xc.detach()[:1000])
    assert y.shape == (1000, 3) and (y > 0).all() and (y < 1).all()
    y.sum().backward()
    assert enc.hash_table.grad.abs().max().item() > 0
    with pytest.raises(_lib.GeoSplatHipError):
        hash_encode(x, table, sc, log2)                                  # CPU tensors: no CPU path
    with pytest.raises(_lib.GeoSplatHipError):
        hash_encode(xc, tc[:100], sc, log2)                              # wrong table size
