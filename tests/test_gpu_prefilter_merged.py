"""Round 6: the prefilter's launches collapsed -- the mip chain in one launch, all specular levels of a direction in one launch
(gs_cubemap_mip_chain_fwd, gs_specular_tiles_apply_multi), `as_splitsum` as one autograd node -- against the per-level launches
they replace (which carry the oracle checks of tests/test_gpu_shading.py): bit for bit."""
import ctypes as C

import pytest
import torch

import geosplatting_amd as gs
import geosplatting_amd.synthetic as syn
from geosplatting_amd import _lib as L
from geosplatting_amd import splitsum as SS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,n", [(64, 2), (512, 5), (128, 3), (32, 1)])
def test_mip_chain_equals_single_calls(cuda, R, n):
    cube = syn.make_cubemap(R, seed=5).to(cuda).contiguous()
    outs = [torch.empty(6, R >> (k + 1), R >> (k + 1), 3, device=cuda) for k in range(n)]
    ptrs = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    L.check(L.lib().gs_cubemap_mip_chain_fwd(R, n, L.ptr(cube), ptrs, L.stream()), "gs_cubemap_mip_chain_fwd")
    ref = cube
    for k in range(n):
        nxt = torch.empty(6, R >> (k + 1), R >> (k + 1), 3, device=cuda)
        L.check(L.lib().gs_cubemap_mip_fwd(R >> k, 3, L.ptr(ref), L.ptr(nxt), L.stream()), "gs_cubemap_mip_fwd")
        assert torch.equal(outs[k], nxt), f"level {k + 1}"
        ref = nxt
    # argument validation
    assert L.lib().gs_cubemap_mip_chain_fwd(R, 6, L.ptr(cube), ptrs, L.stream()) == -1
    assert L.lib().gs_cubemap_mip_chain_fwd(48, 5, L.ptr(cube), ptrs, L.stream()) == -1


@pytest.mark.parametrize("R", [64, 256])
def test_merged_levels_equal_per_level_launches(cuda, R, monkeypatch):
    """forward pyramid and cubemap gradient: one launch per direction == one launch per level, bit for bit"""
    cube = syn.make_cubemap(R, seed=3).to(cuda)
    g = torch.Generator().manual_seed(4)

    def run(merged):
        monkeypatch.setattr(SS, "MERGED_APPLY", merged)
        x = cube.clone().requires_grad_(True)
        env = gs.as_splitsum(x)
        gen = torch.Generator().manual_seed(4)
        loss = (env.base * (torch.rand(env.base.shape, generator=gen) - 0.5).to(cuda)).sum()
        for l in env.levels:
            loss = loss + (l * (torch.rand(l.shape, generator=gen) - 0.5).to(cuda)).sum()
        loss.backward()
        return env, x.grad

    env_m, grad_m = run(True)
    env_p, grad_p = run(False)
    assert torch.equal(env_m.base, env_p.base)
    assert len(env_m.levels) == len(env_p.levels)
    for a, b in zip(env_m.levels, env_p.levels):
        assert torch.equal(a, b)
    assert torch.equal(grad_m, grad_p)


def test_explicit_backward_into_caller_buffer(cuda):
    """as_splitsum_backward(out=): the finest level's transposed apply writes the caller's slice (the engine's gradient bucket)"""
    cube = syn.make_cubemap(64, seed=7).to(cuda)
    with torch.no_grad():
        env = gs.as_splitsum(cube)
    gen = torch.Generator().manual_seed(1)
    gb = (torch.rand(env.base.shape, generator=gen) - 0.5).to(cuda)
    gl = [(torch.rand(l.shape, generator=gen) - 0.5).to(cuda) for l in env.levels]
    a = SS.as_splitsum_backward(gb.clone(), [t.clone() for t in gl])
    flat = torch.full((6 * 64 * 64 * 3 + 64,), 7.0, device=cuda)
    out = flat[64:].view(6, 64, 64, 3)
    b = SS.as_splitsum_backward(gb.clone(), [t.clone() for t in gl], out=out)
    assert b.data_ptr() == out.data_ptr()
    assert torch.equal(a, out)
    assert bool((flat[:64] == 7.0).all())


def test_partial_cotangents_and_sharded_ranges(cuda):
    """a loss that reads only some levels (None cotangents), and merged launches over tile sub-ranges (the sharded prefilter)"""
    cube = syn.make_cubemap(64, seed=11).to(cuda)
    x = cube.clone().requires_grad_(True)
    env = gs.as_splitsum(x)
    env.levels[1].sum().backward()
    assert x.grad is not None and float(x.grad.abs().sum()) > 0
    # two shares of every level summed == the whole level
    with torch.no_grad():
        whole = gs.as_splitsum(cube)
    mips = SS.mip_chain(cube)
    roughs = SS._level_roughness(len(mips), 0.08, 0.5)
    acc = [torch.zeros_like(m) for m in mips]
    for rank in range(2):
        part = [torch.zeros_like(m) for m in mips]
        jobs = []
        for m, o, r in zip(mips, part, roughs):
            e = SS.specular_tiles(int(m.shape[1]), r, 0.99, cuda)
            t0, t1 = SS.shard_tiles(e["n_tiles"], rank, 2)
            jobs.append((e, m, o, t0, t1))
        SS._tiles_apply_multi(jobs, "fwd", 2)
        for a, q in zip(acc, part):
            a += q
    for a, b in zip(acc, whole.levels):
        assert torch.equal(a, b)


def test_activation_chain_equals_torch(cuda):
    N = 100003
    gen = torch.Generator().manual_seed(2)
    gs_act = (torch.randn(N, 3, generator=gen)).to(cuda); s_act = torch.randn(N, 3, generator=gen).exp().to(cuda)
    go = torch.randn(N, generator=gen).to(cuda); o = torch.sigmoid(torch.randn(N, generator=gen)).to(cuda)
    vs = torch.empty(N, 3, device=cuda); vo = torch.empty(N, 1, device=cuda)
    L.check(L.lib().gs_activation_chain(L.i64(N), L.ptr(gs_act), L.ptr(s_act), L.ptr(go), L.ptr(o), L.ptr(vs), L.ptr(vo), L.stream()),
            "gs_activation_chain")
    assert torch.equal(vs, gs_act * s_act)
    assert torch.equal(vo, (go * o * (1.0 - o)).unsqueeze(-1))
