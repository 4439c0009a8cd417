"""GPU parity of the engine's fused per-view kernels (csrc/gs_front.hip, gs_isect_bin_front) against the call-shaped ops they
replace, which are themselves checked against the CPU oracle (tests/test_gpu_shading.py, tests/test_gpu_rasterizer.py):
  gs_front_fwd        == gs_shade_fwd + gs_project_fwd_vis (+ tile rectangles)          bit for bit
  gs_isect_bin_front  == rasterization()'s flatten_ids / isect_offsets                  bit for bit (24- and 32-bit keys, > 8 192 tiles)
  gs_tail_bwd         == gs_project_bwd + gs_shade_bwd                                  1e-5 (summation order of the two mean paths)
and the oracle itself on the whole chain through engine.RenderStep is tests/test_gpu_fullsize.py / test_gpu_parallel.py."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from tests.util import random_case, rel_err, sphere_case

pytestmark = pytest.mark.gpu


def _env(cuda, seed=3, res=(64, 32, 16)):
    import geosplatting_amd as gs
    g = torch.Generator().manual_seed(seed)
    levels = [(torch.rand(6, r, r, 3, generator=g) + 0.05).to(cuda) for r in res]
    base = (torch.rand(6, 16, 16, 3, generator=g) + 0.05).to(cuda)
    return gs.TextureSplitSum(base, levels)


def _inputs(cuda, level=3, res=128, view=1):
    sc, cam = sphere_case(level, res, view=view)
    d = lambda t: t.to(cuda).contiguous()
    sp = sc.splats
    return dict(means=d(sp.means), quats=d(sp.quats), scales=d(sp.scales.exp()), opac=d(torch.sigmoid(sp.opacities).squeeze(-1)),
                normals=d(sc.normals), kd=d(sc.kd), ks=d(sc.ks), vm=d(cam.view_matrix), K=d(cam.intrinsic_matrix),
                cam_pos=d(cam.c2w[:, 3]), W=res, H=res)


def _front(x, env, cuda, key_base=0, key_bits=32, status=None, mode="pbr", tight=False):
    import geosplatting_amd as gs
    from geosplatting_amd import front as F
    from geosplatting_amd.shading import _MODE, _make_env
    e = _make_env(gs.get_fg_lut(cuda), env)
    fr = F.front_stage(x["means"], x["quats"], x["scales"], x["opac"], x["normals"], x["kd"], x["ks"], x["vm"], x["K"], x["cam_pos"], e,
                       x["W"], x["H"], 0.1, 1.0, _MODE[mode], key_base, key_bits, status, tight_tiles=tight)
    torch.cuda.synchronize()
    return fr, e


def _reference_meta(x, env, cuda, mode="pbr"):
    import geosplatting_amd as gs
    col = gs.shade(x["means"], x["normals"], x["kd"], x["ks"], x["cam_pos"], env, min_roughness=0.1, max_metallic=1.0, mode=mode)
    r, a, meta = gs.rasterization(x["means"], x["quats"], x["scales"], x["opac"], col, x["vm"][None], x["K"][None], x["W"], x["H"])
    return col, meta


@pytest.mark.parametrize("mode", ["pbr", "specular", "diffuse"])
def test_front_records_equal_shade_plus_project(cuda, mode):
    x = _inputs(cuda)
    env = _env(cuda)
    fr, _ = _front(x, env, cuda, mode=mode)
    col, meta = _reference_meta(x, env, cuda, mode)
    V, I = int(fr.host_counts[0]), int(fr.host_counts[1])
    assert V == meta["radii"].shape[0] and I == meta["flatten_ids"].shape[0] and V > 1000
    vis = fr.vis[:V].cpu().numpy()
    gid = np.ascontiguousarray(vis[:, 13]).view(np.int32)
    assert np.array_equal(gid, meta["gaussian_ids"].cpu().numpy().astype(np.int32))
    m2, con = meta["means2d"].cpu().numpy(), meta["conics"].cpu().numpy()
    assert np.array_equal(vis[:, 0:2], m2)
    assert np.array_equal(vis[:, 2], 0.5 * con[:, 0]) and np.array_equal(vis[:, 3], con[:, 1]) and np.array_equal(vis[:, 4], 0.5 * con[:, 2])
    assert np.array_equal(vis[:, 5], meta["opacities"].cpu().numpy())
    assert np.array_equal(vis[:, 8:11], col.detach().cpu().numpy()[gid])            # the colours of gs_shade_fwd, bit for bit
    assert np.array_equal(vis[:, 12], meta["compensations"].cpu().numpy())
    assert np.array_equal(vis[:, 14], meta["depths"].cpu().numpy())
    assert np.array_equal(np.ascontiguousarray(vis[:, 15]).view(np.int32), meta["radii"].cpu().numpy())
    keys = fr.keys[:V].cpu().numpy().view(np.uint32)
    assert np.array_equal(keys, meta["depths"].cpu().numpy().view(np.uint32))       # 32-bit keys = the depth bits
    rc = fr.rects[:V].cpu().numpy().view(np.uint32)
    x0, y0, x1, y1 = rc[:, 0] & 0xffff, rc[:, 0] >> 16, rc[:, 1] & 0xffff, rc[:, 1] >> 16
    assert np.array_equal(((x1 - x0) * (y1 - y0)).astype(np.int32), meta["tiles_per_gauss"].cpu().numpy())
    lo, hi = 0xffffffff - int(fr.host_counts[2]), int(fr.host_counts[3])
    bits = meta["depths"].cpu().numpy().view(np.uint32)
    assert lo == int(bits.min()) and hi == int(bits.max())


@pytest.mark.parametrize("key_bits,res", [(32, 128), (24, 128), (24, 1616)])
def test_bin_front_equals_rasterization_order(cuda, key_bits, res):
    """flatten_ids / offsets of the fused binning against rasterization()'s meta (itself bit-exact against the oracle);
    res 1616 -> 101 x 101 = 10 201 tiles: the path that keeps the tile ids (more tiles than the LDS histogram holds)."""
    from geosplatting_amd import front as F
    x = _inputs(cuda, level=3, res=res)
    env = _env(cuda)
    _, meta = _reference_meta(x, env, cuda)
    bits = meta["depths"].cpu().numpy().view(np.uint32)
    status = torch.zeros(4, dtype=torch.int64, device=cuda)
    base = int(bits.min()) - (1 << 20) if key_bits == 24 else 0
    fr, _ = _front(x, env, cuda, key_base=base, key_bits=key_bits, status=status)
    V, I = int(fr.host_counts[0]), int(fr.host_counts[1])
    # exact mode
    state, v, i = F.bin_stage(fr, None, None)
    assert (v, i) == (V, I)
    torch.cuda.synchronize()
    assert np.array_equal(state["flatten_ids"][:I].cpu().numpy(), meta["flatten_ids"].cpu().numpy())
    assert np.array_equal(state["isect_offsets"].cpu().numpy(), meta["isect_offsets"].reshape(-1).cpu().numpy())
    # capacity mode: counts read on the device, buffers sized by (N, I_cap)
    fr2, _ = _front(x, env, cuda, key_base=base, key_bits=key_bits, status=status)
    cap = ((int(I * 1.25) + 65535) // 65536) * 65536
    state2, v2, i2 = F.bin_stage(fr2, cap, status)
    torch.cuda.synchronize()
    assert (v2, i2) == (x["means"].shape[0], cap)
    assert np.array_equal(state2["flatten_ids"][:I].cpu().numpy(), meta["flatten_ids"].cpu().numpy())
    assert np.array_equal(state2["isect_offsets"].cpu().numpy(), meta["isect_offsets"].reshape(-1).cpu().numpy())
    assert status.cpu().tolist() == [0, 0, 0, 0]


def test_bin_front_depth_ties_and_random_scene(cuda):
    """random splats (many tiles per Gaussian, overlapping depths) with a block of exactly equal depths: ties keep packed order"""
    import geosplatting_amd as gs
    from geosplatting_amd import front as F
    sp, cam = random_case(20000, 256)
    d = lambda t: t.to(cuda).contiguous()
    means = sp.means.clone()
    vm = cam.view_matrix
    # 4 000 Gaussians on one plane of constant camera depth: z_cam = const  ->  identical depth bits are likely; force them
    means[:4000] = means[:4000] - (means[:4000] @ vm[2, :3])[:, None] * vm[2, :3][None, :]
    x = dict(means=d(means), quats=d(sp.quats), scales=d(sp.scales.exp()), opac=d(torch.sigmoid(sp.opacities).squeeze(-1)),
             normals=d(torch.nn.functional.normalize(torch.randn(sp.num, 3, generator=torch.Generator().manual_seed(5)), dim=-1)),
             kd=d(torch.rand(sp.num, 3)), ks=d(torch.rand(sp.num, 2)), vm=d(vm), K=d(cam.intrinsic_matrix), cam_pos=d(cam.c2w[:, 3]),
             W=256, H=256)
    env = _env(cuda)
    _, meta = _reference_meta(x, env, cuda)
    fr, _ = _front(x, env, cuda)
    I = int(fr.host_counts[1])
    state, _, _ = F.bin_stage(fr, None, None)
    torch.cuda.synchronize()
    assert I == meta["flatten_ids"].shape[0] and I > 50000
    assert np.array_equal(state["flatten_ids"][:I].cpu().numpy(), meta["flatten_ids"].cpu().numpy())
    assert np.array_equal(state["isect_offsets"].cpu().numpy(), meta["isect_offsets"].reshape(-1).cpu().numpy())


def test_tight_tiles_drop_only_tiles_no_pixel_composites(cuda):
    """tight_tiles: gsplat's tile square clipped to the {alpha >= 1/255} extents.  Same records, same packed order; the tile list of
    every Gaussian is a sub-rectangle of gsplat's; the sorted list is gsplat's list with the dropped (tile, Gaussian) pairs removed,
    order untouched; and NO pixel centre of a dropped tile reaches alpha >= 1/255 (float64, every dropped pair of the scene)."""
    from geosplatting_amd import front as F
    sp, cam = random_case(6000, 256, seed=4)                    # random anisotropic splats: thin diagonal ellipses included
    d = lambda t: t.to(cuda).contiguous()
    g = torch.Generator().manual_seed(8)
    scales = sp.scales.exp() * torch.tensor([1.0, 0.15, 0.02])   # flattened, elongated
    x = dict(means=d(sp.means), quats=d(sp.quats), scales=d(scales), opac=d(torch.rand(sp.num, generator=g) * 0.98 + 0.01),
             normals=d(torch.nn.functional.normalize(torch.randn(sp.num, 3, generator=g), dim=-1)), kd=d(torch.rand(sp.num, 3, generator=g)),
             ks=d(torch.rand(sp.num, 2, generator=g)), vm=d(cam.view_matrix), K=d(cam.intrinsic_matrix), cam_pos=d(cam.c2w[:, 3]), W=256, H=256)
    env = _env(cuda)
    fg, _ = _front(x, env, cuda)
    ft, _ = _front(x, env, cuda, tight=True)
    V = int(fg.host_counts[0])
    assert int(ft.host_counts[0]) == V and int(ft.host_counts[1]) < int(fg.host_counts[1])
    assert torch.equal(fg.vis[:V], ft.vis[:V]) and torch.equal(fg.keys[:V], ft.keys[:V])
    rg = fg.rects[:V].cpu().numpy().view(np.uint32).astype(np.int64); rt = ft.rects[:V].cpu().numpy().view(np.uint32).astype(np.int64)
    box = lambda r: (r[:, 0] & 0xffff, r[:, 0] >> 16, r[:, 1] & 0xffff, r[:, 1] >> 16)
    gx0, gy0, gx1, gy1 = box(rg); tx0, ty0, tx1, ty1 = box(rt)
    nonempty = (tx1 > tx0) & (ty1 > ty0)
    assert (tx0[nonempty] >= gx0[nonempty]).all() and (tx1 <= gx1).all() and (ty0[nonempty] >= gy0[nonempty]).all() and (ty1 <= gy1).all()
    # dropped tiles: alpha at every pixel centre, float64
    vis = fg.vis[:V].cpu().numpy().astype(np.float64)
    mx, my, ha, cb, hc, op = vis[:, 0], vis[:, 1], vis[:, 2], vis[:, 3], vis[:, 4], vis[:, 5]
    worst, n_dropped = 0.0, 0
    for v in range(V):
        for ty in range(gy0[v], gy1[v]):
            for tx in range(gx0[v], gx1[v]):
                if tx0[v] <= tx < tx1[v] and ty0[v] <= ty < ty1[v] and nonempty[v]:
                    continue
                n_dropped += 1
                px = tx * 16 + np.arange(16) + 0.5; py = ty * 16 + np.arange(16) + 0.5
                dx = mx[v] - px[None, :]; dy = my[v] - py[:, None]
                sigma = ha[v] * dx * dx + hc[v] * dy * dy + cb[v] * dx * dy
                alpha = np.where(sigma >= 0, np.minimum(0.999, op[v] * np.exp(-sigma)), 0.0)
                worst = max(worst, float(alpha.max()))
    assert n_dropped > 1000 and worst < 1.0 / 255.0, (n_dropped, worst)
    # the sorted lists: tight == gsplat's with the dropped pairs filtered out
    sg, _, _ = F.bin_stage(fg, None, None)
    st_, _, _ = F.bin_stage(ft, None, None)
    torch.cuda.synchronize()
    Ig, It = int(fg.host_counts[1]), int(ft.host_counts[1])
    flat_g = sg["flatten_ids"][:Ig].cpu().numpy(); off_g = sg["isect_offsets"].cpu().numpy()
    tile_of = np.repeat(np.arange(off_g.size), np.diff(np.append(off_g, Ig)))
    ty_, tx_ = tile_of // 16, tile_of % 16
    keep = nonempty[flat_g] & (tx_ >= tx0[flat_g]) & (tx_ < tx1[flat_g]) & (ty_ >= ty0[flat_g]) & (ty_ < ty1[flat_g])
    assert int(keep.sum()) == It
    assert np.array_equal(st_["flatten_ids"][:It].cpu().numpy(), flat_g[keep])
    assert np.array_equal(st_["isect_offsets"].cpu().numpy(), np.concatenate(([0], np.cumsum(np.bincount(tile_of[keep], minlength=off_g.size))[:-1])))


def test_front_reports_depth_outside_key_range(cuda):
    x = _inputs(cuda)
    env = _env(cuda)
    fr, _ = _front(x, env, cuda)
    lo, hi = 0xffffffff - int(fr.host_counts[2]), int(fr.host_counts[3])
    status = torch.zeros(4, dtype=torch.int64, device=cuda)
    _front(x, env, cuda, key_base=lo + 16, key_bits=24, status=status)             # the nearest Gaussians fall below the base
    assert status.cpu().tolist()[3] == 1
    status.zero_()
    _front(x, env, cuda, key_base=max(0, hi - (1 << 24) - 5), key_bits=24, status=status)   # the farthest ones beyond base + 2^24
    assert status.cpu().tolist()[3] == 1
    status.zero_()
    _front(x, env, cuda, key_base=lo, key_bits=24, status=status)
    assert status.cpu().tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("mode", ["pbr", "diffuse"])
def test_tail_equals_project_bwd_plus_shade_bwd(cuda, mode):
    import geosplatting_amd as gs
    from geosplatting_amd import _lib as L
    from geosplatting_amd import front as F
    from geosplatting_amd.shading import _MODE
    lib = L.lib()
    x = _inputs(cuda, level=4, res=160)
    env = _env(cuda)
    fr, e = _front(x, env, cuda, mode=mode)
    V = int(fr.host_counts[0])
    N = x["means"].shape[0]
    g = torch.Generator().manual_seed(7)
    stride = lib.gs_raster_grad_stride(3)
    v_packed = torch.zeros(V, stride, device=cuda)
    v_packed[:, :9] = (torch.rand(V, 9, generator=g) * 2 - 1).to(cuda)
    v_packed[::7] = 0.0                                                       # Gaussians that reached no pixel
    v_packed[1::11, 6:9] = 0.0                                                # geometry gradient only
    # reference: the two call-shaped kernels
    col, meta = _reference_meta(x, env, cuda, mode)
    f32 = torch.float32
    z = lambda *s: torch.zeros(*s, dtype=f32, device=cuda)
    r = dict(means=z(N, 3), quats=z(N, 4), scales=z(N, 3), opac=z(N), normals=z(N, 3), kd=z(N, 3), ks=z(N, 2))
    g_colors = torch.empty(N, 3, device=cuda)
    gids = meta["gaussian_ids"].int().contiguous()
    L.check(lib.gs_project_bwd(N, V, 3, L.ptr(x["means"]), L.ptr(x["quats"]), L.ptr(x["scales"]), L.ptr(x["opac"]), L.ptr(x["vm"]), L.ptr(x["K"]),
                               x["W"], x["H"], L.f32(0.3), L.ptr(gids), L.ptr(meta["conics"]),
                               L.ptr(meta["compensations"]), L.ptr(v_packed), stride, None, L.ptr(r["means"]), L.ptr(r["quats"]),
                               L.ptr(r["scales"]), L.ptr(r["opac"]), L.ptr(g_colors), 1, L.stream()), "gs_project_bwd")
    rb, rl = torch.zeros_like(env.base), [torch.zeros_like(l) for l in env.levels]
    eg = L.GsEnvGrad(); eg.base = rb.data_ptr()
    for i, t in enumerate(rl):
        eg.levels[i] = t.data_ptr()
    L.check(lib.gs_shade_bwd(N, L.ptr(x["means"]), L.ptr(x["normals"]), L.ptr(x["kd"]), L.ptr(x["ks"]), L.ptr(x["cam_pos"]), L.f32(0.1),
                             L.f32(1.0), _MODE[mode], C.byref(e), L.ptr(g_colors), L.ptr(r["means"]), L.ptr(r["normals"]), L.ptr(r["kd"]),
                             L.ptr(r["ks"]), C.byref(eg), 1, None, C.c_size_t(0), L.stream()), "gs_shade_bwd")
    # fused tail
    t = dict(means=z(N, 3), quats=z(N, 4), scales=z(N, 3), opac=z(N), normals=z(N, 3), kd=z(N, 3), ks=z(N, 2))
    tb, tl = torch.zeros_like(env.base), [torch.zeros_like(l) for l in env.levels]
    eg2 = L.GsEnvGrad(); eg2.base = tb.data_ptr()
    for i, tt in enumerate(tl):
        eg2.levels[i] = tt.data_ptr()
    F.tail_stage(V, None, x["means"], x["quats"], x["scales"], x["opac"], x["normals"], x["kd"], x["ks"], x["vm"], x["K"], x["cam_pos"], e, eg2,
                 x["W"], x["H"], 0.1, 1.0, _MODE[mode], fr.vis, v_packed, t["means"], t["quats"], t["scales"], t["opac"], t["normals"],
                 t["kd"], t["ks"])
    torch.cuda.synchronize()
    for k in r:
        a, b = t[k].cpu().numpy(), r[k].cpu().numpy()
        assert np.abs(b).max() > 0, k
        assert rel_err(a, b) < 1e-5, k
    for a, b in zip([tb] + tl, [rb] + rl):
        if float(b.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0
        else:
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
    # capacity form: the count read on the device
    t2 = dict(means=z(N, 3), quats=z(N, 4), scales=z(N, 3), opac=z(N), normals=z(N, 3), kd=z(N, 3), ks=z(N, 2))
    tb2, tl2 = torch.zeros_like(env.base), [torch.zeros_like(l) for l in env.levels]
    eg3 = L.GsEnvGrad(); eg3.base = tb2.data_ptr()
    for i, tt in enumerate(tl2):
        eg3.levels[i] = tt.data_ptr()
    vp = torch.zeros(N, stride, device=cuda); vp[:V] = v_packed
    F.tail_stage(N, fr.counts, x["means"], x["quats"], x["scales"], x["opac"], x["normals"], x["kd"], x["ks"], x["vm"], x["K"], x["cam_pos"], e,
                 eg3, x["W"], x["H"], 0.1, 1.0, _MODE[mode], fr.vis, vp, t2["means"], t2["quats"], t2["scales"], t2["opac"], t2["normals"],
                 t2["kd"], t2["ks"])
    torch.cuda.synchronize()
    for k in r:
        assert rel_err(t2[k].cpu().numpy(), r[k].cpu().numpy()) < 1e-5, k
    # XCD-private copies of the mid-sized levels (here the 64^2 level): accumulate over two calls, fold once
    t3 = dict(means=z(N, 3), quats=z(N, 4), scales=z(N, 3), opac=z(N), normals=z(N, 3), kd=z(N, 3), ks=z(N, 2))
    tb3, tl3 = torch.zeros_like(env.base), [torch.zeros_like(l) for l in env.levels]
    eg4 = L.GsEnvGrad(); eg4.base = tb3.data_ptr()
    for i, tt in enumerate(tl3):
        eg4.levels[i] = tt.data_ptr()
    priv = F.tail_priv_alloc(e, _MODE[mode], cuda)
    assert (priv is not None) == (mode != "diffuse")
    for _ in range(2):
        F.tail_stage(V, None, x["means"], x["quats"], x["scales"], x["opac"], x["normals"], x["kd"], x["ks"], x["vm"], x["K"], x["cam_pos"], e,
                     eg4, x["W"], x["H"], 0.1, 1.0, _MODE[mode], fr.vis, v_packed, t3["means"], t3["quats"], t3["scales"], t3["opac"],
                     t3["normals"], t3["kd"], t3["ks"], priv=priv)
    F.tail_priv_reduce(e, eg4, _MODE[mode], priv)
    torch.cuda.synchronize()
    for a, b in zip([tb3] + tl3, [rb] + rl):
        if float(b.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0
        else:
            assert rel_err(a.cpu().numpy(), 2.0 * b.cpu().numpy()) < 1e-5


def test_cull_log_backward_equals_plain_backward(cuda):
    """compositor forward with the cull log + the log-driven backward against the plain pair (raster_fwd_window / raster_bwd_lanes2, which
    the rasterizer tests check against the oracle): image, alpha, last_ids bit for bit; gradient records to the order of the float atomics."""
    from geosplatting_amd import _lib as L
    from geosplatting_amd import front as F
    lib = L.lib()
    for level, res in ((4, 160), (5, 400)):
        x = _inputs(cuda, level=level, res=res)
        env = _env(cuda)
        fr, _ = _front(x, env, cuda)
        V, I = int(fr.host_counts[0]), int(fr.host_counts[1])
        state, _, _ = F.bin_stage(fr, None, None)
        W = H = res
        g = torch.Generator().manual_seed(3)
        v_img = (torch.rand(H, W, 4, generator=g) * 2 - 1).to(cuda)
        exposure = torch.tensor([1.3], device=cuda)
        rws = state["raster_ws"]
        outs = []
        for use_log in (False, True):
            render = torch.empty(H, W, 3, device=cuda); alphas = torch.empty(H, W, device=cuda)
            last = torch.empty(H, W, dtype=torch.int32, device=cuda); img = torch.empty(H, W, 4, device=cuda)
            v_packed = torch.zeros(V, lib.gs_raster_grad_stride(3), device=cuda); v_exp = torch.zeros(1, device=cuda)
            if use_log:
                log_ws = torch.empty(lib.gs_raster_log_ws_bytes(L.i64(I), W, H, 16), dtype=torch.uint8, device=cuda)
                L.check(lib.gs_raster_composite_tone_log(W, H, 16, V, None, L.i64(I), None, L.ptr(state["isect_offsets"]), L.ptr(render),
                                                         L.ptr(alphas), L.ptr(last), 1, L.ptr(exposure), L.ptr(img), L.ptr(rws),
                                                         C.c_size_t(rws.numel()), L.ptr(log_ws), C.c_size_t(log_ws.numel()), L.stream()), "fwd log")
                L.check(lib.gs_raster_bwd_tone_log_acc(W, H, 16, V, None, L.i64(I), None, L.ptr(state["isect_offsets"]), L.ptr(render),
                                                       L.ptr(alphas), L.ptr(last), 1, L.ptr(exposure), L.ptr(v_img), L.ptr(v_packed), L.ptr(v_exp),
                                                       L.ptr(rws), C.c_size_t(rws.numel()), L.ptr(log_ws), C.c_size_t(log_ws.numel()), L.stream()),
                        "bwd log")
            else:
                L.check(lib.gs_raster_composite_tone(W, H, 16, V, None, L.i64(I), None, L.ptr(state["isect_offsets"]), L.ptr(render), L.ptr(alphas),
                                                     L.ptr(last), 1, L.ptr(exposure), L.ptr(img), L.ptr(rws), C.c_size_t(rws.numel()), L.stream()), "fwd")
                L.check(lib.gs_raster_bwd_tone_acc(W, H, 16, V, None, L.i64(I), None, L.ptr(state["isect_offsets"]), L.ptr(render), L.ptr(alphas),
                                                   L.ptr(last), 1, L.ptr(exposure), L.ptr(v_img), L.ptr(v_packed), L.ptr(v_exp), L.ptr(rws),
                                                   C.c_size_t(rws.numel()), L.stream()), "bwd")
            torch.cuda.synchronize()
            outs.append((render, alphas, last, img, v_packed, v_exp))
        a, b = outs
        for k in range(4):
            assert torch.equal(a[k], b[k]), k
        assert float(a[4].abs().max()) > 0
        assert rel_err(b[4].cpu().numpy(), a[4].cpu().numpy()) < 2e-6
        # exposure gradient: ten thousand float atomics (one per quadrant wave) of both signs whose order is not fixed -- 1e-5 of the
        # result was seen between two runs of the SAME kernel
        assert abs(float(a[5]) - float(b[5])) <= 1e-4 * abs(float(a[5])) + 1e-4


@pytest.mark.parametrize("mode", ["pbr", "specular", "diffuse"])
def test_tail_multi_equals_sum_of_view_tails(cuda, mode):
    """gs_tail_bwd_multi over n views (n = 1, 2, 3, 5, 8: every lanes-per-Gaussian variant of the pair kernels, with idle lanes at 3
    and 5) == n gs_tail_bwd calls accumulated: parameter gradients and texel gradients.  The pair kernels contract the colour
    cotangent into the cube fetch and sum the views as a tree, gs_tail_bwd carries the Jacobians and sums in view order: 1e-5.
    accumulate=False overwrites, a second call ADDS, and the two halves called separately (gs_tail_bwd_multi_parts) give the same."""
    import geosplatting_amd as gs
    from geosplatting_amd import _lib as L
    from geosplatting_amd import front as F
    from geosplatting_amd.shading import _MODE, _make_env
    lib = L.lib()
    env = _env(cuda)
    e = _make_env(gs.get_fg_lut(cuda), env)
    M = _MODE[mode]
    stride = lib.gs_raster_grad_stride(3)
    g = torch.Generator().manual_seed(9)
    views, xs = [], []
    for view in range(8):
        x = _inputs(cuda, level=4, res=160, view=view)
        fr = F.front_stage(x["means"], x["quats"], x["scales"], x["opac"], x["normals"], x["kd"], x["ks"], x["vm"], x["K"], x["cam_pos"], e,
                           x["W"], x["H"], 0.1, 1.0, M, want_packed_index=True)
        torch.cuda.synchronize()
        V = int(fr.host_counts[0])
        vp = torch.zeros(V, stride, device=cuda)
        vp[:, :9] = (torch.rand(V, 9, generator=g) * 2 - 1).to(cuda)
        vp[::5] = 0.0                                                          # pairs without any gradient
        vp[1::7, :6] = 0.0                                                     # ... with a colour gradient only
        vp[2::7, 6:9] = 0.0                                                    # ... with a geometry gradient only
        views.append((x["vm"], x["K"], x["cam_pos"], fr.vis, vp, fr.packed_index, x["W"], x["H"]))
        xs.append((x, fr, V))
    x0 = xs[0][0]
    N = x0["means"].shape[0]
    pidx = views[0][5].cpu().numpy()
    assert (pidx >= 0).sum() == xs[0][2] and np.array_equal(np.sort(pidx[pidx >= 0]), np.arange(xs[0][2]))
    z = lambda *s: torch.zeros(*s, device=cuda)
    names = ("means", "quats", "scales", "opac", "normals", "kd", "ks")
    def fresh(fill=0.0):
        d = dict(means=z(N, 3), quats=z(N, 4), scales=z(N, 3), opac=z(N), normals=z(N, 3), kd=z(N, 3), ks=z(N, 2))
        for k in d:
            d[k].fill_(fill)
        return d, torch.zeros_like(env.base), [torch.zeros_like(l) for l in env.levels]
    def grad_struct(b, ls):
        eg = L.GsEnvGrad(); eg.base = b.data_ptr()
        for i, t in enumerate(ls):
            eg.levels[i] = t.data_ptr()
        return eg
    def multi(vs, t, egt, accumulate, parts=3, priv=None):
        F.tail_multi_stage(vs, x0["means"], x0["quats"], x0["scales"], x0["opac"], x0["normals"], x0["kd"], x0["ks"], e, egt, 0.1, 1.0, M,
                           t["means"], t["quats"], t["scales"], t["opac"], t["normals"], t["kd"], t["ks"], accumulate=accumulate, priv=priv,
                           parts=parts)
    def close(t, tb, tl, r, rb, rl, what):
        for k in names:
            assert rel_err(t[k].cpu().numpy(), r[k].cpu().numpy()) < 1e-5, (what, k)
        for a, b in zip([tb] + tl, [rb] + rl):
            if float(b.abs().max()) == 0.0:
                assert float(a.abs().max()) == 0.0, what
            else:
                assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5, what
    # reference: per-view tails accumulated, snapshots after 1, 2, 3, 5, 8 views
    r, rb, rl = fresh()
    egr = grad_struct(rb, rl)
    snaps = {}
    for i, ((x, fr, V), vw) in enumerate(zip(xs, views)):
        F.tail_stage(V, None, x["means"], x["quats"], x["scales"], x["opac"], x["normals"], x["kd"], x["ks"], x["vm"], x["K"], x["cam_pos"], e, egr,
                     x["W"], x["H"], 0.1, 1.0, M, fr.vis, vw[4], r["means"], r["quats"], r["scales"], r["opac"], r["normals"], r["kd"], r["ks"])
        if i + 1 in (1, 2, 3, 5, 8):
            snaps[i + 1] = ({k: v.clone() for k, v in r.items()}, rb.clone(), [l.clone() for l in rl])
    torch.cuda.synchronize()
    assert all(float(snaps[8][0][k].abs().max()) > 0 for k in ("means", "quats", "scales", "opac", "normals", "kd"))
    for n in (1, 2, 3, 5, 8):
        t, tb, tl = fresh(123.0)                                               # accumulate=False must overwrite
        priv = F.tail_priv_alloc(e, M, cuda) if n == 3 else None               # (optional XCD-private copies of the mid-sized levels)
        egt = grad_struct(tb, tl)
        multi(views[:n], t, egt, False, priv=priv)
        F.tail_priv_reduce(e, egt, M, priv)
        torch.cuda.synchronize()
        close(t, tb, tl, *snaps[n], ("n", n))
    # a second call adds: views 0-2, then views 3-4 on top == the first five
    t, tb, tl = fresh(123.0)
    egt = grad_struct(tb, tl)
    multi(views[:3], t, egt, False)
    multi(views[3:5], t, egt, True)
    torch.cuda.synchronize()
    close(t, tb, tl, *snaps[5], "3 + 2")
    # the two halves as separate calls (shading first: the projection half adds to its v_means)
    t, tb, tl = fresh(123.0)
    egt = grad_struct(tb, tl)
    multi(views, t, egt, False, parts=1)
    multi(views, t, egt, False, parts=2)
    torch.cuda.synchronize()
    close(t, tb, tl, *snaps[8], "parts")
    # background launches (parts + 4: half of the CUs) as the engine issues them: three views, three more on top, then the last two
    t, tb, tl = fresh(123.0)
    egt = grad_struct(tb, tl)
    multi(views[:3], t, egt, False, parts=7)
    multi(views[3:6], t, egt, True, parts=7)
    multi(views[6:], t, egt, True, parts=3)
    torch.cuda.synchronize()
    close(t, tb, tl, *snaps[8], "background 3 + 3 + 2")


def test_engine_equals_op_by_op_autograd_step(cuda, monkeypatch):
    """One step of the engine (fused front, cull-log compositor, batched tails under several tail schedules) against the op-by-op
    autograd step (RenderStep(fused=False) with GEOSPLAT_SPLAT=ops: shade -> rasterization -> tone_map, each checked against the oracle
    elsewhere): the same images bit for bit, gradients to the order of the float atomics; and a second step runs on the capacity
    protocol with 24-bit keys."""
    import geosplatting_amd.synthetic as syn
    from geosplatting_amd.engine import RenderStep, params_from_scene
    scene = syn.sphere_scene(4, seed=1, cubemap_res=64, device=cuda)
    cams = syn.blender_cameras(num=3, width=160, height=160)
    g = torch.Generator().manual_seed(0)
    ups = [(torch.rand(160, 160, 4, generator=g) * 2 - 1).to(cuda) for _ in cams]
    monkeypatch.setenv("GEOSPLAT_SPLAT", "ops")
    ref = RenderStep(params_from_scene(scene, cuda, exposure=1.2), fused=False)
    rg, ri = ref(cams, lambda i, img: ups[i], all_reduce=False, keep_images=True)
    torch.cuda.synchronize()
    rg = {k: v.clone() for k, v in rg.items()}; ri = [im.clone() for im in ri]
    monkeypatch.delenv("GEOSPLAT_SPLAT")
    for sched in ("auto", "1", "2", "3"):                              # 3 views: 2 + 1 (auto), 1 + 1 + 1, 2 + 1, one launch
        monkeypatch.setenv("GEOSPLAT_TAIL_BATCH", sched)
        step = RenderStep(params_from_scene(scene, cuda, exposure=1.2))
        for which in (0, 1):                                          # exact counts, then the capacity protocol
            grads, images = step(cams, lambda i, img: ups[i], all_reduce=False, keep_images=True)
            torch.cuda.synchronize()
            assert step.poll_capacity(wait=True) and step.truncated_steps == 0
            for a, b in zip(images, ri):
                assert torch.equal(a, b)
            for k in rg:
                # float atomics in another order, and the projection backward is compiled with contraction on: its fused
                # multiply-adds differ between the kernels it is inlined into (2.7e-5 on the scale gradients)
                err = rel_err(grads[k].cpu().numpy(), rg[k].cpu().numpy())
                # (quats / scales of flat disks: cancellation noise of a rotation gradient -- the 1e-4 bar of the full-size tests)
                assert err < (1e-4 if k in ("quats", "scales") else 5e-5), (sched, which, k, err)
        assert step._key_lo is not None and not step._key32 and step._i_cap is not None


@pytest.mark.parametrize("n,res,mode", [(20000, 256, "exact"), (60000, 64, "exact"), (60000, 64, "capacity"), (90000, 32, "capacity")])
def test_binning_24_bit_keys_equal_32_bit_keys(cuda, n, res, mode):
    """gs_isect_bin_front with 24-bit keys (three depth passes over `depth bits - base`) against 32-bit keys (four passes): flatten ids
    and offsets bit for bit, on scenes with very long tile segments (up to tens of thousands of entries), with ties (a plane of equal
    depths keeps slot order) and, in capacity mode, unused capacity behind the list.  (Round 5 built a tile-local depth order against
    this test -- emission in slot order, the segments sorted in LDS -- which passed it and was no faster: DESIGN.md section 6.)"""
    from geosplatting_amd import front as F
    sp, cam = random_case(n, res, seed=11)
    d = lambda t: t.to(cuda).contiguous()
    means = sp.means.clone()
    vm = cam.view_matrix
    k = n // 5
    means[:k] = means[:k] - (means[:k] @ vm[2, :3])[:, None] * vm[2, :3][None, :]          # equal camera depths: ties keep slot order
    g = torch.Generator().manual_seed(3)
    x = dict(means=d(means), quats=d(sp.quats), scales=d(sp.scales.exp()), opac=d(torch.sigmoid(sp.opacities).squeeze(-1)),
             normals=d(torch.nn.functional.normalize(torch.randn(sp.num, 3, generator=g), dim=-1)),
             kd=d(torch.rand(sp.num, 3, generator=g)), ks=d(torch.rand(sp.num, 2, generator=g)), vm=d(vm), K=d(cam.intrinsic_matrix),
             cam_pos=d(cam.c2w[:, 3]), W=res, H=res)
    env = _env(cuda)
    f32, _ = _front(x, env, cuda)
    V, I = int(f32.host_counts[0]), int(f32.host_counts[1])
    ref, _, _ = F.bin_stage(f32, None, None, prepare=False)
    lo, hi = 0xffffffff - int(f32.host_counts[2]), int(f32.host_counts[3])
    base = max(0, lo - 1024)
    assert hi - base < (1 << 24)
    status = torch.zeros(4, dtype=torch.int64, device=cuda)
    f24, _ = _front(x, env, cuda, key_base=base, key_bits=24, status=status)
    if mode == "exact":
        got, _, _ = F.bin_stage(f24, None, None, prepare=False)
    else:
        cap = ((int(I * 1.3) + 4095) // 4096) * 4096
        got, _, _ = F.bin_stage(f24, cap, status, prepare=False)
    torch.cuda.synchronize()
    off = ref["isect_offsets"].cpu().numpy()
    seg = np.diff(np.append(off, I))
    print(f"\n  V={V} I={I} tiles {off.size}: longest segment {seg.max()}")
    assert status.cpu().tolist() == [0, 0, 0, 0]
    assert np.array_equal(got["isect_offsets"].cpu().numpy(), off)
    assert np.array_equal(got["flatten_ids"][:I].cpu().numpy(), ref["flatten_ids"][:I].cpu().numpy())
