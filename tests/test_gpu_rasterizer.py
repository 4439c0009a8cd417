"""GPU parity: HIP rasterizer (through the C-ABI, via geosplatting_amd.rasterization) vs the CPU oracle.

Bar (BASELINE.json north_star): bit-exact tile/sort indices; rendered RGB and grads within 1e-4 relative fp32.
"""
import sys

import numpy as np
import pytest
import torch

import oracle
from tests.util import activated, random_case, rel_err, sphere_case

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4      # max-norm relative, fp32
GRAD_TOL = 1e-4


def _run_case(cuda, means, quats, scales, opac, colors, cam, check_grads=True, background=None):
    import geosplatting_amd as gs
    W, H = cam.width, cam.height
    vm, K = cam.view_matrix.numpy(), cam.intrinsic_matrix.numpy()
    ref = oracle.rasterization(means, quats, scales, opac, colors, vm, K, W, H, background=background)

    t = lambda a: torch.tensor(a, device=cuda, requires_grad=True)
    tm, tq, ts, to, tc = t(means), t(quats), t(scales), t(opac), t(colors)
    bg = None if background is None else torch.tensor(background, device=cuda)
    render, alpha, meta = gs.rasterization(tm, tq, ts, to, tc, torch.tensor(vm, device=cuda)[None],
                                           torch.tensor(K, device=cuda)[None], W, H, backgrounds=bg)
    assert render.shape == (1, H, W, colors.shape[1]) and alpha.shape == (1, H, W, 1)

    # ---- bit-exact integer / index work
    for key in ("gaussian_ids", "radii", "tiles_per_gauss", "isect_ids", "flatten_ids"):
        got = meta[key].cpu().numpy()
        assert got.shape == ref[key].shape, key
        assert np.array_equal(got.astype(np.int64), ref[key].astype(np.int64)), f"{key} not bit-exact"
    assert np.array_equal(meta["isect_offsets"].cpu().numpy().reshape(-1), ref["isect_offsets"].reshape(-1))
    # depth bits are part of the sort key -> depths must be bit-exact too
    assert np.array_equal(meta["depths"].cpu().numpy().view(np.int32), ref["depths"].view(np.int32))
    # ---- floating point per-Gaussian
    assert np.array_equal(meta["means2d"].cpu().numpy(), ref["means2d"])         # canonical op order: exact
    assert rel_err(meta["conics"].cpu().numpy(), ref["conics"]) < 1e-6
    assert rel_err(meta["opacities"].cpu().numpy(), ref["opacities"]) < 1e-6
    # ---- image: the compositor evaluates exp(-sigma), 1 / (1 - alpha) and the recurrence in the oracle's operation order, so
    # the image, the accumulated alpha and the index of the last composited Gaussian are BIT-IDENTICAL -- no "ambiguous
    # pixel" mask, no mismatch budget (the oracle still reports pixels with a threshold inside a 1e-5 band; informational)
    r = render[0].detach().cpu().numpy(); a = alpha[0, ..., 0].detach().cpu().numpy()
    last = meta["last_ids"][0].cpu().numpy()
    assert np.array_equal(last, ref["last_ids"]), f"last_ids differ on {(last != ref['last_ids']).sum()} pixels"
    assert np.array_equal(a, ref["alphas"]), f"alphas differ on {(a != ref['alphas']).sum()} pixels, max {np.abs(a - ref['alphas']).max():.3e}"
    assert np.array_equal(r, ref["render"]), f"render differs on {(r != ref['render']).sum()} values, max {np.abs(r - ref['render']).max():.3e}"

    if not check_grads:
        return ref, meta
    g = torch.Generator().manual_seed(5)
    vr = (torch.rand(H, W, colors.shape[1], generator=g) * 2 - 1)
    va = (torch.rand(H, W, generator=g) * 2 - 1)
    (render[0] * vr.to(cuda)).sum().add((alpha[0, ..., 0] * va.to(cuda)).sum()).backward()
    gref = oracle.rasterization_bwd(means, quats, scales, opac, colors, vm, K, W, H, ref, vr.numpy(), va.numpy(),
                                    background=background)
    for name, tens in (("v_means", tm), ("v_quats", tq), ("v_scales", ts), ("v_opacities", to), ("v_colors", tc)):
        got, want = tens.grad.cpu().numpy(), gref[name]
        scale = np.abs(want).max()
        if name == "v_quats":
            # isotropic Gaussians have d/dquat == 0 up to cancellation noise: measure against the natural size of
            # a covariance-perturbation gradient (|v_scales * scales|) instead of against that noise
            scale = max(scale, np.abs(gref["v_scales"] * scales).max())
        e = float(np.abs(got.astype(np.float64) - want).max() / (scale + 1e-30))
        assert e < GRAD_TOL, f"{name}: rel err {e:.3e}"
    return ref, meta


def test_sphere_surface_splats(cuda):
    """GeoSplatting-like flat opaque disks (MGAdapter on an icosphere), 7 680 Gaussians, 128x128."""
    sc, cam = sphere_case(3, 128)
    means, quats, scales, opac = activated(sc.splats)
    colors = torch.rand(sc.splats.num, 3, generator=torch.Generator().manual_seed(2)).numpy()
    ref, _ = _run_case(cuda, means, quats, scales, opac, colors, cam)
    assert len(ref["gaussian_ids"]) == sc.splats.num


@pytest.mark.parametrize("view", [0, 1, 2, 3])
def test_config0_random_splats(cuda, view):
    """BASELINE.json configs[0]: 10k random Gaussians, 256x256, 4 orbit views."""
    sp, cam = random_case(10000, 256, view=view)
    means, quats, scales, opac = activated(sp)
    _run_case(cuda, means, quats, scales, opac, sp.colors.numpy(), cam)


def test_anisotropic_random_splats(cuda):
    """Random splats with anisotropic scales and non-unit quaternions (exercises the quaternion VJP)."""
    sp, cam = random_case(6000, 160, view=2, seed=7)
    means, quats, scales, opac = activated(sp)
    g = torch.Generator().manual_seed(8)
    scales = (scales * torch.exp(torch.randn(6000, 3, generator=g) * 0.7).numpy()).astype(np.float32)
    quats = (quats * (0.5 + torch.rand(6000, 1, generator=g).numpy())).astype(np.float32)
    _run_case(cuda, means, quats, scales, np.clip(opac * 6, 0, 0.97).astype(np.float32), sp.colors.numpy(), cam)


def test_partial_visibility_and_ragged_image(cuda):
    """Camera inside the cloud (near-plane + frustum culling, huge screen-space radii) and W,H not multiples of 16."""
    import geosplatting_amd.synthetic as syn
    from geosplatting_amd.cameras import Camera, lookat_c2w
    sp = syn.random_splats(4000, seed=3)
    c2w = lookat_c2w(torch.tensor([0.2, 0.1, 0.3]), torch.tensor([0.0, 0.0, -1.0]), torch.tensor([0.0, 1.0, 0.0]))
    cam = Camera(c2w, 90.0, 95.0, 50.0, 37.0, 100, 75)
    means, quats, scales, opac = activated(sp)
    ref, _ = _run_case(cuda, means, quats, scales, np.clip(opac * 5, 0, 0.95).astype(np.float32), sp.colors.numpy(), cam)
    assert 0 < len(ref["gaussian_ids"]) < 4000


def test_many_tiles_and_screen_filling_gaussians(cuda):
    """1920x1080 = 120 x 68 = 8 160 tiles (13 tile bits: more than two 6-bit tile passes), with a few Gaussians scaled up until
    their rectangles cover most of the screen (maximum-size tile rectangles, lists that hold the same Gaussian in thousands of tiles)."""
    from geosplatting_amd.cameras import Camera, lookat_c2w
    import geosplatting_amd.synthetic as syn
    sp = syn.random_splats(3000, seed=12)
    c2w = lookat_c2w(torch.tensor([0.0, 0.3, 3.0]), torch.tensor([0.0, 0.0, 0.0]), torch.tensor([0.0, 1.0, 0.0]))
    cam = Camera(c2w, 1500.0, 1500.0, 960.0, 540.0, 1920, 1080)
    means, quats, scales, opac = activated(sp)
    scales = scales.copy(); opac = np.clip(opac * 4, 0, 0.9).astype(np.float32)
    scales[:6] *= 40.0                                            # six screen-filling splats
    opac[:6] = 0.05
    ref, meta = _run_case(cuda, means, quats, scales, opac, sp.colors.numpy(), cam)
    assert int(ref["tiles_per_gauss"].max()) > 2000 and len(ref["flatten_ids"]) > 20000


def test_background_and_channels(cuda):
    """D=5 channels (generic path) with a background colour."""
    sp, cam = random_case(3000, 96)
    means, quats, scales, opac = activated(sp)
    colors = torch.rand(3000, 5, generator=torch.Generator().manual_seed(4)).numpy()
    _run_case(cuda, means, quats, scales, opac, colors, cam, background=np.array([0.1, 0.2, 0.3, 0.4, 0.5], np.float32))


def test_empty_inputs(cuda):
    """No Gaussian / nothing visible -> zero image, alpha 0, empty meta."""
    import geosplatting_amd as gs
    from geosplatting_amd.cameras import orbit_cameras
    cam = orbit_cameras(1, 3.0, 0.0, 64, 48, hfov_degree=40.0)[0]
    vm = cam.view_matrix.to(cuda)[None]; K = cam.intrinsic_matrix.to(cuda)[None]
    for n, z in ((0, 0.0), (5, 100.0)):           # n=5 placed behind the camera
        means = torch.zeros(n, 3, device=cuda); means[:, 2] = z
        means[:, 0] = 50.0
        q = torch.tensor([1.0, 0, 0, 0], device=cuda).repeat(n, 1)
        r, a, meta = gs.rasterization(means, q, torch.full((n, 3), 0.01, device=cuda), torch.full((n,), 0.5, device=cuda),
                                      torch.rand(n, 3, device=cuda), vm, K, 64, 48)
        assert r.abs().max().item() == 0 and a.abs().max().item() == 0
        assert meta["flatten_ids"].numel() == 0 and meta["gaussian_ids"].numel() == 0


@pytest.mark.parametrize("mode", ["ED", "RGB+D"])
def test_depth_render_modes(cuda, mode):
    """render_mode 'ED' (GSplatter.render_depth, rfstudio/model/gsplat.py:151-172) and 'RGB+D': the camera-space depth
    is composited as one more channel; gradients reach means/quats/scales through v_depths."""
    import geosplatting_amd as gs
    sp, cam = random_case(3000, 96, view=1, seed=5)
    means, quats, scales, opac = activated(sp)
    opac = np.clip(opac * 5, 0, 0.9).astype(np.float32)
    colors = sp.colors.numpy()
    W = H = 96
    vm, K = cam.view_matrix.numpy(), cam.intrinsic_matrix.numpy()
    proj = oracle.project_fwd(means, quats, scales, vm, K, W, H)
    depth_full = np.zeros((means.shape[0], 1), np.float32)
    depth_full[proj["gaussian_ids"], 0] = proj["depths"]
    col_ref = depth_full if mode == "ED" else np.concatenate([colors, depth_full], 1)
    ref = oracle.rasterization(means, quats, scales, opac, col_ref, vm, K, W, H)
    t = lambda a: torch.tensor(a, device=cuda, requires_grad=True)
    tm, tq, ts, to, tc = t(means), t(quats), t(scales), t(opac), t(colors)
    render, alpha, meta = gs.rasterization(tm, tq, ts, to, tc, torch.tensor(vm, device=cuda)[None],
                                           torch.tensor(K, device=cuda)[None], W, H, render_mode=mode)
    want = ref["render"].copy()
    if mode == "ED":
        want = want / np.clip(ref["alphas"][..., None], 1e-10, None)
    assert render.shape == (1, H, W, col_ref.shape[1])
    assert np.array_equal(meta["last_ids"][0].cpu().numpy(), ref["last_ids"])
    assert np.array_equal(render[0].detach().cpu().numpy(), want)        # composited depth (and its division by alpha): bit-identical
    # gradient through the depth channel: compare d(sum accumulated depth)/d(means) with the oracle chain
    if mode == "RGB+D":
        g = torch.Generator().manual_seed(2)
        vr = (torch.rand(H, W, 4, generator=g) * 2 - 1)
        (render[0] * vr.to(cuda)).sum().backward()
        rb = oracle.raster_bwd(W, H, 16, ref["means2d"], ref["conics"], ref["opacities"], ref["colors"],
                               ref["isect_offsets"].reshape(-1), ref["flatten_ids"], ref["alphas"], ref["last_ids"],
                               vr.numpy(), np.zeros((H, W), np.float32))
        v_m2d, v_con, v_col, v_op = rb
        gref = oracle.project_bwd(means, quats, scales, opac, vm, K, W, H, ref["gaussian_ids"], ref["conics"],
                                  ref["compensations"], v_m2d, v_con, v_op, v_col[:, :3], v_depths=v_col[:, 3])
        assert rel_err(tm.grad.cpu().numpy(), gref[0]) < 1e-4
        assert rel_err(tc.grad.cpu().numpy(), gref[4]) < 1e-4


def test_deferred_14_channels(cuda):
    """D = 14 as in the reference's deferred call (rfstudio/model/geosplat.py:276-295)."""
    sp, cam = random_case(2500, 80, view=3, seed=9)
    means, quats, scales, opac = activated(sp)
    colors = torch.rand(2500, 14, generator=torch.Generator().manual_seed(1)).numpy()
    _run_case(cuda, means, quats, scales, np.clip(opac * 4, 0, 0.9).astype(np.float32), colors, cam)


@pytest.mark.parametrize("D", [4, 5, 8, 20, 32])
def test_feature_channels_all_templates(cuda, D):
    """Every D > 3 instantiation of the compositor (4, 8, 16, 32 channels; colours staged in LDS planes sized by the run-time D, more
    than 64 KB of dynamic LDS from D = 20 on) against the oracle, on a scene dense enough for several batches per quadrant."""
    sp, cam = random_case(3000, 96, view=1, seed=21 + D)
    means, quats, scales, opac = activated(sp)
    colors = torch.rand(3000, D, generator=torch.Generator().manual_seed(D)).numpy()
    _run_case(cuda, means, quats, scales, np.clip(opac * 4, 0, 0.9).astype(np.float32), colors, cam)


def test_error_behaviour(cuda):
    import geosplatting_amd as gs
    z = torch.zeros(1, 3, device=cuda)
    args = (z, torch.ones(1, 4, device=cuda), z + 1, torch.ones(1, device=cuda), z, torch.eye(4, device=cuda)[None],
            torch.eye(3, device=cuda)[None], 16, 16)
    with pytest.raises(ValueError):
        gs.rasterization(*args, render_mode="bogus")
    with pytest.raises(NotImplementedError):
        gs.rasterization(*args, rasterize_mode="classic")
    with pytest.raises(ValueError):
        gs.rasterization(z, torch.ones(1, 4, device=cuda), z + 1, torch.ones(1, device=cuda), z,
                         torch.eye(4, device=cuda).repeat(2, 1, 1), torch.eye(3, device=cuda).repeat(2, 1, 1), 16, 16)


@pytest.mark.parametrize("level", [6, 7])
def test_full_size_properties(cuda, level):
    """BASELINE sizes (491 520 and 1 966 080 surface splats, 800x800): size-independent properties -- sortedness of keys,
    stability (flatten_ids ascending inside equal keys), offsets monotone and consistent, alpha in [0,1],
    colour linearity of the compositor, and a sub-sampled oracle check of projection outputs."""
    import geosplatting_amd as gs
    sc, cam = sphere_case(level, 800)
    means, quats, scales, opac = activated(sc.splats)
    N = sc.splats.num
    g = torch.Generator().manual_seed(7)
    colors = torch.rand(N, 3, generator=g)
    t = lambda a: torch.tensor(a, device=cuda)
    vm = cam.view_matrix.to(cuda)[None]; K = cam.intrinsic_matrix.to(cuda)[None]
    r1, a1, meta = gs.rasterization(t(means), t(quats), t(scales), t(opac), colors.to(cuda), vm, K, 800, 800)
    ids = meta["isect_ids"]; flat = meta["flatten_ids"]
    assert bool((ids[1:] >= ids[:-1]).all())
    same = ids[1:] == ids[:-1]
    assert bool((flat[1:][same] > flat[:-1][same]).all())
    off = meta["isect_offsets"].reshape(-1).long()
    assert bool((off[1:] >= off[:-1]).all()) and int(off[0]) == 0 and int(off[-1]) <= ids.numel()
    tile_of = (ids >> 32)
    counts = torch.bincount(tile_of, minlength=off.numel())
    assert torch.equal(torch.cat([off[1:], torch.tensor([ids.numel()], device=cuda)]) - off, counts)
    assert int(meta["tiles_per_gauss"].sum()) == ids.numel()
    assert float(a1.min()) >= 0.0 and float(a1.max()) <= 1.0
    # linearity in colour: render(2c + 1) == 2 render(c) + alpha-weighted... (sum of weights = accumulated vis)
    r2, a2, _ = gs.rasterization(t(means), t(quats), t(scales), t(opac), (2 * colors).to(cuda), vm, K, 800, 800)
    assert torch.equal(a1, a2)
    assert float((r2 - 2 * r1).abs().max()) < 1e-5
    # projection outputs against the oracle on the whole set (cheap on CPU)
    ref = oracle.project_fwd(means, quats, scales, cam.view_matrix.numpy(), cam.intrinsic_matrix.numpy(), 800, 800)
    assert np.array_equal(meta["gaussian_ids"].cpu().numpy(), ref["gaussian_ids"].astype(np.int64))
    assert np.array_equal(meta["radii"].cpu().numpy(), ref["radii"])
    assert np.array_equal(meta["means2d"].cpu().numpy(), ref["means2d"])


@pytest.mark.parametrize("seed", [0, 3, 6, 9])
def test_seeded_random_scenes(cuda, seed):
    """anisotropic random splats with varied opacity, odd resolutions, optional background (scripts/stress_parity.py
    runs the same generator over many more seeds)"""
    g = torch.Generator().manual_seed(100 + seed)
    res = [64, 96, 128, 200][seed // 2 % 4]
    sp, cam = random_case([500, 2000, 5000][seed % 3], res, view=seed % 4, seed=seed + 7)
    sp.scales = sp.scales + torch.randn(sp.scales.shape, generator=g) * 0.5
    sp.opacities = torch.logit(torch.rand(sp.opacities.shape, generator=g) * 0.9 + 0.05)
    means, quats, scales, opac = activated(sp)
    colors = torch.rand(sp.num, 3, generator=g).numpy()
    bg = None if seed % 3 else np.array([0.2, 0.5, 0.9], np.float32)
    _run_case(cuda, means, quats, scales, opac, colors, cam, background=bg)


def test_binning_variants_bit_identical(cuda, monkeypatch):
    """depth-major binning (gs_isect_bin: Gaussians sorted by depth once, intersections emitted in that order, two stable tile
    passes) == the upstream call shape (gs_isect_emit + gs_isect_sort over the 44-bit keys), both hand-written radix code;
    ties in depth (duplicated Gaussians) keep their packed-index order"""
    import geosplatting_amd as gs
    sp, cam = random_case(20000, 320, view=1, seed=5)
    means, quats, scales, opac = activated(sp)
    means[1000:2000] = means[:1000]; quats[1000:2000] = quats[:1000]; scales[1000:2000] = scales[:1000]    # exact depth ties
    t = lambda a: torch.tensor(a, device=cuda)
    args = (t(means), t(quats), t(scales), t(opac), t(sp.colors.numpy()), cam.view_matrix.to(cuda)[None],
            cam.intrinsic_matrix.to(cuda)[None], 320, 320)
    out = {}
    for mode in ("depth_major", "emit_sort"):
        monkeypatch.setattr(sys.modules["geosplatting_amd.rasterization"], "BINNING", mode)
        r, a, meta = gs.rasterization(*args)
        out[mode] = (r, a, meta)
    for key in ("isect_ids", "flatten_ids", "isect_offsets", "last_ids"):
        assert torch.equal(out["depth_major"][2][key], out["emit_sort"][2][key]), key
    assert torch.equal(out["depth_major"][0], out["emit_sort"][0])
    ref = oracle.rasterization(means, quats, scales, opac, sp.colors.numpy(), cam.view_matrix.numpy(), cam.intrinsic_matrix.numpy(), 320, 320)
    assert np.array_equal(out["depth_major"][2]["isect_ids"].cpu().numpy(), ref["isect_ids"])
    assert np.array_equal(out["depth_major"][2]["flatten_ids"].cpu().numpy(), ref["flatten_ids"])


def test_binning_without_isect_ids_bit_identical(cuda):
    """gs_isect_bin_tiles_cap + gs_isect_offsets_tiles_cap (what the step engine runs: int32 tile ids instead of the 64-bit keys, no
    depth gather in the last pass) == gs_isect_bin_cap + gs_isect_offsets_cap: same sorted flatten ids, same tile offsets, and the
    tile ids are the high halves of the keys."""
    import importlib
    R = importlib.import_module("geosplatting_amd.rasterization")      # (the package attribute of that name is the function)
    sp, cam = random_case(20000, 320, view=2, seed=11)
    means, quats, scales, opac = activated(sp)
    t = lambda a: torch.tensor(a, device=cuda)
    out = {}
    for want_ids in (True, False):
        pr = R._project_stage(t(means), t(quats), t(scales), t(opac), t(sp.colors.numpy()), cam.view_matrix.to(cuda), cam.intrinsic_matrix.to(cuda),
                              320, 320, 16, 0.3, 0.01, 1e10, 0.0)
        pr.event.synchronize()
        V, I = (int(x) for x in pr.host_counts.tolist())
        status = torch.zeros(3, dtype=torch.int64, device=cuda)
        state, *_ = R._bin_stage_cap(pr, I + 1000, status, want_ids=want_ids)
        torch.cuda.synchronize()
        assert int(status[0]) == 0
        out[want_ids] = (state["isect_ids"][:I].clone(), state["flatten_ids"][:I].clone(), state["isect_offsets"].clone())
    assert I > 50000
    assert torch.equal(out[True][1], out[False][1]) and torch.equal(out[True][2], out[False][2])
    assert out[False][0].dtype == torch.int32 and torch.equal((out[True][0] >> 32).int(), out[False][0])


@pytest.mark.parametrize("tone", ["naive", "aces", "none"])
@pytest.mark.parametrize("capacity", [False, True])
def test_compositor_with_tone_mapping_inside(cuda, tone, capacity):
    """gs_raster_composite_tone == gs_raster_composite + gs_tonemap_fwd3 (bit-identical image and raw outputs) and
    gs_raster_bwd_tone_acc == gs_tonemap_bwd3 + gs_raster_bwd_acc (the same v_render / v_alpha per pixel, so the packed gradients
    differ only by the order of their fp32 atomics; the exposure gradient by the order of its partial sums), with exact counts and
    with device-side counts."""
    import ctypes as C
    import importlib
    from geosplatting_amd import _lib as L
    R = importlib.import_module("geosplatting_amd.rasterization")
    lib = L.lib()
    mode = {"none": 0, "naive": 1, "aces": 2}[tone]
    sp, cam = random_case(20000, 320, view=1, seed=23)
    means, quats, scales, opac = activated(sp)
    t = lambda a: torch.tensor(a, device=cuda)
    W = H = 320
    pr = R._project_stage(t(means), t(quats), t(scales), t(opac), t(sp.colors.numpy()), cam.view_matrix.to(cuda), cam.intrinsic_matrix.to(cuda),
                          W, H, 16, 0.3, 0.01, 1e10, 0.0)
    pr.event.synchronize()
    V, I = (int(x) for x in pr.host_counts.tolist())
    if capacity:
        status = torch.zeros(3, dtype=torch.int64, device=cuda)
        state, Vc, Ic, D, whs = R._bin_stage_cap(pr, I + 777, status, want_ids=False)
        state = R._prepare_stage_cap(state, Vc, Ic, D, whs)
        render, alphas, st = R._composite_stage_cap(state, Vc, Ic, D, whs, None)
        counts = L.ptr(st["counts"])
    else:
        state, Vc, Ic, D, whs = R._bin_stage(pr)
        state = R._prepare_stage(state, Vc, Ic, D, whs)
        render, alphas, st, _, _ = R._composite_stage(state, Vc, Ic, D, whs, None)
        counts = None
    f32 = torch.float32
    stream = lambda: L.stream()
    exposure = torch.tensor([1.7], device=cuda)
    P = W * H
    img = torch.empty(H, W, 4, dtype=f32, device=cuda)
    L.check(lib.gs_tonemap_fwd3(L.i64(P), mode, L.ptr(render), L.ptr(alphas), L.ptr(exposure), L.ptr(img), stream()), "gs_tonemap_fwd3")
    # fused forward
    render2 = torch.empty_like(render); alphas2 = torch.empty_like(alphas); last2 = torch.empty_like(st["last_ids"]); img2 = torch.empty_like(img)
    rws = st["raster_ws"]
    L.check(lib.gs_raster_composite_tone(W, H, 16, Vc, L.ptr(st["colors"]), L.i64(Ic), counts, L.ptr(st["isect_offsets"]), L.ptr(render2),
                                         L.ptr(alphas2), L.ptr(last2), mode, L.ptr(exposure), L.ptr(img2), L.ptr(rws), C.c_size_t(rws.numel()),
                                         stream()), "gs_raster_composite_tone")
    torch.cuda.synchronize()
    assert float(alphas.max()) > 0.5
    assert torch.equal(render, render2) and torch.equal(alphas, alphas2) and torch.equal(st["last_ids"], last2)
    assert torch.equal(img, img2)
    # backward
    g = torch.Generator(device=cuda).manual_seed(3)
    v_img = torch.rand(H, W, 4, device=cuda, generator=g) * 2 - 1
    stride = lib.gs_raster_grad_stride(3)
    v_render = torch.empty(H, W, 3, dtype=f32, device=cuda); v_alpha = torch.empty(H, W, dtype=f32, device=cuda)
    v_exp = torch.zeros(1, device=cuda); v_exp2 = torch.zeros(1, device=cuda)
    vp = torch.zeros(Vc, stride, dtype=f32, device=cuda); vp2 = torch.zeros_like(vp)
    L.check(lib.gs_tonemap_bwd3(L.i64(P), mode, L.ptr(render), L.ptr(alphas), L.ptr(exposure), L.ptr(v_img), L.ptr(v_render), L.ptr(v_alpha),
                                L.ptr(v_exp), 1, stream()), "gs_tonemap_bwd3")
    if capacity:
        L.check(lib.gs_raster_bwd_acc_cap(W, H, 16, 3, Vc, L.ptr(st["colors"]), None, L.i64(Ic), counts, L.ptr(st["isect_offsets"]), L.ptr(alphas),
                                          L.ptr(st["last_ids"]), L.ptr(v_render), L.ptr(v_alpha), L.ptr(vp), L.ptr(rws), C.c_size_t(rws.numel()),
                                          stream()), "gs_raster_bwd_acc_cap")
    else:
        L.check(lib.gs_raster_bwd_acc(W, H, 16, 3, Vc, L.ptr(st["colors"]), None, L.i64(Ic), L.ptr(st["isect_offsets"]), L.ptr(alphas),
                                      L.ptr(st["last_ids"]), L.ptr(v_render), L.ptr(v_alpha), L.ptr(vp), L.ptr(rws), C.c_size_t(rws.numel()),
                                      stream()), "gs_raster_bwd_acc")
    L.check(lib.gs_raster_bwd_tone_acc(W, H, 16, Vc, L.ptr(st["colors"]), L.i64(Ic), counts, L.ptr(st["isect_offsets"]), L.ptr(render),
                                       L.ptr(alphas), L.ptr(st["last_ids"]), mode, L.ptr(exposure), L.ptr(v_img), L.ptr(vp2), L.ptr(v_exp2),
                                       L.ptr(rws), C.c_size_t(rws.numel()), stream()), "gs_raster_bwd_tone_acc")
    torch.cuda.synchronize()
    scale = float(vp[:V].abs().max())
    assert scale > 0
    assert float((vp[:V] - vp2[:V]).abs().max()) <= 2e-6 * scale
    # the exposure gradient is a sum of 102 400 terms of both signs: compare against the size of the terms, not of the (cancelled) sum,
    # and against the same sum in float64
    terms = (v_render.double() * render.double()).sum(-1) / float(exposure) + (v_alpha.double() * alphas.double() / float(exposure) if mode == 0 else 0.0)
    ref, size = float(terms.sum()), float(terms.abs().sum())
    assert abs(float(v_exp) - ref) <= 2e-6 * size and abs(float(v_exp2) - ref) <= 2e-6 * size, (float(v_exp), float(v_exp2), ref, size)
    # a background is refused (the fused forms are for RenderableAttrs.splat, which passes none) -- through the plain entry point
    # they wrap, D != 3 cannot even be expressed


def test_rcp_exact_exhaustive(cuda):
    """The compositor backward's 1 / (1 - alpha) is a hardware reciprocal plus two correction steps; the oracle (and
    gsplat) divide.  Every float of [2^-11, 1] -- the range 1 - alpha can take is [0.001, 0.9961] -- must give the
    correctly rounded quotient, so the two stay bit-identical."""
    import ctypes as C
    import struct
    from geosplatting_amd import _lib as L
    lo = struct.unpack("<I", struct.pack("<f", 2.0 ** -11))[0]
    hi = struct.unpack("<I", struct.pack("<f", 1.0))[0]
    bad = torch.zeros(1, dtype=torch.int64, device=cuda)
    rc = L.lib().gs_selftest_rcp(C.c_uint32(lo), C.c_uint32(hi), C.c_void_p(bad.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.lib().gs_last_error()
    torch.cuda.synchronize()
    assert int(bad.item()) == 0


def test_canonical_exp_equals_oracle_exhaustive(cuda):
    """gs_exp_neg (HIP compositor) == gso_exp_neg (oracle) on EVERY float of [0, 16] (order-independent checksum of the result
    bits computed on both sides), and its distance from the float64 exponential (see tests/test_oracle_cpu.py for what the
    number means: the same argument rounding as gsplat's `__expf`)."""
    import ctypes as C
    import struct
    from geosplatting_amd import _lib as L
    bits = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]
    out = torch.zeros(2, dtype=torch.int64, device=cuda)
    for lo, hi, bar in ((0.0, 5.55, 5e-7), (0.0, 16.0, 1e-6)):
        rc = L.lib().gs_selftest_exp(C.c_uint32(bits(lo)), C.c_uint32(bits(hi)), C.c_void_p(out.data_ptr()),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, L.lib().gs_last_error()
        torch.cuda.synchronize()
        cs, worst_bits = (int(v) & 0xFFFFFFFFFFFFFFFF for v in out.tolist())
        worst = struct.unpack("<d", struct.pack("<Q", worst_bits))[0]
        want_worst, want_cs = oracle.exp_neg_check(lo, hi)
        print(f"\n  exp(-sigma) on [{lo}, {hi}]: max rel err vs float64 exp {worst:.3e} (oracle {want_worst:.3e}), checksums equal: {cs == want_cs}")
        assert cs == want_cs, "gs_exp_neg and gso_exp_neg differ somewhere in the range"
        assert worst < bar and abs(worst - want_worst) < 1e-9
