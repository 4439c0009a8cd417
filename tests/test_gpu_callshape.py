"""The reference's CALL SHAPE on the fused kernels (geosplatting_amd/viewbatch.py): a Python loop of ``RenderableAttrs.splat`` over the
views of a batch (rfstudio/model/geosplat.py:863-879) and one ``backward()`` (rfstudio/optim/optimizer.py:107).

  * against the op-by-op decomposition of the same call (GEOSPLAT_SPLAT=ops: shade -> rasterization -> tone_map, each checked against
    the oracle in tests/test_gpu_shading.py / test_gpu_rasterizer.py): images bit for bit, every gradient -- cubemap through the
    prefilter's autograd included -- to summation order;
  * the autograd contract of the shared gather node: backward over a subset of the views, one backward per view, a second step on
    new tensors, no_grad calls, exposure-only gradients;
  * the capacity protocol behind that call shape: a view that outgrows the learnt capacity raises GeoSplatCapacityError from
    backward(), and the repeated step is right.
Full-size comparisons with the oracle: tests/test_gpu_fullsize.py::test_step_fullsize_vs_oracle, ::test_view_fullsize_vs_oracle.
"""
import math

import numpy as np
import pytest
import torch

from tests.util import sphere_case

pytestmark = pytest.mark.gpu


def _leaves(sc, cuda):
    d = lambda x: x.clone().to(cuda).requires_grad_(True)
    sp = sc.splats

    class G:
        pass
    g = G(); g.means = d(sp.means); g.scales = d(sp.scales); g.quats = d(sp.quats); g.opacities = d(sp.opacities)
    leaves = dict(means=g.means, scales=g.scales, quats=g.quats, opacities=g.opacities, kd=d(sc.kd), ks=d(sc.ks), normals=d(sc.normals),
                  cubemap=d(sc.cubemap), exposure=torch.tensor(1.2, device=cuda, requires_grad=True))
    return g, leaves


def _cams(res, n=3):
    from geosplatting_amd.cameras import orbit_cameras
    return orbit_cameras(8, 4.0 * (2.0 / 3.0), 30.0, res, res, focal=0.5 * res / math.tan(0.5 * 0.6911112))[:n]


def _run(sc, cams, ups, cuda, backward="one", subset=None, mode="pbr", tone="naive"):
    import geosplatting_amd as gs
    g, lv = _leaves(sc, cuda)
    attrs = gs.RenderableAttrs(kd=lv["kd"], ks=lv["ks"], normals=lv["normals"])
    env = gs.as_splitsum(lv["cubemap"])
    imgs = [attrs.splat(g, [c], exposure=lv["exposure"], envmap=env, min_roughness=0.1, max_metallic=1.0, mode=mode, tone_type=tone)
            for c in cams]
    idx = list(range(len(cams))) if subset is None else subset
    if backward == "one":
        sum((imgs[i] * ups[i]).sum() for i in idx).backward()
    else:
        for i in idx:
            (imgs[i] * ups[i]).sum().backward(retain_graph=True)      # (retain: the prefilter's graph is shared by the views)
    torch.cuda.synchronize()
    return [i.detach().clone() for i in imgs], {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in lv.items()}


def _compare(got, want, tol=2e-5):
    for k, w in want.items():
        a = got[k]
        assert (a is None) == (w is None), k
        if w is None:
            continue
        scale = float(w.abs().max())
        if k == "quats" and want.get("scales") is not None:      # isotropic / flat splats: the rotation gradient is pure cancellation noise --
            scale = max(scale, float(want["scales"].abs().max()))   # measured against the natural size of a covariance-perturbation gradient
        err = float((a - w).abs().max()) / (scale + 1e-30)
        print(f"  {k:10s} {err:.2e}")
        assert err < (1e-4 if k in ("quats", "scales", "exposure") else tol), (k, err)


@pytest.mark.parametrize("mode,tone", [("pbr", "naive"), ("diffuse", "aces"), ("specular", "none")])
def test_fused_splat_equals_op_by_op(cuda, monkeypatch, mode, tone):
    import geosplatting_amd as gs
    sc, _ = sphere_case(4, 160, cubemap_res=64)
    cams = _cams(160)
    g = torch.Generator().manual_seed(2)
    ups = [(torch.rand(160, 160, 4, generator=g) * 2 - 1).to(cuda) for _ in cams]
    monkeypatch.setenv("GEOSPLAT_SPLAT", "ops")
    ref_i, ref_g = _run(sc, cams, ups, cuda, mode=mode, tone=tone)
    monkeypatch.setenv("GEOSPLAT_SPLAT", "fused")
    gs.viewbatch.reset()
    for it in range(2):                                               # first pass: exact counts; second: capacity protocol, 24-bit keys
        i, gr = _run(sc, cams, ups, cuda, mode=mode, tone=tone)
        for a, b in zip(i, ref_i):
            assert torch.equal(a, b)
        print(f"\n pass {it} ({mode}, {tone})")
        _compare(gr, ref_g)
    cap = gs.viewbatch._state(cuda).caps[(sc.splats.num, 160, 160)]
    assert cap.i_cap is not None and cap.max_i > 0


def test_autograd_contract_of_the_gather_node(cuda):
    """Subset backward, one backward per view and a plain single backward agree; untouched views cost nothing and break nothing."""
    import geosplatting_amd as gs
    sc, _ = sphere_case(3, 96, cubemap_res=64)
    cams = _cams(96, 4)
    g = torch.Generator().manual_seed(4)
    ups = [(torch.rand(96, 96, 4, generator=g) * 2 - 1).to(cuda) for _ in cams]
    gs.viewbatch.reset()
    _, all_one = _run(sc, cams, ups, cuda)
    _, all_each = _run(sc, cams, ups, cuda, backward="each")
    print("\n one backward per view vs one backward")
    _compare(all_each, all_one)
    _, sub = _run(sc, cams, ups, cuda, subset=[0, 2])
    _, a = _run(sc, [cams[0], cams[2]], [ups[0], ups[2]], cuda)
    print(" subset of a 4-view step vs a 2-view step")
    _compare(sub, a)


def test_no_grad_and_exposure_only(cuda):
    import geosplatting_amd as gs
    sc, _ = sphere_case(3, 96, cubemap_res=64)
    cam = _cams(96, 1)[0]
    g, lv = _leaves(sc, cuda)
    attrs = gs.RenderableAttrs(kd=lv["kd"], ks=lv["ks"], normals=lv["normals"])
    with torch.no_grad():
        env = gs.as_splitsum(lv["cubemap"])
        img0 = attrs.splat(g, [cam], exposure=lv["exposure"], envmap=env, min_roughness=0.1, max_metallic=1.0)
    assert not img0.requires_grad
    img1 = attrs.splat(g, [cam], exposure=lv["exposure"], envmap=env, min_roughness=0.1, max_metallic=1.0)
    assert torch.equal(img0, img1) and img1.requires_grad
    # only the exposure needs a gradient: no gather node, the compositor backward alone
    sp = sc.splats

    class G2:
        means = sp.means.to(cuda); scales = sp.scales.to(cuda); quats = sp.quats.to(cuda); opacities = sp.opacities.to(cuda)
    attrs2 = gs.RenderableAttrs(kd=sc.kd.to(cuda), ks=sc.ks.to(cuda), normals=sc.normals.to(cuda))
    e = torch.tensor(1.2, device=cuda, requires_grad=True)
    img2 = attrs2.splat(G2, [cam], exposure=e, envmap=env, min_roughness=0.1, max_metallic=1.0)
    assert torch.equal(img2, img0)
    v = torch.rand(96, 96, 4, device=cuda)
    (img2 * v).sum().backward()
    (img1 * v).sum().backward()
    assert abs(float(e.grad) - float(lv["exposure"].grad)) < 1e-4 * abs(float(e.grad))


def test_capacity_overflow_raises_in_backward_and_the_retry_is_right(cuda, monkeypatch):
    import geosplatting_amd as gs
    from geosplatting_amd.cameras import orbit_cameras
    sc, _ = sphere_case(4, 128, cubemap_res=64)
    far = orbit_cameras(4, 24.0, 30.0, 128, 128, focal=0.5 * 128 / math.tan(0.5 * 0.6911112))[:2]
    near = orbit_cameras(4, 2.667, 30.0, 128, 128, focal=0.5 * 128 / math.tan(0.5 * 0.6911112))[:2]     # the sphere fills the frame: several tiles per Gaussian
    g = torch.Generator().manual_seed(6)
    ups = [(torch.rand(128, 128, 4, generator=g) * 2 - 1).to(cuda) for _ in far]
    monkeypatch.setenv("GEOSPLAT_SPLAT", "ops")
    _, want = _run(sc, near, ups, cuda)
    monkeypatch.setenv("GEOSPLAT_SPLAT", "fused")
    gs.viewbatch.reset()
    _run(sc, far, ups, cuda)                                          # learns a capacity from the far views
    cap = gs.viewbatch._state(cuda).caps[(sc.splats.num, 128, 128)]
    small = cap.i_cap
    with pytest.raises(gs.viewbatch.GeoSplatCapacityError):
        _run(sc, near, ups, cuda)                                     # far more intersections than 1.25 x what the far views had
    assert cap.i_cap > small
    _, got = _run(sc, near, ups, cuda)                                # the retry runs with the raised capacity (32-bit keys if the range moved)
    print()
    _compare(got, want)


def test_steps_do_not_accumulate_memory(cuda):
    """A step that can no longer be joined must be collectable: step -> gather outputs -> grad_fn -> ctx -> step is a cycle through
    C++ ownership that Python's collector does not see (round 5: 56 MB per step leaked at the bench size before the gather outputs
    were dropped when the next step starts; scripts/soak_callshape.py)."""
    import gc
    import geosplatting_amd as gs
    sc, _ = sphere_case(4, 128, cubemap_res=64)
    cams = _cams(128, 3)
    g, lv = _leaves(sc, cuda)
    attrs = gs.RenderableAttrs(kd=lv["kd"], ks=lv["ks"], normals=lv["normals"])
    opt = torch.optim.SGD(list(lv.values()), lr=1e-9)              # new parameter versions every step: every step is a new _Step

    def step():
        opt.zero_grad(set_to_none=True)
        env = gs.as_splitsum(lv["cubemap"])
        imgs = [attrs.splat(g, [c], exposure=lv["exposure"], envmap=env, min_roughness=0.1, max_metallic=1.0) for c in cams]
        sum(i.sum() for i in imgs).backward()
        opt.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize(); gc.collect()
    base = torch.cuda.memory_allocated()
    for _ in range(40):
        step()
    torch.cuda.synchronize(); gc.collect()
    grown = torch.cuda.memory_allocated() - base
    assert grown < (1 << 20), f"{grown} bytes retained over 40 steps"


def test_fused_splat_on_large_overlapping_splats(cuda, monkeypatch):
    """A scene the bench does not look like: 12 000 random (isotropic) splats with footprints of tens of pixels (many tiles per Gaussian,
    hundreds of entries per pixel, long cull logs) through the fused call shape against the op-by-op decomposition."""
    import geosplatting_amd as gs
    from tests.util import random_case
    sp, _ = random_case(12000, 192, seed=17)
    g = torch.Generator().manual_seed(8)

    class Scene:
        pass
    sc = Scene(); sc.splats = sp
    sc.normals = torch.nn.functional.normalize(torch.randn(sp.num, 3, generator=g), dim=-1)
    sc.kd = torch.rand(sp.num, 3, generator=g); sc.ks = torch.rand(sp.num, 2, generator=g)
    from geosplatting_amd.synthetic import make_cubemap
    sc.cubemap = make_cubemap(64, seed=2)
    from geosplatting_amd.cameras import orbit_cameras
    cams = orbit_cameras(4, 3.0, 30.0, 192, 192, hfov_degree=40.0)[:2]
    ups = [(torch.rand(192, 192, 4, generator=g) * 2 - 1).to(cuda) for _ in cams]
    monkeypatch.setenv("GEOSPLAT_SPLAT", "ops")
    ref_i, ref_g = _run(sc, cams, ups, cuda)
    monkeypatch.setenv("GEOSPLAT_SPLAT", "fused")
    gs.viewbatch.reset()
    for it in range(2):
        i, gr = _run(sc, cams, ups, cuda)
        for a, b in zip(i, ref_i):
            assert torch.equal(a, b)
        print(f"\n pass {it}")
        _compare(gr, ref_g, tol=5e-5)
