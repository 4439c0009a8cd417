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
    cap = gs.viewbatch._state(cuda).caps[(160, 160)]
    assert cap.i_cap(sc.splats.num) is not None and cap.max_i > 0


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
    cap = gs.viewbatch._state(cuda).caps[(128, 128)]
    small = cap.i_cap(sc.splats.num)
    with pytest.raises(gs.viewbatch.GeoSplatCapacityError):
        _run(sc, near, ups, cuda)                                     # far more intersections than 1.5 x what the far views had
    assert cap.i_cap(sc.splats.num) > small
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


def test_first_step_is_exact_and_capacity_follows_n(cuda):
    """every view of the first step at an image size runs exact (no view of it can overflow a capacity learnt from one
    view), later steps run on the ratio learnt -- also with a DIFFERENT number of Gaussians (a stage-1 loop), and the table of
    capacities is bounded."""
    import geosplatting_amd as gs
    vb = gs.viewbatch
    sc, _ = sphere_case(4, 128, cubemap_res=64)
    cams = _cams(128, 3)
    g = torch.Generator().manual_seed(3)
    ups = [(torch.rand(128, 128, 4, generator=g) * 2 - 1).to(cuda) for _ in cams]
    vb.reset()
    seen = []
    orig = vb._view_forward

    def spy(step, cam, exposure, tone, want_grad):
        img, v = orig(step, cam, exposure, tone, want_grad)
        seen.append(v.exact)
        return img, v
    vb._view_forward = spy
    try:
        _run(sc, cams, ups, cuda)
        assert seen == [True, True, True]                            # the whole first step
        del seen[:]
        _run(sc, cams, ups, cuda)
        assert seen == [False, False, False]
        sc5, _ = sphere_case(5, 128, cubemap_res=64)                  # four times the Gaussians, same image size: capacity scales with N
        del seen[:]
        _run(sc5, cams, ups, cuda)
        assert seen == [False, False, False]
    finally:
        vb._view_forward = orig
    st = vb._state(cuda)
    assert list(st.caps.keys()) == [(128, 128)]
    r = st.caps[(128, 128)]
    assert r.i_cap(sc5.splats.num) > 3 * r.i_cap(sc.splats.num) // 1
    for k in range(vb._MAX_CAPS + 8):                                  # bounded: the least recently used sizes go when a new one arrives
        st.caps[(10000 + k, 1)] = vb._Capacity()
    st.caps.move_to_end((128, 128))
    _run(sc, _cams(96, 1), [torch.rand(96, 96, 4, device=cuda)], cuda)
    assert len(st.caps) == vb._MAX_CAPS and (128, 128) in st.caps and (96, 96) in st.caps
    vb.reset()


def test_backward_that_skips_the_gather_node_leaves_no_stale_sums(cuda):
    """a backward pass that runs view nodes but not the gather node (autograd.grad w.r.t. the exposure only) must not leave
    its partial sums behind for the next pass over the same step."""
    import geosplatting_amd as gs
    sc, _ = sphere_case(3, 96, cubemap_res=64)
    cams = _cams(96, 2)
    g, lv = _leaves(sc, cuda)
    attrs = gs.RenderableAttrs(kd=lv["kd"], ks=lv["ks"], normals=lv["normals"])
    gen = torch.Generator().manual_seed(5)
    ups = [(torch.rand(96, 96, 4, generator=gen) * 2 - 1).to(cuda) for _ in cams]
    gs.viewbatch.reset()
    env = gs.as_splitsum(lv["cubemap"])
    imgs = [attrs.splat(g, [c], exposure=lv["exposure"], envmap=env, min_roughness=0.1, max_metallic=1.0) for c in cams]
    # pass 1: view 0 only, w.r.t. the exposure only: the gather node is not part of this pass
    ge, = torch.autograd.grad((imgs[0] * ups[0]).sum(), [lv["exposure"]], retain_graph=False)
    step = gs.viewbatch._state(cuda).current
    assert step.g is None and step.pending == []
    # pass 2: view 1 into every parameter == a fresh one-view step on view 1
    (imgs[1] * ups[1]).sum().backward()
    got = {k: (v.grad.clone() if v.grad is not None else None) for k, v in lv.items()}
    _, want = _run(sc, [cams[1]], [ups[1]], cuda)
    print()
    _compare(got, want)


def test_backward_on_a_worker_thread_while_the_next_step_is_issued(cuda):
    """what a prefetching trainer does -- `backward()` of step k on a worker thread while the main thread already
    issues the `splat()` calls of step k + 1 on the same device.  Gradients equal the serial run; an overflow of step k is raised from
    ITS backward (on the worker), never from step k + 1's forward."""
    import threading
    import geosplatting_amd as gs
    vb = gs.viewbatch
    sc, _ = sphere_case(4, 128, cubemap_res=64)
    cams = _cams(128, 4)
    gen = torch.Generator().manual_seed(12)
    ups = [(torch.rand(128, 128, 4, generator=gen) * 2 - 1).to(cuda) for _ in cams]
    vb.reset()
    _, want = _run(sc, cams, ups, cuda)                                # serial reference (also teaches the capacity: exact step)
    _, want2 = _run(sc, cams, ups, cuda)                               # capacity step, serial
    _compare(want2, want)

    def forward(lv, g):
        attrs = gs.RenderableAttrs(kd=lv["kd"], ks=lv["ks"], normals=lv["normals"])
        env = gs.as_splitsum(lv["cubemap"])
        imgs = [attrs.splat(g, [c], exposure=lv["exposure"], envmap=env, min_roughness=0.1, max_metallic=1.0) for c in cams]
        return sum((i * u).sum() for i, u in zip(imgs, ups))

    errors = []
    for rep in range(3):
        gA, lvA = _leaves(sc, cuda)
        gB, lvB = _leaves(sc, cuda)
        lossA = forward(lvA, gA)
        ev = torch.cuda.Event(); ev.record()

        def work():
            try:
                with torch.cuda.device(cuda):
                    torch.cuda.current_stream().wait_event(ev)
                    lossA.backward()
                    torch.cuda.current_stream().synchronize()
            except BaseException as e:                                 # noqa: BLE001
                errors.append(e)
        th = threading.Thread(target=work)
        th.start()
        lossB = forward(lvB, gB)                                       # step k + 1's forward while step k's backward runs
        th.join()
        lossB.backward()
        torch.cuda.synchronize()
        assert not errors, errors
        print(f"\n rep {rep}: threaded step A / following step B vs serial")
        _compare({k: v.grad for k, v in lvA.items()}, want)
        _compare({k: v.grad for k, v in lvB.items()}, want)
    vb.reset()
