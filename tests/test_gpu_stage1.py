"""BASELINE config 5 -- the reference's whole stage-1 iteration on the HIP kernels, in `GeoSplatter.render_report` /
`GeoSplatTrainer.step` order (rfstudio/model/geosplat.py:856-927, rfstudio/trainer/geosplat_trainer.py:150-186):
get_geometry (FlexiCubes) -> vertex normals -> MGAdapter -> hash-grid field -> split-sum prefilter -> shade + rasterize +
tone-map -> per-view loss -> backward into SDF / deformation / FlexiCubes weights / field / cubemap / exposure; Adam steps
reduce the loss, and two data-parallel ranks produce the single-process gradient."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, HW, N_VIEWS = 32, 160, 4


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _scene(dev):
    """target images: an ellipsoid with a fixed albedo, rendered through the same path"""
    import geosplatting_amd as gs
    from geosplatting_amd import synthetic as syn
    cams = syn.blender_cameras(N_VIEWS, HW, HW)
    grid = gs.FlexiCubes.from_resolution(R, device=dev, random_sdf=False, scale=1.0)
    with torch.no_grad():
        sdf_gt = (grid.vertices * torch.tensor([1.0, 1.25, 0.85], device=dev)).norm(dim=-1, keepdim=True) - 0.55
        (vg, fg), _ = grid.replace(sdf_values=sdf_gt).dual_marching_cubes()
        sp, n = gs.mesh_to_splats(vg, fg, gs.vertex_normals(vg, fg))
        N = sp.means.shape[0]
        attrs = gs.RenderableAttrs(kd=torch.tensor([0.8, 0.3, 0.2], device=dev).expand(N, 3).contiguous(),
                                   ks=torch.tensor([0.4, 0.1], device=dev).expand(N, 2).contiguous(), normals=n)
        env = gs.as_splitsum(syn.make_cubemap(64).to(dev))
        gts = [attrs.splat(sp, [c], exposure=torch.tensor(1.0, device=dev), envmap=env, min_roughness=0.1,
                           max_metallic=1.0).reshape(HW, HW, 4) for c in cams]
    return cams, gts, grid


def _model(dev, grid):
    from geosplatting_amd.stage1 import Stage1Model
    m = Stage1Model(R, scale=1.0, light_resolution=64, device=dev, seed=1, log2_hashmap_size=15, initial_guess="outdoor",
                    sdf_init=grid.vertices.norm(dim=-1, keepdim=True) - 0.5)
    m.sdf_weight, m.light_weight = 0.1, 0.01
    m.kd_regualr_perturb_std = m.ks_regualr_perturb_std = 0.02; m.kd_grad_weight = m.ks_grad_weight = 0.05
    return m


def test_stage1_iterations_reduce_the_loss():
    from geosplatting_amd.stage1 import train_step
    dev = torch.device("cuda")
    torch.manual_seed(3)
    cams, gts, grid = _scene(dev)
    model = _model(dev, grid)
    p = model.named_parameters()
    opt = torch.optim.Adam([dict(params=[p["sdf_params"]], lr=3e-3),
                            dict(params=[v for k, v in p.items() if k != "sdf_params"], lr=1e-2)])
    losses, faces_seen = [], set()
    for it in range(12):
        m = train_step(model, cams, gts, gt_is_srgb=False)
        for k, t in p.items():
            assert t.grad is not None and torch.isfinite(t.grad).all(), k
        for k in ("sdf_params", "deform_params", "weight_params", "cubemap", "exposure_params"):
            assert p[k].grad.abs().max() > 0, k
        opt.step()
        losses.append(float(m["loss_local_views"] + m["regularization"]))
        faces_seen.add(int(m["#gaussians"]))
    assert losses[-1] < 0.8 * losses[0], losses
    assert len(faces_seen) > 1                              # the topology (face count) changed while optimising
    print("\nstage-1 losses:", " ".join(f"{l:.4f}" for l in losses))


def _grads(dev, rank, world, fused=False):
    from geosplatting_amd.stage1 import train_step, train_step_fused
    train_step = train_step_fused if fused else train_step
    torch.manual_seed(11)                                   # field jitter draws: the same on every rank
    cams, gts, grid = _scene(dev)
    model = _model(dev, grid)
    model.kd_regualr_perturb_std = model.ks_regualr_perturb_std = 0.0    # jitter is a per-rank random draw: off for the comparison
    bgs = [torch.rand(HW, HW, 3, generator=torch.Generator().manual_seed(70 + i)).to(dev) for i in range(N_VIEWS)]
    train_step(model, cams, gts, gt_is_srgb=False, rank=rank, world_size=world, train_bg=bgs)
    torch.cuda.synchronize()
    return {k: v.grad.detach().cpu().clone() for k, v in model.named_parameters().items()}


def _worker(rank, world, port, out_dir, fused=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      GEOSPLAT_DEBUG_SHARE_GPU="1")
    import torch.distributed as dist
    from geosplatting_amd.parallel import init_distributed_from_env
    r, w, dev = init_distributed_from_env("cuda")
    torch.save(_grads(dev, r, w, fused), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_stage1_two_ranks_equal_single_process(tmp_path):
    """views sharded over two ranks (gloo on one GPU), one flat gradient all-reduce: every parameter's gradient -- SDF,
    deformation, FlexiCubes weights, hash tables, MLPs, cubemap, exposure -- equals the single-process one"""
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(2)]
    want = _grads(torch.device("cuda", 0), 0, 1)
    for k, w in want.items():
        scale = w.abs().max().item() + 1e-30
        for r in range(2):
            err = (got[r][k] - w).abs().max().item() / scale
            assert err < 5e-4, (k, r, err)          # atomics / summation order across views
        assert torch.equal(got[0][k], got[1][k]), k


def test_stage1_fused_engine_step_equals_autograd_step(tmp_path):
    """the same trainer step with the render / loss half on engine.RenderStep (C-ABI drivers on three streams, HIP loss
    gradient, all-reduce at the per-Gaussian cut): one process against the autograd step, two ranks against one"""
    want = _grads(torch.device("cuda", 0), 0, 1)
    one = _grads(torch.device("cuda", 0), 0, 1, fused=True)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), True), nprocs=2, join=True)
    two = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(2)]
    for k, w in want.items():
        scale = w.abs().max().item() + 1e-30
        for name, got in (("fused", one), ("rank0", two[0]), ("rank1", two[1])):
            err = (got[k] - w).abs().max().item() / scale
            assert err < 5e-4, (k, name, err)


def test_smoothing_branches(cuda):
    """'grad' / 'tv' / normal_grad branches of render_report (rfstudio/model/geosplat.py:881-922): the regulariser equals the
    formulae evaluated on the un-shaded renders, and its gradient reaches the kd / ks encoders and the geometry."""
    import geosplatting_amd as gs
    from geosplatting_amd import synthetic as syn
    from geosplatting_amd.shading import render_rgb
    from geosplatting_amd.stage1 import Stage1Model, spatial_gradient
    cams = syn.blender_cameras(2, 96, 96)
    m = Stage1Model(16, light_resolution=64, device=cuda, log2_hashmap_size=12, sdf_init=None, seed=1)
    with torch.no_grad():
        m.sdf_params.copy_(m.grid.vertices.norm(dim=-1, keepdim=True) - 0.6)
    g = torch.Generator().manual_seed(4)
    gts = [torch.rand(96, 96, 4, generator=g).to(cuda) for _ in cams]
    # Sobel restatement: constant image -> 0, a ramp along x -> d/dx = slope, d/dy = 0 (interior)
    ramp = torch.arange(12.0, device=cuda)[None, :, None].expand(9, 12, 3) * 0.5
    sg = spatial_gradient(ramp.contiguous())
    assert sg.shape == (3, 2, 9, 12) and torch.allclose(sg[:, 0, :, 1:-1], torch.full_like(sg[:, 0, :, 1:-1], 0.5)) and sg[:, 1].abs().max() == 0
    for mode, kw, kn in (("tv", 2.0, 0.0), ("grad", 1.5, 0.7)):
        m.smooth_type, m.kd_grad_weight, m.ks_grad_weight, m.normal_grad_weight = mode, kw, 0.5 * kw, kn
        for p in m.parameters():
            p.grad = None
        # perturbation stds > 0 must be IGNORED outside smooth_type == 'jitter' (rfstudio/model/geosplat.py:800-801): no jittered
        # encoder evaluation, no |kd_jitter - kd| term, nothing drawn from the jitter generator
        m.kd_regualr_perturb_std = m.ks_regualr_perturb_std = 0.0
        _, _, _, reg0 = m.get_gsplat()
        m.kd_regualr_perturb_std = m.ks_regualr_perturb_std = 0.05
        rng_before = m._jitter_gen.get_state().clone()
        _, _, attrs_j, reg1 = m.get_gsplat()
        assert attrs_j.kd_jitter is None and attrs_j.ks_jitter is None
        assert float(reg1) == float(reg0) and torch.equal(m._jitter_gen.get_state(), rng_before)
        _, _, reg = m.render_report(cams, gts)
        smooth = m._last_smoothing
        assert float(smooth) > 0
        # independent evaluation on the same Gaussians
        _, splats, attrs, _ = m.get_gsplat()
        bg = m.background_color
        want = 0.0
        ks3 = torch.cat((torch.zeros_like(attrs.ks[..., :1]), attrs.ks), -1)
        rr = lambda colors, cam: render_rgb(splats.means, splats.scales, splats.quats, splats.opacities, colors, cam, bg)
        tv = lambda img: (img[1:] - img[:-1]).square().mean() + (img[:, 1:] - img[:, :-1]).square().mean()
        for cam, gt in zip(cams, gts):
            gt_rgb = gt[..., :3] * gt[..., 3:] + bg * (1 - gt[..., 3:])
            edge = lambda img: (spatial_gradient(img).abs() * (-spatial_gradient(gt_rgb).abs()).exp()).sum(1).mean()
            if mode == "tv":
                want = want + (tv(rr(attrs.kd, cam)) * m.kd_grad_weight + tv(rr(ks3, cam)) * m.ks_grad_weight) / len(cams)
            else:
                want = want + (edge(rr(attrs.kd, cam)) * m.kd_grad_weight + edge(rr(ks3, cam)) * m.ks_grad_weight
                               + edge(rr(attrs.normals * 0.5 + 0.5, cam)) * m.normal_grad_weight) / len(cams)
        assert abs(float(smooth) - float(want)) < 1e-5 * max(1.0, abs(float(want))), (mode, float(smooth), float(want))
        smooth.backward()
        assert float(m.field.kd_enc.hash_table.grad.abs().sum()) > 0 and float(m.field.ks_enc.hash_table.grad.abs().sum()) > 0
        assert float(m.sdf_params.grad.abs().sum()) > 0


def test_scheduled_run_starts_with_vertex_sampling():
    """The reference's schedule (GeoSplatTrainer.before_update / after_update, rfstudio/trainer/geosplat_trainer.py:209-266) on the
    HIP path: the first `vertex_sample_warmup` steps render ONE Gaussian per mesh vertex (GaussianField.get_gaussians_from_vertex,
    rfstudio/model/geosplat.py:559-620), then the MGAdapter's six per face; the fused step and the autograd step agree in both
    samplings, every parameter receives a finite gradient, the environment gradient is scaled by 64 and the cubemap floored."""
    from geosplatting_amd.stage1 import GeoSplatSchedule, train_step, train_step_fused
    dev = torch.device("cuda")
    torch.manual_seed(3)
    cams, gts, grid = _scene(dev)
    model = _model(dev, grid)
    sch = GeoSplatSchedule(vertex_sample_warmup=2)
    opt = torch.optim.Adam(model.parameters(), lr=3e-3)
    g_b = torch.Generator().manual_seed(4)
    bgs = [torch.rand(HW, HW, 3, generator=g_b).to(dev) for _ in range(N_VIEWS)]
    counts = []
    for it in range(4):
        sch.before_update(model, it)
        assert model.sample_method == ("vertex" if it < 2 else "face")
        assert abs(model.sdf_weight - (0.2 - 0.08 * it / 500)) < 1e-9 and model.kd_regualr_perturb_std == 0.01
        model.kd_regualr_perturb_std = model.ks_regualr_perturb_std = 0.0           # (the jitter is a random draw: off for the comparison)
        (v, f), splats, attrs, _ = model.get_gsplat()
        assert splats.means.shape[0] == (v.shape[0] if it < 2 else 6 * f.shape[0]) == model.last_num_gaussians
        assert torch.allclose(attrs.normals.norm(dim=-1), torch.ones_like(attrs.normals[:, 0]), atol=1e-5)
        ref = {}
        rng = model._jitter_gen.get_state().clone()          # a vertex normal exactly opposite to +z takes rotation_between's random restart
        for name, fn in (("autograd", train_step), ("fused", train_step_fused)):
            model._jitter_gen.set_state(rng)
            out = fn(model, cams, gts, gt_is_srgb=False, train_bg=bgs)
            ref[name] = {k: p.grad.detach().clone() for k, p in model.named_parameters().items()}
            assert int(out["#gaussians"]) == model.last_num_gaussians
        for k, w in ref["autograd"].items():
            assert torch.isfinite(w).all(), k
            err = (ref["fused"][k] - w).abs().max().item() / (w.abs().max().item() + 1e-30)
            # scripts/vertex_mode_errors.py (three seeds): fused vs autograd <= 1.3e-5 in both samplings -- but the AUTOGRAD step against
            # ITSELF, same inputs, reached 1.8e-3 once in vertex mode (2 300 large Gaussians: thousands of float atomics per gradient
            # record, whose order is not fixed) and 2.1e-5 in face mode.  The bound is that run-to-run spread, not a bias.
            assert err < (3e-3 if it < 2 else 1e-4), (it, k, err)
        counts.append(model.last_num_gaussians)
        before = model.cubemap.grad.clone()
        sch.scale_light_gradient(model)
        assert torch.equal(model.cubemap.grad, before * 64)
        opt.step()
        sch.after_update(model, it)
        assert float(model.cubemap.min()) >= 1e-2
    assert counts[2] > 3 * counts[0]                                                # six per face >> one per vertex
