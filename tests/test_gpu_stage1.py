"""One stage-1 iteration of the reference's geosplat loop, end to end on the HIP kernels, in the order
`GeoSplatter.render_report` runs it (rfstudio/model/geosplat.py:856-927): get_geometry (FlexiCubes) -> vertex normals ->
MGAdapter -> hash-grid field -> split-sum prefilter -> shade + rasterize + tone-map -> trainer loss -> backward into the
SDF / deformation / FlexiCubes weights / field / cubemap / exposure, stepped by Adam for a few iterations."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stage1_iterations_reduce_the_loss():
    import geosplatting_amd as gs
    from geosplatting_amd import synthetic as syn
    from geosplatting_amd.field import GaussianField
    from geosplatting_amd.loss import photo_loss
    dev = torch.device("cuda")
    torch.manual_seed(3)
    R, scale, HW = 32, 1.0, 160
    grid = gs.FlexiCubes.from_resolution(R, device=dev, random_sdf=False, scale=scale)
    cams = syn.blender_cameras(4, HW, HW)
    env_gt = gs.as_splitsum(syn.make_cubemap(64).to(dev))
    # target: an ellipsoid with a fixed albedo, rendered through the same path
    with torch.no_grad():
        sdf_gt = (grid.vertices * torch.tensor([1.0, 1.25, 0.85], device=dev)).norm(dim=-1, keepdim=True) - 0.55
        (vg, fg), _ = grid.replace(sdf_values=sdf_gt).dual_marching_cubes()
        sp_gt, n_gt = gs.mesh_to_splats(vg, fg, gs.vertex_normals(vg, fg))
        N = sp_gt.means.shape[0]
        attrs_gt = gs.RenderableAttrs(kd=torch.tensor([0.8, 0.3, 0.2], device=dev).expand(N, 3).contiguous(),
                                      ks=torch.tensor([0.4, 0.1], device=dev).expand(N, 2).contiguous(), normals=n_gt)
        gts = [attrs_gt.splat(sp_gt, [c], exposure=torch.tensor(1.0, device=dev), envmap=env_gt, min_roughness=0.1,
                              max_metallic=1.0).reshape(HW, HW, 4) for c in cams]
    # model: a sphere SDF, zero deformation / weights, fresh field, grey cubemap
    sdf = (grid.vertices.norm(dim=-1, keepdim=True) - 0.5).clone().requires_grad_(True)
    deform = torch.zeros_like(grid.vertices).requires_grad_(True)
    weights = torch.zeros(R ** 3, 21, device=dev).requires_grad_(True)
    cubemap = torch.full((6, 64, 64, 3), 0.5, device=dev).requires_grad_(True)
    log_exposure = torch.zeros(1, device=dev).requires_grad_(True)
    field = GaussianField(device=dev, log2_hashmap_size=15, seed=1)
    guess = torch.tensor([0.0, -1.0], device=dev)
    opt = torch.optim.Adam([dict(params=[sdf], lr=3e-3), dict(params=[deform, weights], lr=1e-2),
                            dict(params=field.parameters(), lr=1e-2), dict(params=[cubemap, log_exposure], lr=1e-2)])
    losses, faces_seen = [], set()
    for it in range(12):
        opt.zero_grad(set_to_none=True)
        (v, f), reg = gs.get_geometry(grid, deform, sdf, weights, scale=scale, resolution=R, sdf_weight=0.1)
        faces_seen.add(f.shape[0])
        splats, attrs, _ = field.get_gaussians_from_face(v, f, 0.0, 0.0, scale=scale, initial_guess=guess)
        envmap = gs.as_splitsum(cubemap)
        total = reg
        for cam, gt in zip(cams, gts):
            img = attrs.splat(splats, [cam], exposure=log_exposure.exp()[0], envmap=envmap, min_roughness=0.1,
                              max_metallic=1.0).reshape(HW, HW, 4)
            loss, _ = photo_loss(img[..., :3], img[..., 3:], gt, torch.rand(HW, HW, 3, device=dev), gt_is_srgb=False)
            total = total + loss / len(cams)
        total.backward()
        for t in [sdf, deform, weights, cubemap, log_exposure] + field.parameters():
            assert t.grad is not None and torch.isfinite(t.grad).all()
        assert sdf.grad.abs().max() > 0 and deform.grad.abs().max() > 0 and weights.grad.abs().max() > 0
        opt.step()
        losses.append(float(total.detach()))
    assert losses[-1] < 0.8 * losses[0], losses
    assert len(faces_seen) > 1                              # the topology (face count) changed while optimising
    print("\nstage-1 losses:", " ".join(f"{l:.4f}" for l in losses))
