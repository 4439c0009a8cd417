"""Diagnostic for tests/test_gpu_stage1.py::test_scheduled_run_starts_with_vertex_sampling: the fused-vs-autograd gradient differences
per iteration (vertex sampling 0-1, face sampling 2-3) and parameter, max-norm relative."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tests.test_gpu_stage1 import _scene, _model, HW, N_VIEWS
from geosplatting_amd.stage1 import GeoSplatSchedule, train_step, train_step_fused
dev = torch.device("cuda")
for seed in (3, 4, 5):
    torch.manual_seed(seed)
    cams, gts, grid = _scene(dev)
    model = _model(dev, grid)
    sch = GeoSplatSchedule(vertex_sample_warmup=2)
    opt = torch.optim.Adam(model.parameters(), lr=3e-3)
    g_b = torch.Generator().manual_seed(4)
    bgs = [torch.rand(HW, HW, 3, generator=g_b).to(dev) for _ in range(N_VIEWS)]
    for it in range(4):
        sch.before_update(model, it)
        model.kd_regualr_perturb_std = model.ks_regualr_perturb_std = 0.0
        model.get_gsplat()
        ref = {}
        rng = model._jitter_gen.get_state().clone()
        for name, fn in (("autograd", train_step), ("fused", train_step_fused), ("autograd2", train_step)):
            model._jitter_gen.set_state(rng)
            fn(model, cams, gts, gt_is_srgb=False, train_bg=bgs)
            ref[name] = {k: p.grad.detach().clone() for k, p in model.named_parameters().items()}
        errs = {k: (ref["fused"][k] - w).abs().max().item() / (w.abs().max().item() + 1e-30) for k, w in ref["autograd"].items()}
        rep = {k: (ref["autograd2"][k] - w).abs().max().item() / (w.abs().max().item() + 1e-30) for k, w in ref["autograd"].items()}
        worst = max(errs, key=errs.get)
        print(f"seed {seed} it {it} ({model.sample_method}, N={model.last_num_gaussians}): fused vs autograd max {errs[worst]:.2e} at {worst}; "
              f"autograd run-to-run max {max(rep.values()):.2e};  " + " ".join(f"{k.split('.')[-1]}={v:.1e}" for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:4]))
        sch.scale_light_gradient(model); opt.step(); sch.after_update(model, it)
