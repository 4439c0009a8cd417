"""Isolate the stages of the full-size comparison: (1) rasterizer backward on the ORACLE's colours, (2) shading backward on the
ORACLE's colour gradients -- which stage carries the worst element?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
import geosplatting_amd as gs
from tests.util import activated, sphere_case
level = int(sys.argv[1]) if len(sys.argv) > 1 else 7
cuda = torch.device("cuda:0")
sc, cam = sphere_case(level, 800, view=1, cubemap_res=512)
with torch.no_grad():
    env0 = gs.as_splitsum(sc.cubemap.to(cuda))
base = env0.base.cpu(); levels = [l.cpu() for l in env0.levels]
lut = gs.get_fg_lut(torch.device("cpu"))[0].numpy()
means, quats, scales, opac = activated(sc.splats)
cam_pos = cam.c2w[:, 3].numpy(); lv = [l.numpy() for l in levels]
vm, K = cam.view_matrix.numpy(), cam.intrinsic_matrix.numpy()
W = H = 800
col = oracle.shade_fwd(means, sc.normals.numpy(), sc.kd.numpy(), sc.ks.numpy(), cam_pos, lut, base.numpy(), lv)
m = oracle.rasterization(means, quats, scales, opac, col, vm, K, W, H)
g = torch.Generator().manual_seed(3)
v = torch.rand(H, W, 4, generator=g) * 2 - 1
v[torch.tensor(m["ambiguous"])] = 0
gr = oracle.rasterization_bwd(means, quats, scales, opac, col, vm, K, W, H, m, v[..., :3].numpy(), v[..., 3].numpy())
t = lambda a: torch.tensor(a, device=cuda, requires_grad=True)
tm, tq, ts, to, tc = t(means), t(quats), t(scales), t(opac), t(col)
r, a, meta = gs.rasterization(tm, tq, ts, to, tc, torch.tensor(vm, device=cuda)[None], torch.tensor(K, device=cuda)[None], W, H)
print("image max abs diff", float(np.abs(r[0].detach().cpu().numpy() - m["render"]).max()), " last_ids equal:",
      bool(np.array_equal(meta["last_ids"][0].cpu().numpy(), m["last_ids"])), " alpha bit-equal:",
      bool(np.array_equal(a[0, ..., 0].detach().cpu().numpy(), m["alphas"])))
(r[0] * v[..., :3].to(cuda)).sum().add((a[0, ..., 0] * v[..., 3].to(cuda)).sum()).backward()
def rep(name, got, want):
    got = got.astype(np.float64); want = want.astype(np.float64)
    d = np.abs(got - want); w = int(d.argmax())
    print(f"  {name:10s} max-norm {d.max() / np.abs(want).max():.3e}  worst element {w}: {got.reshape(-1)[w]:.7e} vs {want.reshape(-1)[w]:.7e}")
    return w
print("rasterizer backward on identical colours:")
wc = rep("v_colors", tc.grad.cpu().numpy(), gr["v_colors"])
rep("v_means", tm.grad.cpu().numpy(), gr["v_means"]); rep("v_opac", to.grad.cpu().numpy(), gr["v_opacities"])
gi = wc // 3
print("  worst Gaussian", gi, "radius", int(m["radii"][np.searchsorted(m["gaussian_ids"], gi)]), "means2d", m["means2d"][np.searchsorted(m["gaussian_ids"], gi)],
      "opacity", m["opacities"][np.searchsorted(m["gaussian_ids"], gi)], "conic", m["conics"][np.searchsorted(m["gaussian_ids"], gi)])
print("shading backward on identical colour gradients:")
gsh = oracle.shade_bwd(means, sc.normals.numpy(), sc.kd.numpy(), sc.ks.numpy(), cam_pos, lut, base.numpy(), lv, gr["v_colors"])
d = lambda x: x.clone().to(cuda).requires_grad_(True)
tmm, tn, tkd, tks = d(sc.splats.means), d(sc.normals), d(sc.kd), d(sc.ks)
env = gs.TextureSplitSum(base.to(cuda), [l.to(cuda) for l in levels])
c2 = gs.shade(tmm, tn, tkd, tks, cam.c2w[:, 3].to(cuda).contiguous(), env, min_roughness=0.1, max_metallic=1.0)
c2.backward(torch.tensor(gr["v_colors"], device=cuda))
rep("colors", c2.detach().cpu().numpy(), col)
rep("v_kd", tkd.grad.cpu().numpy(), gsh["v_kd"]); rep("v_ks", tks.grad.cpu().numpy(), gsh["v_ks"]); rep("v_normals", tn.grad.cpu().numpy(), gsh["v_normals"])
