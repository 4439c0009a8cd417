#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the IMPORTABLE pure-PyTorch glue of the reference (run in the build
container only; /root/reference does not exist on the GPU box).  Recipe from SURVEY.md section 8c: a permissive
`jaxtyping` stub + MagicMock entries for the absent third-party packages, then import rfstudio.

    cd /tmp && PYTHONPATH=/tmp/stubs:/root/reference python /root/repo/scripts/make_golden.py

Only INPUT/OUTPUT VECTORS are written (no reference source text).  The texture fetches inside
RenderableAttrs.splat (nvdiffrast, absent) are served by this repo's oracle, so that what the fixture pins is the
reference's own S1 arithmetic, mip-level map, camera matrices, tone mapping, atlas packing and MGAdapter.
"""
import math
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

for name in ["open3d", "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
             "cv2", "pyexr", "trimesh", "nvdiffrast", "nvdiffrast.torch", "kornia", "kornia.filters", "gsplat",
             "torchmetrics", "torchmetrics.functional", "torchmetrics.functional.image", "ffmpegcv", "nerfacc", "tyro",
             "skimage", "skimage.measure", "rfviser", "viser", "appdirs", "huggingface_hub", "pytorch3d",
             "pytorch3d.loss", "pytorch3d.structures", "rfstudio.graphics._mesh._optix", "rfstudio.graphics._mesh._splitsum",
             "tinycudann", "plotext", "imageio", "lpips", "matplotlib", "matplotlib.pyplot", "viser.transforms",
             "rfviser.transforms", "torchmetrics.image", "torchmetrics.image.lpip"]:
    sys.modules.setdefault(name, MagicMock())

import rfstudio.graphics as G                                           # noqa: E402
from rfstudio.graphics import Cameras                                    # noqa: E402
from rfstudio.graphics.math import quat2rot, rot2quat, safe_normalize    # noqa: E402
import rfstudio.model.geosplat as GEO                                    # noqa: E402
import rfstudio.graphics._mesh._texture as TEX                           # noqa: E402

import oracle                                                            # noqa: E402
import geosplatting_amd.synthetic as syn                                 # noqa: E402

os.makedirs(OUT, exist_ok=True)
g = torch.Generator().manual_seed(1234)

# ---------------------------------------------------------------- cameras (G2)
cams = Cameras.from_orbit(center=(0., 0., 0.), up=(0., 1., 0.), radius=3.0, pitch_degree=30.0, num_samples=4,
                          resolution=(256, 256), hfov_degree=40.0)
# NB: 2 cameras, not 3 -- the reference calls torch.cross without dim=, which picks the FIRST size-3 dimension
eye = torch.tensor([[1.0, 0.5, 2.0], [-2.0, 1.0, 0.3]])
look = Cameras.from_lookat(eye=eye, target=torch.zeros(2, 3), up=torch.tensor([[0., 1., 0.]]).repeat(2, 1),
                           resolution=(800, 800), hfov_degree=2 * math.degrees(math.atan(400 / 1111.111)))
np.savez_compressed(os.path.join(OUT, "ref_cameras.npz"),
         orbit_c2w=cams.c2w.numpy(), orbit_view=cams.view_matrix.numpy(), orbit_K=cams.intrinsic_matrix.numpy(),
         orbit_fx=cams.fx.numpy(), lookat_eye=eye.numpy(), lookat_c2w=look.c2w.numpy(), lookat_view=look.view_matrix.numpy(),
         lookat_K=look.intrinsic_matrix.numpy())

# ---------------------------------------------------------------- math
v = torch.randn(64, 3, generator=g); v[:4] = 0.0; v[4] = 1e-7
rots = quat2rot(torch.randn(32, 4, generator=g))
np.savez_compressed(os.path.join(OUT, "ref_math.npz"), v=v.numpy(), safe_normalize=safe_normalize(v).numpy(),
         rots=rots.numpy(), rot2quat=rot2quat(rots).numpy())

# ---------------------------------------------------------------- tone mapping (S4)
rgba = torch.rand(8, 8, 4, generator=g) * 1.7
exposure = torch.tensor(1.3)
np.savez_compressed(os.path.join(OUT, "ref_tonemap.npz"), rgba=rgba.numpy(), exposure=1.3,
         naive=GEO._tone_mapping_naive(rgba, exposure).numpy(), aces=GEO._tone_mapping_aces(rgba, exposure).numpy())

# ---------------------------------------------------------------- atlas packing + mip chain (S5 python side)
levels = [torch.rand(6, r, r, 3, generator=g) for r in (32, 16, 8)]
atlas = TEX._merge_mipmaps(levels)
atlas[..., 3, 24:, 24:] = 0          # the unused corner is torch.empty in the reference
split = TEX._split_mipmaps(atlas, num_mipmaps=3)
cube = torch.rand(6, 8, 8, 3, generator=g)
np.savez_compressed(os.path.join(OUT, "ref_atlas.npz"), l0=levels[0].numpy(), l1=levels[1].numpy(), l2=levels[2].numpy(),
         atlas=atlas.numpy(), s0=split[0].numpy(), s1=split[1].numpy(), s2=split[2].numpy(),
         cube=cube.numpy(), cube_mip=TEX._CubeMapMip.forward(None, cube).numpy())

# ---------------------------------------------------------------- MGAdapter (8f rank 1)
verts, faces = syn.icosphere(1, 0.8)
from rfstudio.graphics import TriangleMesh                              # noqa: E402
mesh = TriangleMesh(vertices=verts, indices=faces).compute_vertex_normals()
splats, offsets = GEO.MGAdapter().make(mesh)
np.savez_compressed(os.path.join(OUT, "ref_mgadapter.npz"), vertices=verts.numpy(), faces=faces.numpy(),
         vnormals=mesh.normals.numpy(), means=splats.means.numpy(), scales=splats.scales.numpy(), quats=splats.quats.numpy(),
         opacities=splats.opacities.numpy(), colors=splats.colors.numpy())

# ---------------------------------------------------------------- S1 arithmetic + roughness->mip map through the real splat()
# dr.texture is replaced by this repo's oracle fetches; GSplatter.render_rgba is intercepted to capture the colours.
ref_lut = np.fromfile("/root/reference/rfstudio/assets/geometry/pbr/bsdf_256_256.bin", dtype=np.float32).reshape(256, 256, 2)
# the S1 pin uses THIS repo's LUT (the reference asset is not shipped); which table is sampled does not matter for the arithmetic
lut = np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "geosplatting_amd", "assets", "fg_lut_256.bin"), dtype=np.float32).reshape(256, 256, 2)
N = 512
from oracle import mesh_ref                                              # noqa: E402
sc = syn.sphere_scene(1, seed=3, cubemap_res=16, mesh_to_splats_fn=mesh_ref.scene_builder)
idx = torch.randperm(sc.splats.num, generator=g)[:N]
means, normals, kd, ks = sc.splats.means[idx], sc.normals[idx], sc.kd[idx], sc.ks[idx]
N = means.shape[0]
ks[:8, 0] = 1.0; ks[8:16, 0] = 0.0; ks[16:24, 0] = (0.5 - 0.1) / 0.9
base = torch.rand(6, 16, 16, 3, generator=g)
pyr = [torch.rand(6, r, r, 3, generator=g) for r in (64, 32, 16, 8, 4, 2)]
captured = {}


def fake_texture(tex, uv, *args, mip=None, mip_level_bias=None, filter_mode=None, boundary_mode=None, **kw):
    if boundary_mode == "clamp":                                        # FG LUT (geosplat.py:93-98)
        out, _, _ = oracle.tex2d_linear_clamp(tex[0].numpy(), uv.reshape(-1, 2).numpy())
        return torch.from_numpy(out).view(*uv.shape[:-1], 2)
    d = uv.reshape(-1, 3).numpy()
    if mip is None:                                                     # diffuse base lookup
        out, _ = oracle.cube_linear(tex[0].numpy(), d)
    else:
        captured["mip_level_bias"] = mip_level_bias.reshape(-1).numpy().copy()
        lv = [tex[0].numpy()] + [m[0].numpy() for m in mip]
        out, _, _ = oracle.cube_mip_linear(lv, d, mip_level_bias.reshape(-1).numpy())
    return torch.from_numpy(out).view(*uv.shape[:-1], 3)


GEO.dr.texture = fake_texture
TEX.dr.texture = fake_texture
GEO._get_fg_lut = lambda resolution, device: torch.from_numpy(lut).view(1, 256, 256, 2)
env = TEX.TextureSplitSum(base=base, mipmaps=TEX._merge_mipmaps(pyr), num_mipmaps=torch.tensor([6]),
                          min_roughness=torch.tensor([0.08]), max_roughness=torch.tensor([0.5]), transform=None)


class FakeSplatter:
    def __init__(self, gaussians): self.gaussians = gaussians
    def render_rgba(self, cameras):
        captured["colors"] = self.gaussians.colors.detach().numpy().copy()
        m = MagicMock(); m.item.return_value = torch.zeros(4, 4, 4); return m


from rfstudio.graphics import Splats                                    # noqa: E402
gaussians = Splats(means=means, scales=torch.zeros(N, 3), quats=torch.randn(N, 4, generator=g), colors=torch.zeros(N, 3),
                   opacities=torch.zeros(N, 1), shs=torch.zeros(N, 0, 3))
cam1 = cams[1:2]
out = {}
for mode in ("pbr", "diffuse", "specular"):
    attrs = GEO.RenderableAttrs(kd=kd, ks=ks, occ=None, normals=normals, kd_jitter=None, ks_jitter=None)
    attrs.splat(FakeSplatter(gaussians), cam1, exposure=torch.tensor(1.0), envmap=env, min_roughness=0.1,
                max_metallic=1.0, mode=mode, tone_type="naive")
    out["colors_" + mode] = captured["colors"]
np.savez_compressed(os.path.join(OUT, "ref_splat_arith.npz"), means=means.numpy(), normals=normals.numpy(), kd=kd.numpy(),
         ks=ks.numpy(), cam_pos=cam1.c2w[0, :, 3].numpy(), base=base.numpy(), 
         mip_level_bias=captured["mip_level_bias"], **{f"level{i}": p.numpy() for i, p in enumerate(pyr)}, **out)

# ---------------------------------------------------------------- FG LUT sub-sample (reference DATA, 2 KB)
np.savez_compressed(os.path.join(OUT, "ref_fg_lut_sub16.npz"), rows=np.arange(8, 256, 16), cols=np.arange(8, 256, 16),
         values=ref_lut[8::16, 8::16].copy())
print("golden written to", os.path.normpath(OUT))
for f in sorted(os.listdir(OUT)):
    print(f, os.path.getsize(os.path.join(OUT, f)))
