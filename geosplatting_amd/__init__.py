"""geosplatting_amd -- MI355X (gfx950) implementation of the GeoSplatting render-and-backward hot path.

Public surface (mirrors the reference's two call boundaries):
  rasterization(...)                      == gsplat.rasterization as called at rfstudio/model/gsplat.py:334-355
  RenderableAttrs(kd, ks, normals).splat  == rfstudio/model/geosplat.py:53-132  (one autograd node per view on the fused kernels:
                                             viewbatch.py; GEOSPLAT_SPLAT=ops = shade -> rasterization -> tone_map op by op)
  as_splitsum(cubemap) / TextureSplitSum  == rfstudio/graphics/_mesh/_texture.py:530-613
  render_rgba, shade, tone_map            == the pieces in between
  mesh_to_splats(vertices, faces, vn)     == MGAdapter.make, rfstudio/model/geosplat.py:426-472 (section 8f rank 1)
  photo_loss(rgb, alpha, gt, bg)          == per-view trainer loss, rfstudio/trainer/geosplat_trainer.py:171-195 (rank 2)
  HashEncoding / hash_encode              == HashEncoding backend='torch', rfstudio/model/components/encoding.py:87-241 (rank 3)
  FlexiCubes / get_geometry               == FlexiCubes.dual_marching_cubes / compute_entropy, rfstudio/graphics/_mesh/_flexicubes.py:369-802,
                                             GeoSplatter.get_geometry rfstudio/model/geosplat.py:751-769 (rank 4)
  stage1.Stage1Model / stage1.train_step  == GeoSplatter (stage 1) + GeoSplatTrainer.step, rfstudio/model/geosplat.py:676-927,
                                             rfstudio/trainer/geosplat_trainer.py:150-186 (BASELINE config 5)
Every op calls hand-written HIP kernels in libgeosplat_hip.so through the C-ABI of include/geosplat_hip.h and
raises if that library is missing -- there is no CPU or PyTorch fallback path.
"""
from .cameras import Camera, intrinsic_matrix, lookat_c2w, orbit_cameras, view_matrix  # noqa: F401
from .field import HashEncoding, hash_encode  # noqa: F401
from .flexicubes import FlexiCubes, get_geometry  # noqa: F401
from .loss import photo_loss  # noqa: F401
from .mesh import mesh_to_splats, vertex_normals  # noqa: F401
from .rasterization import rasterization  # noqa: F401
from .shading import RenderableAttrs, get_fg_lut, render_depth, render_rgb, render_rgba, shade, tone_map  # noqa: F401
from . import viewbatch  # noqa: F401
from .splitsum import TextureSplitSum, as_splitsum, diffuse_cubemap, specular_cubemap  # noqa: F401

__version__ = "0.1.0"
