"""One view per step through engine.RenderStep (pyramid fixed, so nothing of another view overlaps a kernel): run under
`rocprofv3 --kernel-trace --stats` to get every kernel of the ENGINE's per-view chain alone.
python scripts/view_kernels_engine.py [level=7] [reps=8]        (GEOSPLAT_FRONT=split: the round-3 launch sequence)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import geosplatting_amd.synthetic as syn
from geosplatting_amd.engine import RenderStep, params_from_scene

dev = torch.device("cuda", 0)
level = int(sys.argv[1]) if len(sys.argv) > 1 else 7
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
scene = syn.sphere_scene(level, seed=1, cubemap_res=512, device=dev)
cam = syn.blender_cameras(8, 800, 800)[1]
step = RenderStep(params_from_scene(scene, dev), prefilter=False)
up = (torch.rand(800, 800, 4, device=dev) * 2 - 1)
step([cam], lambda i, img: up, all_reduce=False); torch.cuda.synchronize()
step.poll_capacity(wait=True)
for _ in range(reps):
    step([cam], lambda i, img: up, all_reduce=False)
    torch.cuda.synchronize()
print("i_cap", step._i_cap, "key32", step._key32, "key_lo/hi", step._key_lo, step._key_hi)
