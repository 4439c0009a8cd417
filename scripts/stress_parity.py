"""Stress run (not part of the test suite): many seeded random scenes / cameras / resolutions through the rasterizer parity
check of tests/test_gpu_rasterizer.py, to look for rare mismatches."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.util import activated, random_case, sphere_case
from tests.test_gpu_rasterizer import _run_case
cuda = torch.device("cuda:0")
fails = 0
cases = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    g = torch.Generator().manual_seed(100 + seed)
    try:
        if seed % 2 == 0:
            res = [64, 96, 128, 200][seed // 2 % 4]
            sp, cam = random_case([500, 2000, 5000][seed % 3], res, view=seed % 4, seed=seed + 7)
            # anisotropic + varied opacity
            sp.scales = sp.scales + torch.randn(sp.scales.shape, generator=g) * 0.5
            sp.opacities = torch.logit(torch.rand(sp.opacities.shape, generator=g) * 0.9 + 0.05)
            means, quats, scales, opac = activated(sp)
            colors = torch.rand(sp.num, 3, generator=g).numpy()
        else:
            level = 2 + seed % 3
            sc, cam = sphere_case(level, [80, 128, 176][seed % 3], view=seed % 8, seed=seed)
            means, quats, scales, opac = activated(sc.splats)
            colors = torch.rand(sc.splats.num, 3, generator=g).numpy()
        bg = None if seed % 3 else np.array([0.2, 0.5, 0.9], np.float32)
        _run_case(cuda, means, quats, scales, opac, colors, cam, background=bg)
        cases += 1
    except AssertionError as e:
        fails += 1
        print("FAIL seed", seed, str(e)[:200])
    except Exception:
        fails += 1
        print("ERROR seed", seed); traceback.print_exc()
print(f"stress: {cases} passed, {fails} failed")
