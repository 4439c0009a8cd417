"""Timing experiment: cached-weights prefilter apply per pyramid level, variants via -D flags (diagnostic builds in /tmp)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import geosplatting_amd.build as B
tag = sys.argv[1] if len(sys.argv) > 1 else "base"
defs = sys.argv[2:]
so = f"/tmp/libgeosplat_ap_{tag}.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, *defs, "-shared", "-o", so, *[os.path.join(B.CSRC, s) for s in B.SOURCES]])
import geosplatting_amd._lib as L
L.LIB_PATH = so
import geosplatting_amd.splitsum as ss, geosplatting_amd.synthetic as syn
dev = torch.device("cuda:0")
cube = syn.make_cubemap(512).to(dev)
levels = [cube]
while levels[-1].shape[1] > 16:
    levels.append(ss._CubeMapMip.apply(levels[-1]))
tot_ms = 0.0
for i, lv in enumerate(levels):
    res = lv.shape[1]
    rough = (i / (len(levels) - 1)) * (0.5 - 0.08) + 0.08
    e = ss.specular_weights(res, rough, 0.99, dev)
    for rep in range(3):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); out = ss._SpecularCubemapCached.apply(lv, res, rough, 0.99); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1); tot_ms += ms
    print(f"{tag} R={res} patches={e['total']} weights={e['total'] * 256 / 1e9:.2f} GB  fwd {ms:.3f} ms  -> {e['total'] * 256 / ms / 1e9:.2f} TB/s")
print(tag, "total fwd ms", tot_ms)
