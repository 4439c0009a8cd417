"""S5 alone on the GPU box: table build time and size, forward / backward time per level and for the whole pyramid.
    python scripts/prefilter_bench.py [cubemap_res=512]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import geosplatting_amd._lib as _L
if os.environ.get("GEOSPLAT_LIB"):
    _L.LIB_PATH = os.path.abspath(os.environ["GEOSPLAT_LIB"])
import geosplatting_amd as gs
import geosplatting_amd.synthetic as syn
from geosplatting_amd import splitsum as ss

dev = torch.device("cuda", 0)
R0 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cube = syn.make_cubemap(R0, seed=1).to(dev)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t0 = time.time()
with torch.no_grad():
    env = gs.as_splitsum(cube)
torch.cuda.synchronize()
print(f"first as_splitsum (bounds + tables): {time.time() - t0:.2f} s")
n = len(env.levels)
roughs = ss._level_roughness(n, 0.08, 0.5)
tot_bytes = 0
for lvl, rough in zip(env.levels, roughs):
    R = lvl.shape[1]
    e = ss.specular_tiles(R, rough, 0.99, dev)
    src = torch.rand(6, R, R, 3, device=dev) + 0.1
    if e is None:
        f = timed(lambda: ss.specular_cubemap(src, rough))
        print(f"  R={R:4d} rough={rough:.3f}: direct kernels, fwd {f * 1e3:.1f} us")
        continue
    out = torch.empty_like(src)
    f = timed(lambda: ss._tiles_apply(e, "fwd", src, out))
    b = timed(lambda: ss._tiles_apply(e, "bwd", src, out))
    by = sum(e[d]["weights"].numel() * 4 + e[d]["desc"].numel() * 4 for d in ("fwd", "bwd"))
    tot_bytes += by
    dens = e["fwd"]["pairs"] / (64.0 * max(e["fwd"]["kept_rows"], 1))
    rows = e["fwd"]["rows"] * e["n_mirrors"]
    print(f"  R={R:4d} rough={rough:.3f}: mirrors {e['n_mirrors']} check {e['symmetry_check']} tiles {e['n_tiles']} rows/dir {e['fwd']['rows']} "
          f"density {dens:.2f} tables {by / 2**20:.0f} MiB lds {e['fwd']['lds_bytes']}/{e['bwd']['lds_bytes']} | fwd {f * 1e3:.1f} us bwd {b * 1e3:.1f} us "
          f"({rows * 64 / f / 1e6:.0f} / {rows * 64 / b / 1e6:.0f} G slot-pairs/s)")
print(f"tables total {tot_bytes / 2**30:.2f} GiB")
x = cube.clone().requires_grad_(True)
gb = torch.rand(6, 16, 16, 3, device=dev); gl = [torch.rand_like(l) for l in env.levels]
with torch.no_grad():
    f = timed(lambda: gs.as_splitsum(cube))
    b = timed(lambda: ss.as_splitsum_backward(gb, gl))
print(f"as_splitsum forward {f:.3f} ms, explicit backward {b:.3f} ms  (round 2: 2.4 + 2.4 ms in the engine, 1.5 + 1.5 alone)")
