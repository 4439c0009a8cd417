"""Diagnostics: per-tile list lengths and per-wave walk depth at the bench workload."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
level = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device('cuda:0')
sc = syn.sphere_scene(level, seed=1, cubemap_res=64)
cam = syn.blender_cameras(8)[0]
sp = sc.splats.to(dev)
colors = torch.rand(sp.num, 3, device=dev)
r, a, meta = gs.rasterization(sp.means, sp.quats, sp.scales.exp(), torch.sigmoid(sp.opacities).squeeze(-1), colors,
                              cam.view_matrix.to(dev)[None], cam.intrinsic_matrix.to(dev)[None], 800, 800)
off = meta['isect_offsets'].reshape(-1).long().cpu().numpy()
I = meta['flatten_ids'].numel()
cnt = np.diff(np.concatenate([off, [I]]))
print('tiles', len(cnt), 'I', I, 'mean', cnt.mean(), 'max', cnt.max(), 'p50/p90/p99', np.percentile(cnt, [50, 90, 99]))
print('nonempty tiles', (cnt > 0).sum(), 'tiles >4096:', (cnt > 4096).sum(), '>8192:', (cnt > 8192).sum(), '>16384', (cnt>16384).sum())
last = meta['last_ids'][0].cpu().numpy().astype(np.int64)
alpha = a[0, ..., 0].cpu().numpy()
H = W = 800
# per 8x8 quadrant (wave): walked depth
lq = last.reshape(100, 8, 100, 8).transpose(0, 2, 1, 3).reshape(100, 100, 64)
aq = alpha.reshape(100, 8, 100, 8).transpose(0, 2, 1, 3).reshape(100, 100, 64)
tile_of_q = (np.arange(100)[:, None] // 2) * 50 + (np.arange(100)[None, :] // 2)
start = off[tile_of_q]; n = cnt[tile_of_q]
unsat = (aq < 0.99).any(-1)       # some pixel never terminated -> wave walks the whole list
depth = np.where(unsat, n, np.minimum(n, lq.max(-1) - start + 64))
batches = np.ceil(depth / 64)
print('waves', batches.size, 'mean batches', batches.mean(), 'max', batches.max(), 'sum', batches.sum(),
      'frac waves walking whole list', unsat.mean(), 'sum batches if whole', np.ceil(n / 64).sum())
tb = batches.reshape(50, 2, 50, 2).max(axis=(1, 3))
print('per-tile max batches: mean', tb.mean(), 'p99', np.percentile(tb, 99), 'max', tb.max())
tpg = meta['tiles_per_gauss'].cpu().numpy(); print('tiles per gauss mean', tpg.mean(), 'max', tpg.max(), 'radii mean', meta['radii'].float().mean().item(), 'max', meta['radii'].max().item())
