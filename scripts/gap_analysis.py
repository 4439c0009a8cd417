"""Aggregate GPU idle gaps between consecutive kernels over the LAST full step of a rocprofv3 (rocpd sqlite) trace of
bench.py: a step = from its first specular_apply (prefilter forward) to its last one (prefilter backward)."""
import sqlite3, sys, collections
db = sys.argv[1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kt} d join {ks} s on d.kernel_id=s.id order by d.start"))
short = lambda n: n.split('(')[0].replace('void ', '')[:36]
ap = [i for i, r in enumerate(rows) if 'specular_apply' in r[0]]
steps = [(ap[k], ap[k + 11]) for k in range(0, len(ap) - 11, 12)]
lo, hi = steps[-1]
seg = rows[lo:hi + 1]
wall = seg[-1][2] - seg[0][1]; busy = sum(r[2] - r[1] for r in seg)
nb = sum(1 for r in seg if 'raster_bwd' in r[0])
print(f"last step: {len(seg)} kernels, {nb} views, wall {wall / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(wall - busy) / 1e6:.2f} ms")
agg = collections.Counter(); cnt = collections.Counter()
for a, b in zip(seg[:-1], seg[1:]):
    g = b[1] - a[2]
    if g > 0:
        agg[(short(a[0]), short(b[0]))] += g; cnt[(short(a[0]), short(b[0]))] += 1
for k, v in agg.most_common(24):
    print(f"{v / 1e3:9.1f} us  n={cnt[k]:3d}  {k[0]} -> {k[1]}")
