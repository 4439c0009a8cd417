"""Every kernel between the LAST compositor backward of one engine step and the FIRST compositor forward of the next (rocprofv3
kernel trace of bench.py, rocpd sqlite): start (ms since that backward's end), duration, stream -- what the step's ends consist of."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kt})")]
key = 'stream_id' if 'stream_id' in cols else 'queue_id'
qcol = 'queue_id' if 'queue_id' in cols else key
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.{key}, d.{qcol} from {kt} d join {ks} s on d.kernel_id=s.id order by d.start"))
short = lambda n: n.split('(')[0].replace('void ', '')[:44]
comp = [i for i, r in enumerate(rows) if 'raster_fwd' in r[0] or 'raster_bwd' in r[0]]
# boundaries: a raster_bwd followed (as the next compositor kernel) by a raster_fwd more than 1.2 ms later
bounds = [(a, b) for a, b in zip(comp[:-1], comp[1:]) if 'raster_bwd' in rows[a][0] and 'raster_fwd' in rows[b][0] and rows[b][1] - rows[a][2] > 1.2e6]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(bounds) // 2
a, b = bounds[which]
t0 = rows[a][2]
print(f"{len(bounds)} step boundaries; number {which}: {((rows[b][1] - t0) / 1e6):.3f} ms between the last backward's end and the next step's first forward")
for n, s, e, q, hq in rows[a:b + 1]:
    if e < t0 and n is not rows[a][0]:
        continue
    print(f"  {(s - t0) / 1e6:8.3f} {(e - s) / 1e6:7.3f}  s{q:<3d} q{hq:<3d} {short(n)}")
