"""Per-queue timeline of the LAST step in a rocprofv3 kernel trace (rocpd sqlite): a step = the window between the starts of two
consecutive prefilter-forward runs.  Prints, per queue, the busy time, and every kernel longer than `min_us` (start, duration)
with the gap to the previous kernel of the same queue.   usage: python scripts/stream_timeline.py <results.db> [min_us] [step_from_end]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kt})")]
key = 'stream_id' if 'stream_id' in cols else 'queue_id'
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.{key} from {kt} d join {ks} s on d.kernel_id=s.id order by d.start"))
short = lambda n: n.split('(')[0].replace('void ', '')[:44]
# step boundaries: first forward tile_apply of each prefilter run (runs separated by > 2 ms without an apply)
ap = [i for i, r in enumerate(rows) if 'tile_apply_kernel<false' in r[0] or ('tile_apply' in r[0] and 'false' in r[0])]
if not ap:
    ap = [i for i, r in enumerate(rows) if 'tile_apply' in r[0]]
starts = []
for i in ap:
    if not starts or rows[i][1] - rows[starts[-1]][1] > 4e6:
        starts.append(i)
lo, hi = starts[-back - 1], starts[-back]
seg = rows[lo:hi]
t0 = seg[0][1]
wall = (rows[hi][1] - t0) / 1e6
print(f"step of {len(seg)} kernels, {wall:.3f} ms between two prefilter starts; queues by {key}")
byq = collections.defaultdict(list)
for n, s, e, q in seg:
    byq[q].append((n, s, e))
for q, xs in sorted(byq.items(), key=lambda kv: kv[1][0][1]):
    busy = sum(e - s for _, s, e in xs) / 1e6
    print(f"-- queue {q}: {len(xs)} kernels, busy {busy:.3f} ms")
    prev = None
    for n, s, e in xs:
        if (e - s) / 1e3 >= min_us:
            gap = float('nan') if prev is None else (s - prev) / 1e6
            print(f"   {(s - t0) / 1e6:8.3f} {(e - s) / 1e6:7.3f}  gap {gap:7.3f}  {short(n)}")
        prev = e
