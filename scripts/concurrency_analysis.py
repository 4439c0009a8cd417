"""Concurrency of the three-stream engine over the LAST full step of a rocprofv3 kernel trace of bench.py (rocpd sqlite):
how long is no kernel / exactly one kernel / several kernels in flight, and which kernels run alone."""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kt} d join {ks} s on d.kernel_id=s.id order by d.start"))
short = lambda n: n.split('(')[0].replace('void ', '')[:40]
ap = [i for i, r in enumerate(rows) if ('tile_apply' in r[0] or 'specular_apply' in r[0])]
rb = [i for i, r in enumerate(rows) if 'raster_bwd' in r[0]]
groups = []                                               # runs of prefilter applies (one run = one direction of one step)
for i in ap:                                              # (by TIME: the other streams' kernels sit between the applies of one run)
    if groups and rows[i][1] - rows[groups[-1][-1]][2] < 0.25e6:
        groups[-1].append(i)
    else:
        groups.append([i])
steps = [(a[0], b[-1]) for a, b in zip(groups[:-1], groups[1:]) if sum(1 for r in rb if a[-1] < r < b[0]) == 8]
# an engine step: prefilter forward ... views ... prefilter backward; of the candidates take the one with the median wall time
# (the window between two prefilter runs of bench.py's own measurement loops can also hold 8 compositor launches)
walls = sorted((max(r[2] for r in rows[a:b + 1]) - rows[a][1], a, b) for a, b in steps)
_, lo, hi = walls[len(walls) // 2] if len(walls) > 2 else walls[0]
seg = rows[lo:hi + 1]
t0, t1 = seg[0][1], max(r[2] for r in seg)
ev = []
for n, s, e in seg:
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
live = collections.Counter(); depth_time = collections.Counter(); alone = collections.Counter(); last = t0
for t, d, n in ev:
    k = sum(live.values())
    depth_time[k] += t - last
    if k == 1:
        alone[short(next(x for x, v in live.items() if v > 0))] += t - last
    last = t
    live[n] += d
wall = t1 - t0
print(f"one engine step (median of {len(steps)} candidates): {len(seg)} kernels, wall {wall / 1e6:.2f} ms, sum of kernel durations {sum(r[2] - r[1] for r in seg) / 1e6:.2f} ms")
for k in sorted(depth_time):
    print(f"  {k} kernels in flight: {depth_time[k] / 1e6:7.3f} ms ({100.0 * depth_time[k] / wall:4.1f} %)")
print("  alone on the GPU:")
for n, v in alone.most_common(14):
    print(f"    {v / 1e3:8.1f} us  {n}")
# the compositor (main stream) is the step's critical path: where does it start, how long are its kernels inside the step, and what
# lies between them (tone map + loss glue, waiting for the next view's front)?
print("  main-stream timeline (ms from the step's first kernel): kernel, start, duration, gap since the previous compositor kernel")
prev_end = None
for n, s, e in seg:
    if 'raster_fwd' in n or 'raster_bwd' in n:
        gap = (s - prev_end) / 1e6 if prev_end is not None else float('nan')
        print(f"    {short(n):36s} {(s - t0) / 1e6:7.3f} {(e - s) / 1e6:6.3f}   gap {gap:6.3f}")
        prev_end = e
print(f"    last compositor kernel ends at {(prev_end - t0) / 1e6:.3f} ms; step ends at {wall / 1e6:.3f} ms")
for pat in ('tile_apply', 'shade_bwd', 'project_bwd', 'project_fwd', 'shade_fwd', 'build_stream'):
    xs = [(s - t0, e - t0) for n, s, e in seg if pat in n]
    if xs:
        print(f"    {pat:14s} starts (ms): " + " ".join(f"{a / 1e6:.2f}" for a, _ in xs[:20]))
# everything the compositor's own stream runs around the first three views (what sits between a forward and its backward)
try:
    cols = [r[1] for r in c.execute(f"pragma table_info({kt})")]
    key = 'stream_id' if 'stream_id' in cols else ('queue_id' if 'queue_id' in cols else None)
    if key:
        full = list(c.execute(f"select s.kernel_name, d.start, d.end, d.{key} from {kt} d join {ks} s on d.kernel_id=s.id where d.start >= {t0} and d.start <= {t1} order by d.start"))
        main_id = next(r[3] for r in full if 'raster_fwd' in r[0])
        print(f"  kernels of the compositor's stream ({key} {main_id}), first three views: start, duration (ms)")
        seen_bwd = 0
        for n, s_, e_, q_ in full:
            if q_ != main_id:
                continue
            print(f"    {(s_ - t0) / 1e6:7.3f} {(e_ - s_) / 1e6:6.3f}  {short(n)}")
            seen_bwd += 'raster_bwd' in n
            if seen_bwd == 3:
                break
except Exception as ex:
    print("  (no per-stream listing:", ex, ")")
