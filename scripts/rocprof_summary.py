#!/usr/bin/env python3
"""Dump the per-kernel statistics of a rocprofv3 rocpd database (…_results.db) as a text table
(durations in microseconds, as rocprofv3's own `top_kernels` view reports them).
usage: rocprof_summary.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    regs = {}
    for name, v, a, s, lds, gx, wx in cur.execute(
            "select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
            "from kernels group by name"):
        regs[name] = (v, a, s, lds, gx, wx)
    lines = [f"{'calls':>7} {'total_us':>12} {'avg_us':>11} {'pct':>7} {'vgpr':>5} {'sgpr':>5} {'lds':>7} {'grid':>9} {'wg':>5}  kernel"]
    for name, calls, total, avg, pct in rows:
        v, a, s, lds, gx, wx = regs.get(name, (0, 0, 0, 0, 0, 0))
        lines.append(f"{calls:7d} {total:12.1f} {avg:11.2f} {pct:7.2f} {v:5d} {s:5d} {lds:7d} {gx:9d} {wx:5d}  {name[:100]}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
