#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite output): per-kernel statistics over the whole run and the
launch timeline of one frame.  usage: trace_db.py <results.db> [frames_from_end]"""
import sqlite3
import sys
from collections import defaultdict


def short(n):
    return n.replace("sdm::(anonymous namespace)::", "").replace("void ", "").split("(")[0]


c = sqlite3.connect(sys.argv[1])
rows = [(short(n), s, e, st) for n, s, e, st in c.execute("select name,start,end,stream_id from kernels order by start")]
stat = defaultdict(list)
for n, s, e, _ in rows:
    stat[n].append((e - s) / 1e3)
tot = sum(sum(v) for v in stat.values())
print("%-44s %7s %10s %10s %10s %10s %6s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_us", "%"))
for k, v in sorted(stat.items(), key=lambda kv: -sum(kv[1])):
    print("%-44s %7d %10.2f %10.2f %10.2f %10.1f %6.2f" % (k[:44], len(v), sum(v) / len(v), min(v), max(v), sum(v),
                                                         100 * sum(v) / tot))
idx = [i for i, r in enumerate(rows) if r[0].startswith("k_frame_begin")]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if len(idx) > back:
    lo, hi = idx[-back - 1], idx[-back]
    t0 = rows[lo][1]
    print("\ntimeline of one frame under the profiler (us from k_frame_begin): start dur stream kernel")
    for n, s, e, st in rows[lo:hi]:
        print("%8.1f %7.1f  s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, st, n[:60]))
    print("next frame begins at %.1f" % ((rows[hi][1] - t0) / 1e3))
