#!/usr/bin/env python
"""Timeline of the last SemanticDSPMap::update of an adapter run traced with rocprofv3 --kernel-trace --memory-copy-trace
(rocpd sqlite output): kernels and copies in start order.  usage: adapter_timeline.py <results.db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
ev = []
for n, s, e, st in c.execute("select name,start,end,stream_id from kernels"):
    ev.append((s, e, "s%s" % st, n.replace("sdm::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]))
mc = [n for n in names if n == "memory_copies"]
if mc:
    cols = [r[1] for r in c.execute("pragma table_info(memory_copies)")]
    size_col = "size" if "size" in cols else None
    q = "select name,start,end%s from memory_copies" % ("," + size_col if size_col else "")
    for row in c.execute(q):
        ev.append((row[1], row[2], "copy", "%s %s B" % (row[0], row[3] if size_col else "?")))
else:
    print("no memory_copies view; have:", names)
ev.sort()
idx = [i for i, r in enumerate(ev) if r[3].startswith("k_labeled_cloud")]
if not idx:
    print("no k_labeled_cloud in trace")
    sys.exit(0)
lo = idx[-2] if len(idx) > 1 else 0
# include the uploads that precede the cloud kernel
while lo > 0 and ev[lo - 1][2] == "copy" and ev[lo][0] - ev[lo - 1][1] < 400e3:
    lo -= 1
hi = idx[-1]
while hi > lo and ev[hi - 1][2] == "copy" and "HOST_TO_DEVICE" in ev[hi - 1][3].upper():
    hi -= 1
t0 = ev[lo][0]
print("one update() on the device (us from the first upload): start dur where what")
for s, e, st, n in ev[lo:hi]:
    print("%8.1f %8.1f  %-5s %s" % ((s - t0) / 1e3, (e - s) / 1e3, st, n))
