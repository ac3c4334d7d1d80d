#!/usr/bin/env python
"""Timeline of the last frames of a rocprofv3 --kernel-trace CSV: per frame (k_set_frame .. next k_set_frame) the kernels in
start order with start offset, duration and the gap to the previous kernel's end."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("sdm::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
starts = [i for i, r in enumerate(rows) if name(r).startswith("k_set_frame")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -8
a, b = starts[which], starts[which + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
print("frame span %.1f us, %d kernels" % ((max(int(r["End_Timestamp"]) for r in rows[a:b]) - t0) / 1e3, b - a))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %7.1f  gap %6.1f  q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), name(r)))
    prev_end = max(prev_end, e)
