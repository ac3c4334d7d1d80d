"""Development aid: which hardware queue did each of a map's streams land on?  Reads a rocprofv3 --kernel-trace CSV of
tools/probes/crossframe.py (several maps in one process; every map starts with k_pack_pos4 = its sdm_load_state) and prints,
per map, the queue ids its main stream (k_frame_begin), frustum chain (k_vertex_mask) and birth chain (k_birth_candidates) ran on."""
import csv
import sys
from collections import Counter, defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seg = -1
per = defaultdict(lambda: defaultdict(Counter))
for r in rows:
    n = r["Kernel_Name"]
    if "k_pack_pos4" in n:
        seg += 1
    for key in ("k_frame_begin", "k_vertex_mask", "k_birth_candidates", "k_occupancy<", "k_set_frame", "k_pack_pos4"):
        if key in n:
            per[seg][key][r.get("Queue_Id", "?")] += 1
for s in sorted(per):
    print("map %d: " % s + "; ".join("%s on queue(s) %s" % (k, dict(v)) for k, v in per[s].items()))
