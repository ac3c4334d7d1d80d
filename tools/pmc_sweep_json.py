#!/usr/bin/env python
"""profiles/<tag>_sweep_pmc.json from the two counter passes of tools/pmc_sweep.sh (in-frame sweep launches of bench.py;
only the last 6 launches count: the 6 frames after the timed region, the ones bench.py takes the sweep's launch time,
tile and voxel counts from)."""
import csv
import json
import shutil
import sys

tag = sys.argv[1]
g = "gpurun_out/"
last = 6
res = {}
for c in ["FETCH_SIZE", "WRITE_SIZE"]:
    rows = list(csv.DictReader(open(g + "%s_sweep_%s.csv" % (tag, c))))[-last:]
    v = [float(r["Counter_Value"]) for r in rows]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    res[c] = (sum(v) / len(v), sum(d) / len(d), len(v))
    shutil.copy(g + "%s_sweep_%s.csv" % (tag, c), "profiles/%s_sweep_pmc_%s.csv" % (tag, c.lower()))
b = json.loads(open(g + "%s_sweep_bench.json" % tag).read())
rf = b["roofline"]
fetch_kib, us_f, n = res["FETCH_SIZE"]
write_kib, us_w, _ = res["WRITE_SIZE"]
out = {"kernel": rf["kernel"], "voxels": rf["voxels"], "voxels_evaluated_in_full": rf["layout"]["voxels_evaluated_in_full"],
       "tiles": rf["layout"]["tiles"], "tiles_looked_into": rf["layout"]["tiles_looked_into"],
       "state": "in-frame launches of `python bench.py --no-cpu --no-dense --steps 20 --warmup 5` (C3 benchmark map), the last %d launches of the run" % last,
       "launches_averaged": n,
       "FETCH_SIZE_KiB_per_launch": round(fetch_kib, 1), "WRITE_SIZE_KiB_per_launch": round(write_kib, 1),
       "fetch_correction": "x2 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE tallies 128-B requests at 64 B; calibrated there for 16 B/lane "
                           "streaming reads - this kernel streams 16 B/lane stamps and 8 B/lane flags and gathers 128-B records, so "
                           "the corrected figure is an upper estimate)",
       "fetch_bytes_per_launch": int(fetch_kib * 1024 * 2), "write_bytes_per_launch": int(write_kib * 1024),
       "traffic_bytes_per_launch": int(fetch_kib * 1024 * 2 + write_kib * 1024),
       "layout_bytes_per_launch": rf["layout"]["bytes_per_launch"], "dense_slot_bytes_per_launch": rf["bytes_per_launch"],
       "avg_kernel_us_under_pmc": round((us_f + us_w) / 2, 1),
       "commands": ["rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -- python bench.py --no-cpu --no-dense --steps 20 --warmup 5",
                    "rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -- python bench.py --no-cpu --no-dense --steps 20 --warmup 5"]}
json.dump(out, open("profiles/%s_sweep_pmc.json" % tag, "w"), indent=1)
print(json.dumps({k: out[k] for k in ["voxels_evaluated_in_full", "traffic_bytes_per_launch", "layout_bytes_per_launch",
                                      "avg_kernel_us_under_pmc"]}))
