#!/usr/bin/env python
"""profiles/<tag>_sweep_pmc.json from the two counter passes of tools/pmc_sweep.sh.

Launch order of the bench command: one non-incremental sweep (k_occupancy_scan + k_occupancy_dense, first frame after
sdm_load_state), k_occupancy<S> for the other warm-up + timed frames, then 6 profiled frames (the ones bench.py takes the
in-frame launch time, tile and voxel counts from), 6 x-shift frames, then the non-incremental sweeps on the benchmark map
(201 untimed, 1 + 10 timed), on the dense map (eight track ids per slot: 41 untimed, 1 + 10 timed) and on the dense map with one
track id per voxel (`dense_case_surface`, the same).  A non-incremental sweep is two launches (three where the library took the
lists): their counters and durations are added.  FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B; calibrated there for
wide coalesced streaming reads, so for the in-frame launch with its scattered record fetches the corrected figure is an
upper estimate)."""
import csv
import json
import shutil
import sys

tag = sys.argv[1]
g = "gpurun_out/"
b = json.loads(open(g + "%s_sweep_bench.json" % tag).read())
rf = b["roofline"]
res = {}
for c in ["FETCH_SIZE", "WRITE_SIZE"]:
    rows = list(csv.DictReader(open(g + "%s_sweep_%s.csv" % (tag, c))))
    def one(r):
        return [(float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)]
    inc = [one(r) for r in rows if "k_occupancy<" in r["Kernel_Name"]]
    # a non-incremental sweep = k_occupancy_scan (or k_occupancy_scan_lists + k_occupancy_listed) + k_occupancy_dense, in launch order
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # (round 6: a sweep of a map whose every group was dense the last time is k_occupancy_dense alone)
    allr, closed = [], True
    for r in rows:
        n = r["Kernel_Name"]
        if "k_occupancy_scan" in n:
            allr.append(one(r))
            closed = False
        elif "k_occupancy_listed" in n:
            allr[-1] += one(r)
        elif "k_occupancy_dense" in n:
            if closed:
                allr.append(one(r))
            else:
                allr[-1] += one(r)
            closed = True
    n = len(allr)
    # 1 (first frame) + 201 + 11 (benchmark map: untimed, then 1 + 10 timed) + 41 + 11 (dense case) + 41 + 11 (dense case, one track id per voxel)
    assert n == 317, (n, len(rows))
    sets = {"in_frame": inc[-12:-6], "x_shift_frames": inc[-6:], "full_evaluation": allr[n - 114:n - 104], "dense_case": allr[n - 62:n - 52],
            "dense_case_surface": allr[n - 10:]}
    for name, rs in sets.items():
        v = [sum(x[0] for x in r) for r in rs]
        d = [sum(x[1] for x in r) for r in rs]
        res.setdefault(name, {})[c] = (sum(v) / len(v), sum(d) / len(d), len(v))
    shutil.copy(g + "%s_sweep_%s.csv" % (tag, c), "profiles/%s_sweep_pmc_%s.csv" % (tag, c.lower()))
out = {"kernel": rf["kernel"], "non_incremental_kernels": "k_occupancy_scan + k_occupancy_dense (two launches, added; the first sweep of a map: k_occupancy_scan_lists + k_occupancy_listed + k_occupancy_dense; the dense cases after their second sweep: k_occupancy_dense alone)", "voxels": rf["voxels"], "voxels_evaluated_in_full": rf["voxels_evaluated_in_full"], "tiles": rf["tiles"],
       "tiles_looked_into": rf["tiles_looked_into"],
       "fetch_correction": "x2 (MI355X_MICROARCH.md, HBM section)", "cases": {},
       "commands": ["SDM_GRAPH=0 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -- python bench.py --no-cpu --no-strong --no-stress --steps 20 --warmup 5",
                    "SDM_GRAPH=0 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -- python bench.py --no-cpu --no-strong --no-stress --steps 20 --warmup 5"]}
layout = {"in_frame": rf["bytes_per_launch"], "full_evaluation": rf.get("full_evaluation", {}).get("bytes_per_launch"),
          "dense_case": rf.get("dense_case", {}).get("bytes_per_launch"),
          "dense_case_surface": rf.get("dense_case_surface", {}).get("bytes_per_launch"), "x_shift_frames": None}
for name, r in res.items():
    f_kib, us_f, n = r["FETCH_SIZE"]
    w_kib, us_w, _ = r["WRITE_SIZE"]
    traffic = int(f_kib * 1024 * 2 + w_kib * 1024)
    us = (us_f + us_w) / 2
    out["cases"][name] = {"launches_averaged": n, "FETCH_SIZE_KiB_per_launch": round(f_kib, 1), "WRITE_SIZE_KiB_per_launch": round(w_kib, 1),
                          "fetch_bytes_per_launch": int(f_kib * 1024 * 2), "write_bytes_per_launch": int(w_kib * 1024),
                          "traffic_bytes_per_launch": traffic, "avg_kernel_us_under_pmc": round(us, 1),
                          "traffic_GBps": round(traffic / us / 1e3, 1), "traffic_frac_of_8TBps": round(traffic / us / 1e3 / 8000.0, 4),
                          "layout_bytes_per_launch": layout[name],
                          "traffic_over_layout": round(traffic / layout[name], 3) if layout[name] else None}
out["traffic_bytes_per_launch"] = out["cases"]["in_frame"]["traffic_bytes_per_launch"]
json.dump(out, open("profiles/%s_sweep_pmc.json" % tag, "w"), indent=1)
print(json.dumps({k: (v["traffic_bytes_per_launch"], v["layout_bytes_per_launch"], v["avg_kernel_us_under_pmc"], v["traffic_frac_of_8TBps"])
                  for k, v in out["cases"].items()}))
