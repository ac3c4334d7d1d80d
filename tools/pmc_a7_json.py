#!/usr/bin/env python
"""gpurun_out/<tag>_a7_sq.json from the counter passes of tools/pmc_a7.sh: per kernel and workload the SQ counters
(averages per launch), instruction mix per wave, VALU and LDS issue utilisation.

Utilisation = quad-cycles the SIMDs spent issuing that instruction class (SQ_ACTIVE_INST_*: one count = 4 shader cycles of
one SIMD, MI355X_MICROARCH.md) / (kernel duration x 2.4 GHz / 4 x 1024 SIMDs); the duration is the dispatch's in the
kernel trace of the same run.  (GRBM_GUI_ACTIVE, which an earlier version of this script divided by, is summed over the
8 XCDs - 8 x the kernel's cycles - and made every utilisation 8 x too small.)"""
import csv
import glob
import json
import sys
from collections import defaultdict

tag = sys.argv[1]
KERNELS = ("k_ck", "k_weight", "k_visibility", "k_birth_replay", "k_occupancy<")
SIMDS = 256 * 4
CLOCK_MHZ = 2400.0


def short(n):
    return n.replace("sdm::(anonymous namespace)::", "").replace("void ", "").split("(")[0]


out = {"how": "rocprofv3 --kernel-trace --pmc <8 SQ counters> per pass, separate passes; averages over the launches of the run; "
              "SDM_GRAPH=0 (launch by launch) so that every kernel is its own dispatch",
       "workloads": {}}
for wl in ("c3", "stress"):
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for p in ("valu", "lds"):
        for fn in glob.glob("gpurun_out/pmca7_%s_%s_%s/**/*counter_collection.csv" % (tag, wl, p), recursive=True):
            for r in csv.DictReader(open(fn)):
                k = short(r["Kernel_Name"])
                if not k.startswith(KERNELS):
                    continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for fn in glob.glob("gpurun_out/pmca7_%s_%s_%s/**/*kernel_trace.csv" % (tag, wl, p), recursive=True):
            for r in csv.DictReader(open(fn)):
                k = short(r["Kernel_Name"])
                if k.startswith(KERNELS):
                    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    res = {}
    for k in sorted(acc):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        dur_us = sum(dur[k]) / max(len(dur[k]), 1)
        quads = dur_us * CLOCK_MHZ / 4.0 * SIMDS
        waves = max(c.get("SQ_WAVES", 1.0), 1.0)
        res[k] = {"launches": max(len(v) for v in acc[k].values()), "avg_us_under_pmc": round(sum(dur[k]) / max(len(dur[k]), 1), 2),
                  "counters": {n: round(v, 1) for n, v in sorted(c.items())},
                  "per_wave": {"valu": round(c.get("SQ_INSTS_VALU", 0) / waves, 1), "salu": round(c.get("SQ_INSTS_SALU", 0) / waves, 1),
                               "lds": round(c.get("SQ_INSTS_LDS", 0) / waves, 1), "vmem": round(c.get("SQ_INSTS_VMEM", 0) / waves, 1)},
                  "valu_issue_utilisation": round(c.get("SQ_ACTIVE_INST_VALU", 0) / quads, 4) if quads else None,
                  "lds_issue_utilisation": round(c.get("SQ_ACTIVE_INST_LDS", 0) / quads, 4) if quads else None,
                  "any_issue_utilisation": round(c.get("SQ_ACTIVE_INST_ANY", 0) / quads, 4) if quads else None,
                  "lds_bank_conflict_fraction": round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 4)
                  if c.get("SQ_LDS_IDX_ACTIVE") else None,
                  "wave_time_split": {n: round(c.get(s, 0) / c["SQ_WAVE_CYCLES"], 3) for n, s in
                                      (("issuing", "SQ_ACTIVE_INST_ANY"), ("waiting_memory_or_barrier", "SQ_WAIT_ANY"),
                                       ("issue_stalled", "SQ_WAIT_INST_ANY"))} if c.get("SQ_WAVE_CYCLES") else None}
    out["workloads"][wl] = res
json.dump(out, open("gpurun_out/%s_a7_sq.json" % tag, "w"), indent=1)
for wl, res in out["workloads"].items():
    for k, v in res.items():
        print(wl, k, v["avg_us_under_pmc"], "us  VALU util", v["valu_issue_utilisation"], " LDS util", v["lds_issue_utilisation"], v["per_wave"])
