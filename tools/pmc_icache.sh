#!/bin/bash
# usage (GPU box): tools/pmc_icache.sh <tag> -> gpurun_out/<tag>_icache.txt : instruction-cache requests / hits / misses per kernel of
# the benchmark frames (launch by launch), beside each kernel's waves and average duration
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SDM_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU \
  -d gpurun_out/pmci_$tag -o p -- python bench.py --no-cpu --no-dense --no-strong --steps 20 --warmup 5 > gpurun_out/pmci_$tag.log 2>&1
python tools/pmc_summary.py gpurun_out/pmci_$tag > gpurun_out/${tag}_icache.csv 2>&1
python - <<PY
import csv, glob
dur = {}
for f in glob.glob("gpurun_out/pmci_$tag/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("sdm::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        dur.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = list(csv.DictReader(open("gpurun_out/${tag}_icache.csv")))
out = []
for r in rows:
    k = r["kernel"]
    try:
        req, hit, miss, dup = (float(r[c]) for c in ("SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE"))
        w, iv, isa = float(r["SQ_WAVES"]), float(r["SQ_INSTS_VALU"]), float(r["SQ_INSTS_SALU"])
    except (ValueError, KeyError):
        continue
    d = dur.get(k, [0.0])
    out.append((sum(d) / len(d), k, w, (iv + isa) / max(w, 1), req, miss, dup, miss / max(req, 1)))
print("%-34s %8s %8s %10s %10s %10s %10s %6s" % ("kernel", "avg us", "waves", "insts/wave", "ic req", "ic miss", "miss dup", "miss%"))
for a in sorted(out, reverse=True)[:24]:
    print("%-34s %8.1f %8.0f %10.0f %10.0f %10.0f %10.0f %6.1f" % (a[1][:34], a[0], a[2], a[3], a[4], a[5], a[6], 100 * a[7]))
PY
rm -rf gpurun_out/pmci_$tag
