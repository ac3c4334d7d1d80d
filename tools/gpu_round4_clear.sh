#!/bin/bash
# sdm_clear A/B: slot-per-thread kernel (old) against the piece-linear one (records rewritten whole / kept pieces left
# alone / plain instead of non-temporal accesses); HBM counters of each; the clear tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/clear
timeout 600 python -m pytest tests/test_clear_gpu.py tests/test_kat_gpu.py tests/test_parity_edge_gpu.py -k clear -m gpu -x -q 2>&1 | tail -5
for round in 1 2; do
  for tag in old rec1 rec0 rec1t; do
    SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 300 python tools/probes/clear_time.py 10 2>&1 | tail -1
  done
done
for tag in old rec1 rec0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    SDM_LIB_PATH=build/ab/libsdm_$tag.so timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/pmc_clear_${tag}_$c -o s -- python tools/probes/clear_time.py 3 > gpurun_out/clear/${tag}_$c.log 2>&1
    grep -E "Kernel_Name|k_clear" gpurun_out/pmc_clear_${tag}_$c/s_counter_collection.csv > gpurun_out/clear/${tag}_$c.csv
    rm -rf gpurun_out/pmc_clear_${tag}_$c
  done
  python - $tag <<'PY'
import csv, sys
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open("gpurun_out/clear/%s_%s.csv" % (tag, c))))
    rows = [r for r in rows if "k_clear_slots" in r["Kernel_Name"] or "k_clear_map" in r["Kernel_Name"]]
    out[c] = (sum(float(r["Counter_Value"]) for r in rows) / len(rows), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / len(rows) / 1e3, len(rows))
f, w = out["FETCH_SIZE"], out["WRITE_SIZE"]
print(tag, "fetch GB (x2 corrected) %.3f  write GB %.3f  kernel us %.1f / %.1f  launches %d" % (f[0] * 1024 * 2 / 1e9, w[0] * 1024 / 1e9, f[1], w[1], f[2]))
PY
done
