#!/bin/bash
# HBM fetch of the hash-grid forward in the level-major 2-D launch and in the XCD-aware launch (FETCH_SIZE pass only)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r03/pmc_hash_fwd
mkdir -p $out
for spec in "1 0" "1 36" "0 36" "0 0"; do
  set -- $spec
  tag=x$1_l$2
  timeout -k 10 120 python3 tools/hash_fwd_mode.py $1 $2 > $out/$tag.time 2>&1 || { echo "$tag timing failed"; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -o $tag -- python3 tools/hash_fwd_mode.py $1 $2 > $out/$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/$tag.log; exit 1; }
  python3 - $out/${tag}_counter_collection.csv $tag <<'PY'
import csv, sys, statistics
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "hash_fwd_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
# FETCH_SIZE counts 32-byte units... the guide's gfx950 correction: x2 of the KiB reading
print(sys.argv[2], "launches", len(v), "median FETCH_SIZE", statistics.median(v), "-> MB (x1024 x2):", statistics.median(v) * 1024 * 2 / 1e6)
PY
  grep "us" $out/$tag.time | tail -2
done
find $out \( -name '*_counter_collection.csv' -o -name '*_kernel_trace.csv' \) -delete
