#!/bin/bash
# per-level kernel times of the binned hash backward (rocprofv3 averages, us): count | plan | scatter | reduce
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r03/hash_bwd_levels
mkdir -p $out
for spec in "0 16" "0 1" "1 2" "2 3" "3 4" "4 5" "5 6" "6 7" "8 9" "11 12" "15 16"; do
  set -- $spec
  tag=l$1_$2
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 tools/hash_bwd_level.py $1 $2 > $out/$tag.log 2>&1 || { echo "$tag failed"; tail -3 $out/$tag.log; exit 1; }
  python3 - $out/${tag}_kernel_stats.csv "$1" "$2" <<'PY'
import csv, sys
t = {}
for r in csv.DictReader(open(sys.argv[1])):
    for k in ("count", "plan", "scatter", "reduce"):
        if f"hash_bin_{k}" in r["Name"]:
            t[k] = t.get(k, 0.0) + float(r["TotalDurationNs"]) / 30 / 1e3
print(f"levels [{sys.argv[2]}, {sys.argv[3]}): " + " | ".join(f"{k} {t.get(k, 0):6.1f}" for k in ("count", "plan", "scatter", "reduce")) + f" | sum {sum(t.values()):6.1f} us", flush=True)
PY
done
find $out -name '*_kernel_trace.csv' -delete
