#!/bin/bash
# GPU durations (kernel trace) of the wgrad skeleton modes: whole, compute-only, DMA-only, neither, at 4096 and 2048 rays
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r03/wgrad_modes
mkdir -p $out
for spec in "0 4096" "1 4096" "2 4096" "3 4096" "3 2048" "0 2048"; do
  set -- $spec
  tag=d$1_r$2
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 tools/wgrad_mode.py $1 $2 > $out/$tag.log 2>&1 || { echo "$tag failed"; tail -5 $out/$tag.log; exit 1; }
  echo "== $tag: $(tail -1 $out/$tag.log)"
  grep -E "wgrad" $out/${tag}_kernel_stats.csv | cut -c1-200
done
find $out -name '*_kernel_trace.csv' -delete
