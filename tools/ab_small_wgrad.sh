#!/bin/bash
# tiny-MLP weight-gradient launch: span length / workgroup cap sweep, kernel time inside the Part 4 step (torch profiler)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
for spec in "4 3" "8 3" "16 3" "32 3" "64 3" "16 2" "32 2" "4 3"; do
  set -- $spec
  export NERF_WGRAD_SMALL_SPAN=$1 NERF_WGRAD_SMALL_CAP=$2
  timeout -k 10 200 python tools/time_part4.py > gpurun_out/r03/ab_sw.txt 2>&1 || { echo "failed"; tail -5 gpurun_out/r03/ab_sw.txt; exit 1; }
  echo "span $1 cap $2: $(grep 'part4 engine step' gpurun_out/r03/ab_sw.txt | cut -c1-60) | $(grep 'mlp_wgrad_small' gpurun_out/r03/ab_sw.txt | cut -c1-30) | $(grep 'GPU time per step' gpurun_out/r03/ab_sw.txt | cut -c1-70)"
done
