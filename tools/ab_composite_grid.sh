#!/bin/bash
# fused compositing backward: workgroups per CU (same-address atomics at the end of every workgroup) -- kernel time inside the Part 4 step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
for w in 8 4 2 1 8; do
  export NERF_COMPOSITE_WGS=$w
  timeout -k 10 200 python tools/time_part4.py > gpurun_out/r03/ab_cg.txt 2>&1 || { echo "failed"; tail -5 gpurun_out/r03/ab_cg.txt; exit 1; }
  echo "wgs/cu $w: $(grep 'part4 engine step' gpurun_out/r03/ab_cg.txt | cut -c1-45) | $(grep 'composite_mse' gpurun_out/r03/ab_cg.txt | cut -c1-25) | $(grep 'GPU time per step' gpurun_out/r03/ab_cg.txt | cut -c1-70)"
done
