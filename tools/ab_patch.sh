#!/bin/bash
# Same-box A/B of a source patch (development aid): time the tree as it is, reverse-apply the patch, rebuild, time
# again, restore; twice, interleaved, so that clock drift of the box shows.  usage: bash tools/ab_patch.sh <patch> [timing command]
cd $GRAFT_REPO_ROOT
PATCH=$1; shift
CMD=${@:-python3 tools/time_chain_grid.py}
build() { rm -f project-nerf_amd/csrc/mlp_stream_asm.h; timeout 900 python3 project-nerf_amd/build.py -q > /dev/null || exit 1; }
restore() { patch -p1 -s -N < $PATCH > /dev/null 2>&1; build; }
trap restore EXIT
for rep in 1 2; do
  echo "== with the patch (round $rep)"; timeout -k 10 300 $CMD 2>&1 | grep -v amdgpu.ids | tail -3
  patch -p1 -s -R < $PATCH || exit 1; build
  echo "== without the patch (round $rep)"; timeout -k 10 300 $CMD 2>&1 | grep -v amdgpu.ids | tail -3
  patch -p1 -s < $PATCH || exit 1; build
done
