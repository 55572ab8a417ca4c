#!/bin/bash
# Timing ablations given as patches (development aid): apply, rebuild, time, reverse; the tree ends as it started.
# usage: bash tools/ab_ablate.sh "<timing command>" patch1 patch2 ...
cd $GRAFT_REPO_ROOT
CMD=$1; shift
build() { timeout 900 python3 project-nerf_amd/build.py -q > /dev/null || exit 1; }
APPLIED=""
restore() { if [ -n "$APPLIED" ]; then patch -p1 -s -R < $APPLIED; APPLIED=""; build; fi; }   # never leave an ablated tree or library behind
trap restore EXIT
echo "== as is"; timeout -k 10 300 $CMD 2>&1 | grep -v amdgpu.ids | tail -2
for P in "$@"; do
  patch -p1 -s < $P || exit 1; APPLIED=$P; build
  echo "== $P"; timeout -k 10 300 $CMD 2>&1 | grep -v amdgpu.ids | tail -2
  restore
done
