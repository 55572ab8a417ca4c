#!/bin/bash
# tools/time_chain_grid.py on the default stream and on timing ablations of it (GEN_NO, results wrong);
# the default stream and library are restored on exit.  usage: bash tools/chain_grid_ab.sh stinst store ...
cd $GRAFT_REPO_ROOT
restore() {
  rm -f project-nerf_amd/csrc/mlp_stream_asm.h
  env -u GEN_NO -u GEN_D timeout 900 python3 project-nerf_amd/build.py -q > /dev/null
}
trap restore EXIT
echo "== default"; timeout -k 10 300 python3 tools/time_chain_grid.py 2>&1 | grep workgroups || exit 1
for no in "$@"; do
  (cd project-nerf_amd/csrc && GEN_NO=$no python3 gen_stream_asm.py > mlp_stream_asm.h 2>/dev/null && touch mlp_stream_asm.h)
  NERF_BUILD_KEEP_HEADERS=1 timeout 900 python3 project-nerf_amd/build.py -q > /dev/null || exit 1
  echo "== NO=$no"; timeout -k 10 300 python3 tools/time_chain_grid.py 2>&1 | grep workgroups || exit 1
done
