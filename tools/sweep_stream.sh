#!/bin/bash
# timing sweep of the stream generator's tunables (window depth D, first epilogue gap EPI); results stay correct, the
# default header and library are restored on exit.  usage: bash tools/sweep_stream.sh "<D>:<EPI>" ...
cd $GRAFT_REPO_ROOT/project-nerf_amd/csrc
restore() {
  rm -f mlp_stream_asm.h
  (cd ../.. && env -u GEN_NO -u GEN_D -u GEN_EPI timeout 900 python3 project-nerf_amd/build.py -q > /dev/null)
}
trap restore EXIT
for v in "${@}"; do
  IFS=: read d e <<< "$v"
  GEN_D=${d:-4} GEN_EPI=${e:-3} python3 gen_stream_asm.py > mlp_stream_asm.h 2>/dev/null || { echo "== D=$d EPI=$e: generator failed"; continue; }
  touch mlp_stream_asm.h
  (cd ../.. && NERF_BUILD_KEEP_HEADERS=1 timeout 900 python3 project-nerf_amd/build.py -q > /dev/null) || { echo "== D=$d EPI=$e: build failed"; continue; }
  echo "== D=${d:-4} EPI=${e:-3}: $(cd ../.. && timeout -k 10 200 python3 tools/time_decoder.py 2>&1 | grep 'fwd\|bwd' | tr '\n' ' ')"
done
