#!/bin/bash
# Timing ablations of the stream kernels: regenerate the asm with pieces removed (GEN_NO=dma,epi,bar,
# read,store), rebuild, time the decoder kernels.  Results of ablated builds are numerically wrong,
# so the default stream and library are restored on exit (and build.py refuses to reuse a header whose
# GEN_CONFIG line is not the default, should the restore ever be skipped).
# usage: bash tools/ablate_stream.sh "<GEN_NO>:<GEN_D>" ...
cd $GRAFT_REPO_ROOT/project-nerf_amd/csrc
restore() {
  rm -f mlp_stream_asm.h
  (cd ../.. && env -u GEN_NO -u GEN_D timeout 900 python3 project-nerf_amd/build.py -q > /dev/null)
}
trap restore EXIT
for v in "${@}"; do
  IFS=: read no d <<< "$v"
  GEN_NO=$no GEN_D=${d:-4} python3 gen_stream_asm.py > mlp_stream_asm.h 2>/dev/null
  touch mlp_stream_asm.h
  (cd ../.. && NERF_BUILD_KEEP_HEADERS=1 timeout 900 python3 project-nerf_amd/build.py -q > /dev/null) || exit 1
  echo "== NO=$no D=${d:-4}: $(cd ../.. && timeout -k 10 200 python3 tools/time_decoder.py 2>&1 | grep 'fwd\|bwd' | tr '\n' ' ')"
done
