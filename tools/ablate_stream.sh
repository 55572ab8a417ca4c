#!/bin/bash
# Timing ablations of the inference stream kernel: regenerate the asm with pieces removed, rebuild
# mlp_fwd.hip, time 65,536 x 128 samples.  Results of ablated builds are numerically wrong.
cd $GRAFT_REPO_ROOT/project-nerf_amd/csrc
for v in "${@}"; do
  IFS=: read no d epi <<< "$v"
  GEN_NO=$no GEN_D=${d:-6} GEN_EPI=${epi:-1} python3 gen_fwd_stream_asm.py > mlp_fwd_stream_asm.h 2>/dev/null
  (cd ../.. && timeout 600 python3 project-nerf_amd/build.py -q > /dev/null) || exit 1
  echo "== NO=$no D=${d:-6} EPI=${epi:-1}: $(cd ../.. && NERF_FWD_CYCLES=1 timeout -k 10 200 python3 tools/time_decoder.py 2>&1 | grep 'fwd \|cycles' | tail -2 | tr '\n' ' ')"
done
