#!/bin/bash
# timing sweep of the infer64 stream generator's tunables (window depth GEN_D64, epilogue instructions per fragment GEN_S64_MIN, two units interleaved GEN_S64_ORDER; SWEEP="d min order;..."):
# regenerate, rebuild, A/B against the 32-sample stream (tools/ab_infer64.py); results stay bit-equal
cd "$GRAFT_REPO_ROOT" || exit 1
IFS=";" read -ra CFGS <<< "${SWEEP:-4 6 0;4 6 1;4 5 0;4 7 0}"
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  (cd project-nerf_amd/csrc && GEN_D64=$1 GEN_S64_MIN=$2 GEN_S64_ORDER=${3:-1} python3 gen_stream_asm.py > mlp_stream_asm.h 2>/dev/null)
  NERF_BUILD_KEEP_HEADERS=1 python3 project-nerf_amd/build.py > /dev/null 2>&1 || { echo "build failed for $cfg"; continue; }
  echo "== GEN_D64=$1 GEN_S64_MIN=$2 GEN_S64_ORDER=${3:-1}"
  timeout -k 10 200 python3 tools/ab_infer64.py 2>/dev/null | grep -v "^rgb\|^ragged" | tail -n 4
done
rm -f project-nerf_amd/csrc/mlp_stream_asm.h
python3 project-nerf_amd/build.py > /dev/null 2>&1
