#!/bin/bash
# A/B of generator variants of the 16x16x32 inference stream (development aid): by default each
# fragment read sits between the two MFMAs that share an A fragment; GEN_NOSPLIT16 puts it after them.  Restores the default build on exit.
cd $GRAFT_REPO_ROOT/project-nerf_amd/csrc
restore() { rm -f mlp_stream_asm.h; (cd ../.. && env -u GEN_NOSPLIT16 timeout 900 python3 project-nerf_amd/build.py -q > /dev/null); }
trap restore EXIT
for v in "1" "" "1" ""; do
  if [ -n "$v" ]; then export GEN_NOSPLIT16=1; else unset GEN_NOSPLIT16; fi
  python3 gen_stream_asm.py > mlp_stream_asm.h 2>/dev/null; touch mlp_stream_asm.h
  (cd ../.. && NERF_BUILD_KEEP_HEADERS=1 timeout 900 python3 project-nerf_amd/build.py -q > /dev/null) || exit 1
  echo "== NOSPLIT16=$v: $(cd ../.. && timeout -k 10 200 python3 tools/time_decoder.py 2>&1 | grep 'R=65536' | tr '\n' ' ')"
done
