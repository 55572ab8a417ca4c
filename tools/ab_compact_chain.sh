#!/bin/bash
# same-box A/B of the chained / publishing compaction against the copy + event form (NERF_NO_COMPACT_CHAIN=1): Part 4 step and the
# Instant steady-state loop, twice, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
python3 tools/instant_profile_loop.py train /tmp/ab_state.pt > /dev/null 2>&1
for rep in 1 2; do
  for v in "" 1; do
    echo "NERF_NO_COMPACT_CHAIN=$v"
    NERF_NO_COMPACT_CHAIN=$v python3 tools/instant_profile_loop.py loop /tmp/ab_state.pt 300 2>/dev/null | tail -n 1
    NERF_NO_COMPACT_CHAIN=$v python3 tools/time_part4.py 2>/dev/null | head -n 1
  done
done
