#!/bin/bash
# SQ counters of the training chain kernels (forward + 8-bit images, dgrad), 8 passes per workgroup, on the whole
# chip and on half of it (option chain_grid): held clock = GRBM_GUI_ACTIVE / 8 / duration, MFMA-busy fraction.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for grid in 256 128; do
  out=gpurun_out/pmc_chain/g$grid
  mkdir -p $out
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out -o sq -- python3 tools/pmc_train_chains_loop.py $grid > $out/sq.log 2>&1 || exit 1
  python3 - $out $grid <<'PY'
import csv, collections, sys
out, grid = sys.argv[1], int(sys.argv[2])
for key in ("mlp_fwd_stream_kernel", "mlp_bwd_stream_kernel"):
    c = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{out}/sq_counter_collection.csv")):
        if key in r["Kernel_Name"]:
            c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f"{out}/sq_kernel_trace.csv")) if key in r["Kernel_Name"]]
    av = lambda k: sum(c[k][1:]) / len(c[k][1:])
    g, m, d = av("GRBM_GUI_ACTIVE") / 8, av("SQ_VALU_MFMA_BUSY_CYCLES"), sum(dur[1:]) / len(dur[1:])
    print(f"grid {grid:3d} {key:24s} {d/1e3:8.1f} us  cycles {g:9.0f}  clock {g/d:.3f} GHz  mfma_busy {m/(g*4*grid):.3f}  "
          f"wait_any/wave_cycles {av('SQ_WAIT_INST_ANY')/av('SQ_WAVE_CYCLES'):.3f}")
PY
done
