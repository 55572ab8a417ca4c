#!/bin/bash
# SQ counters of the inference decoder alone (65,536 x 128 samples, 3 launches): MFMA-busy fraction
# and the clock the chip holds = GRBM_GUI_ACTIVE / 8 / kernel duration.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_dec
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_dec -o sq -- python3 tools/pmc_decoder_loop.py > gpurun_out/pmc_dec/sq.log 2>&1
python3 - <<'PY'
import csv, collections
c = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/pmc_dec/sq_counter_collection.csv")):
    if "mlp_fwd" in r["Kernel_Name"]:
        c[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open("gpurun_out/pmc_dec/sq_kernel_trace.csv")) if "mlp_fwd" in r["Kernel_Name"]]
g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8
m = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
d = sum(dur) / len(dur)
print(f"kernel {d/1e6:.3f} ms  cycles {g:.0f}  clock {g/d:.3f} GHz  mfma_busy {m/(g*1024):.3f}")
PY
