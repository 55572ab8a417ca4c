#!/bin/bash
# where the training chains' 8-bit image stores stall: SQ -> TA FIFO-full counters and TA / TCP stall cycles of the
# stream kernels (separate --pmc passes), summary -> gpurun_out/r02/store_path.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/sp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/r02/sp -o sq -- python3 tools/time_decoder.py > gpurun_out/r02/sp/sq.log 2>&1
# (a second pass with TA_* / TCP_* derived counters made rocprofv3 abort with signal 6 and hang on this pool: not collected)
python3 - <<'PY' > gpurun_out/r02/store_path.txt
import csv, collections, statistics, glob
for tag in ("sq",):
    files = glob.glob(f"gpurun_out/r02/sp/{tag}_counter_collection.csv")
    if not files:
        print(tag, "no counter file"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"]
        key = "fwd_train" if "mlp_fwd_stream_kernel<true>" in name else "infer16" if "stream16" in name else "dgrad" if "mlp_bwd_stream" in name else "wgrad" if "mlp_wgrad_kernel" in name else None
        if key: per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in per.items():
        print(tag, k, {c: statistics.median(v) for c, v in d.items()})
PY
rm -f gpurun_out/r02/sp/*_counter_collection.csv gpurun_out/r02/sp/*_kernel_trace.csv
cat gpurun_out/r02/store_path.txt
