#!/bin/bash
# where the training chains' image stores stall: SQ -> TA FIFO-full counters, then TA / TCP stall counters of the stream
# kernels.  Round 2 put all TA_* / TCP_* counters into ONE --pmc pass: rocprofv3 aborted in rocprofiler_create_counter_config
# with "error code 38: Request exceeds the capabilities of the hardware to collect" (gpurun_out/r02/sp/ta.log) -- a block has
# a fixed number of counter slots (TA: two).  Here every pass carries at most two counters of one block, the program comes
# directly after `--`, every pass has its own time limit and a failed pass ends the sequence.
# Summary -> gpurun_out/r03/store_path.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03/sp
pass() {
  local tag=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/r03/sp -o "$tag" -- python3 tools/time_decoder.py > "gpurun_out/r03/sp/$tag.log" 2>&1
  local rc=$?
  echo "$tag rc=$rc"
  return $rc
}
pass sq SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY \
 && pass ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum \
 && pass ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_COALESCED_WRITE_CYCLES_sum \
 && pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum \
 && pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
python3 - <<'PY' > gpurun_out/r03/store_path.txt
import csv, collections, statistics, glob
for tag in ("sq", "ta1", "ta2", "tcp1", "tcp2"):
    files = glob.glob(f"gpurun_out/r03/sp/{tag}_counter_collection.csv")
    if not files:
        print(tag, f"no counter file (see gpurun_out/r03/sp/{tag}.log)")
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"]
        key = ("fwd_train" if "mlp_fwd_stream_kernel<true" in name else "infer16" if "stream16" in name else "dgrad" if "mlp_bwd_stream" in name
               else "wgrad" if "mlp_wgrad_kernel" in name else None)
        if key:
            per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in per.items():
        print(tag, k, {c: statistics.median(v) for c, v in d.items()})
PY
find gpurun_out/r03/sp \( -name '*_counter_collection.csv' -o -name '*_kernel_trace.csv' \) -delete
for t in ta1 ta2 tcp1 tcp2; do echo "== $t"; tail -3 "gpurun_out/r03/sp/$t.log" | cut -c1-240; done
cat gpurun_out/r03/store_path.txt
