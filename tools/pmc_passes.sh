#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_passes.sh <tag> <kernel-substring> <script + args...>
# The counter passes that decide what binds a kernel: SQ wave-cycle decomposition, per-class issue
# cycles, instruction counts, texture-addresser / data-return / L1 busy and stall cycles, LDS cycles.  One rocprofv3 --pmc
# pass per line (8 SQ slots per pass; never combined with trace domains other than --kernel-trace).
tag=$1; shift
kern=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
: > $out/summary.txt
i=0
while read -r ctrs; do
  [ -z "$ctrs" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/raw$i -- python3 "$@" > $out/run$i.log 2>&1
  echo "pass $i rc=$? : $ctrs" >> $out/summary.txt
  f=$(find $out/raw$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$kern" >> $out/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row.get("Kernel_Name", "")
        if sys.argv[2] not in k: continue
        acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
for k, (s, n) in sorted(acc.items()):
    print("  %-36s avg_per_launch %18.1f  launches %d" % (k, s / max(n, 1), n))
PY
  rm -rf $out/raw$i
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_IFETCH TA_BUSY_avr TA_BUSY_max
LIST
cat $out/summary.txt
