#!/bin/bash
# PMC passes over scripts/bench_conv.py (the 12 fat conv layers of the step): instruction mix, stall split, LDS / TA
# queue pressure of the conv kernels.  usage (on the GPU box): scripts/pmc_conv.sh <outdir>
out=${1:-gpurun_out/pmc_conv}
what=${2:-convonly}
mkdir -p $out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $root/$out/$name -- python3 $root/scripts/bench_conv.py bf16 32 $what > $root/$out/$name.log 2>&1
}
run p1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA
run p2 SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES
run p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_BUSY_CYCLES
cd $root
python3 scripts/pmc_summary.py $out/summary.json $(find $out -name "*counter_collection.csv") > $out/summary.txt
grep -E "conv_pp_kernel|conv_kernel|wgrad" $out/summary.txt
