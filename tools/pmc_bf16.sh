#!/bin/bash
# SQ counters of bf_bf16_kernel in the bf_dot bench.  Usage: bash tools/pmc_bf16.sh
root=$(pwd); out=$root/gpurun_out/pmc_bf16; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT \
   --output-format csv -d $out -o pmc -- python3 $root/bench.py --workload bf_dot --steps 3 --warmup 1 --no-cpu-baseline --no-recall > $out/stdout.txt 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE \
   --output-format csv -d $out -o pmc2 -- python3 $root/bench.py --workload bf_dot --steps 3 --warmup 1 --no-cpu-baseline --no-recall > $out/stdout2.txt 2>&1
cd $root
for f in $(find $out -name "*counter_collection.csv"); do python3 tools/pmc_summary.py $f bf_bf16_kernel; done
