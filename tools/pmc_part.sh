#!/bin/bash
# SQ counters of leaf_exact_scan_kernel (one batch).  Usage: bash tools/pmc_part.sh
root=$(pwd); out=$root/gpurun_out/pmc_part; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM \
   --output-format csv -d $out -o pmc -- python3 $root/tools/time_partitioned.py --reps 1 > $out/stdout.txt 2>&1
cd $root
python3 tools/pmc_summary.py $(find $out -name "*counter_collection.csv" | head -1) leaf_exact_scan
tail -2 $out/stdout.txt
