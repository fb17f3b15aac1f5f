# SQ counters of one kernel of the default workload, in passes of <= 8 counters
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-p}
KERNEL=${2:-adc_mfma_kernel}
O=gpurun_out/$TAG
mkdir -p $O
B="--no-cpu-baseline --no-recall --no-batch-sweep"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$i -- python3 bench.py --steps 3 --warmup 1 $B > $O/pmc_$i.log 2>&1
  f=$(find $O/pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|$KERNEL" $f > $O/${TAG}_pmc_$i.csv && python3 tools/pmc_summary.py $O/${TAG}_pmc_$i.csv $KERNEL
  rm -rf $O/pmc_$i
done
