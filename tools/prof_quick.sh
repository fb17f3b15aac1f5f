# quick per-kernel picture of the default workload: kernel-trace stats + SQ counters of one kernel
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-q}
KERNEL=${2:-adc_mfma_kernel}
O=gpurun_out/$TAG
mkdir -p $O
B="--no-cpu-baseline --no-recall --no-batch-sweep"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 50 $B > $O/ks.log 2>&1 &&
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 3 --warmup 1 $B > $O/pmc_sq.log 2>&1 &&
grep -E "Counter_Name|$KERNEL" $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) > $O/${TAG}_pmc_sq.csv &&
python3 tools/pmc_summary.py $O/${TAG}_pmc_sq.csv $KERNEL > $O/${TAG}_pmc_sq_summary.txt
rm -rf $O/ks $O/pmc_sq
cut -c1-120 $O/${TAG}_kernel_stats.csv | head -14
cat $O/${TAG}_pmc_sq_summary.txt
