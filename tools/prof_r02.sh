# Regenerates the profiles/r02_* evidence of the default workload on the GPU box (run through gpurun):
# kernel-trace stats, HBM traffic PMC (FETCH_SIZE / WRITE_SIZE in separate passes), SQ counters of the
# dominant kernel, then the bench line itself (which then finds a traffic.json entry for this build).
# The program after `--` is python3 bench.py directly (no env/bash hop: gpurun's exec rule).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02_a}
KERNEL=${2:-adc_mfma_kernel}
O=gpurun_out/$TAG
mkdir -p $O
B="--no-cpu-baseline --no-recall --no-batch-sweep"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 50 $B > $O/ks.log 2>&1 &&
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/${TAG}_ah_kernel_stats.csv &&
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 3 --warmup 1 $B > $O/pmc_$c.log 2>&1 &&
  grep -E "Counter_Name|$KERNEL" $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) > $O/${TAG}_pmc_ah_$c.csv
done &&
python3 tools/make_traffic.py ah $KERNEL $O/${TAG}_pmc_ah_FETCH_SIZE.csv $O/${TAG}_pmc_ah_WRITE_SIZE.csv &&
cp profiles/traffic.json $O/traffic.json &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 3 --warmup 1 $B > $O/pmc_sq.log 2>&1 &&
grep -E "Counter_Name|$KERNEL" $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) > $O/${TAG}_pmc_ah_sq.csv &&
python3 tools/pmc_summary.py $O/${TAG}_pmc_ah_sq.csv $KERNEL > $O/${TAG}_pmc_ah_sq_summary.txt
python3 bench.py > $O/${TAG}_bench_ah.json 2> $O/bench_ah.err
rm -rf $O/ks $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq
ls -la $O
cat $O/${TAG}_pmc_ah_sq_summary.txt
tail -c 1500 $O/${TAG}_bench_ah.json
