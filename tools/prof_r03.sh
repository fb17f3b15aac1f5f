# Round-3 evidence of the default workload (C3) on the GPU box (run through gpurun): bash tools/prof_r03.sh r03_a
#   kernel-trace stats of the default command (two caller streams: kernels of consecutive batches overlap) and of the
#   one-stream run (clean per-kernel times), HBM traffic PMC (FETCH_SIZE / WRITE_SIZE in separate passes) -> the keyed
#   profiles/traffic.json entry, SQ counters of the dominant kernel, then the bench lines (which then find the entry).
# The program after `--` is python3 bench.py directly (no env/bash hop: gpurun's exec rule).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r03_a}
KERNEL=${2:-adc_smfmac_kernel}
O=gpurun_out/$TAG
mkdir -p $O
B="--no-cpu-baseline --no-recall --no-batch-sweep"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 50 $B > $O/ks.log 2>&1 &&
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/${TAG}_ah_kernel_stats.csv
export SCANN_BENCH_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks1 -- python3 bench.py --steps 50 $B > $O/ks1.log 2>&1 &&
cp $(find $O/ks1 -name "*kernel_stats.csv" | head -1) $O/${TAG}_ah_kernel_stats_1stream.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 3 --warmup 1 $B > $O/pmc_$c.log 2>&1 &&
  grep -E "Counter_Name|$KERNEL" $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) > $O/${TAG}_pmc_ah_$c.csv
done
python3 tools/make_traffic.py ah $KERNEL $O/${TAG}_pmc_ah_FETCH_SIZE.csv $O/${TAG}_pmc_ah_WRITE_SIZE.csv &&
cp profiles/traffic.json $O/traffic.json
i=0
: > $O/${TAG}_pmc_ah_sq_summary.txt
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_sq$i -- python3 bench.py --steps 3 --warmup 1 $B > $O/pmc_sq$i.log 2>&1
  f=$(find $O/pmc_sq$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|$KERNEL" $f > $O/${TAG}_pmc_ah_sq$i.csv && python3 tools/pmc_summary.py $O/${TAG}_pmc_ah_sq$i.csv $KERNEL >> $O/${TAG}_pmc_ah_sq_summary.txt
done
python3 bench.py > $O/${TAG}_bench_ah_1stream.json 2> $O/bench_ah_1s.err
unset SCANN_BENCH_STREAMS
python3 bench.py > $O/${TAG}_bench_ah.json 2> $O/bench_ah.err
rm -rf $O/ks $O/ks1 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq1 $O/pmc_sq2
ls -la $O
cat $O/${TAG}_pmc_ah_sq_summary.txt
tail -c 1800 $O/${TAG}_bench_ah.json
