set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01f
mkdir -p $O
for w in ah bf_dot; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -- python3 bench.py --workload $w --steps 20 --no-cpu-baseline --no-recall > $O/ks_$w.log 2>&1
  cp $(find $O/ks_$w -name "*kernel_stats.csv" | head -1) $O/r01_f_${w}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${w}_$c -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-recall > $O/pmc_${w}_$c.log 2>&1
    cp $(find $O/pmc_${w}_$c -name "*counter_collection.csv" | head -1) $O/r01_f_pmc_${w}_$c.csv
    python3 tools/pmc_summary.py $O/r01_f_pmc_${w}_$c.csv $( [ $w = ah ] && echo adc_scan_res_kernel || echo bf_bf16_kernel )
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_txh -- python3 bench.py --workload txh --dist clustered --partitions-to-search 10 --pre-reorder-k 1000 --steps 20 --no-cpu-baseline --no-recall > $O/ks_txh.log 2>&1
cp $(find $O/ks_txh -name "*kernel_stats.csv" | head -1) $O/r01_f_txh_1m_kernel_stats.csv
python3 bench.py > $O/r01_f_bench_ah.json 2> $O/bench_ah.err
python3 bench.py --workload bf_dot > $O/r01_f_bench_bf_dot.json 2> $O/bench_bf.err
python3 bench.py --workload txh --dist clustered --partitions-to-search 10 --pre-reorder-k 1000 > $O/r01_f_bench_txh_1m.json 2> $O/bench_txh.err
rm -rf $O/ks_* $O/pmc_*
ls -la $O
tail -c 600 $O/r01_f_bench_ah.json
