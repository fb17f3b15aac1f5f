# Round-2 measurements of the non-default workloads (run through gpurun): bench lines of brute-force
# DotProduct (C2) and Tree-X-Hybrid 1M, the Tree-X-Hybrid kernel split, the C4 / C5-shard sweeps and the
# reference README's ann_benchmark table. Usage: bash tools/prof_r02_rest.sh r02_c
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02_c}
O=gpurun_out/${TAG}_rest
mkdir -p $O
python3 bench.py --workload bf_dot > $O/${TAG}_bench_bf_dot.json 2> $O/bench_bf.err &&
python3 bench.py --workload txh --dist clustered --partitions-to-search 10 --pre-reorder-k 1000 > $O/${TAG}_bench_txh_1m.json 2> $O/bench_txh.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_txh -- python3 bench.py --workload txh --dist clustered --partitions-to-search 10 --pre-reorder-k 1000 --steps 50 --no-cpu-baseline --no-recall --no-batch-sweep > $O/ks_txh.log 2>&1 &&
cp $(find $O/ks_txh -name "*kernel_stats.csv" | head -1) $O/${TAG}_txh_1m_kernel_stats.csv
rm -rf $O/ks_txh
bash tools/ann_table.sh $O/${TAG}_ann_benchmark.txt > $O/ann.log 2>&1
timeout -k 10 400 python3 tools/sweep_txh.py --num-points 10000000 --dim 128 --S 32 --leaves 1000 --Ps 10,25,50,100 --ms 300,1000,4000,8192 --json $O/${TAG}_txh_10m_clustered_sweep.json > $O/sweep_10m.log 2>&1
timeout -k 10 400 python3 tools/sweep_txh.py --num-points 12500000 --dim 96 --S 24 --leaves 1250 --Ps 10 --ms 1000,8192 --json $O/${TAG}_txh_c5_shard_12m5x96_sweep.json > $O/sweep_c5.log 2>&1
ls -la $O
tail -5 $O/sweep_10m.log $O/sweep_c5.log
grep -E "^(algorithm|qps|recall|batched_qps)" $O/${TAG}_ann_benchmark.txt
