# small-batch path: parity tests, the README ann_benchmark table (per-query loop) and the kernel split
# of the Partitioned / brute-force per-query runs and of Tree-X-Hybrid 1M at batch 1
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small_batch or bf_small or concurrent or scann_" > gpurun_out/t_small.log 2>&1 || { tail -20 gpurun_out/t_small.log; exit 1; }
tail -2 gpurun_out/t_small.log
bash tools/ann_table.sh gpurun_out/ann_small.txt > /dev/null 2>&1 || exit 1
grep -E "^(algorithm|qps|batched_qps)" gpurun_out/ann_small.txt | paste - - - | head -4
bash tools/prof_cli.sh cli_part --algorithm partitioned --distance squared-l2 --k 10 --synthetic-train 10000 --synthetic-test 200 --dim 64 --seed 42 | grep "small_\|select_leaves\|^qps" || exit 1
bash tools/prof_cli.sh cli_bf --algorithm brute-force --distance squared-l2 --k 10 --synthetic-train 10000 --synthetic-test 200 --dim 64 --seed 42 | grep "small_\|^qps" || exit 1
bash tools/kstat_txh1.sh 2>&1 | grep -A4 "m=1000"
