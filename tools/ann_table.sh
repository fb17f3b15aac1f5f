#!/bin/bash
# The reference README's ann_benchmark example (README.md:703-716) for all four algorithms.
# Usage (GPU box): bash tools/ann_table.sh gpurun_out/ann_benchmark.txt
out=${1:-gpurun_out/ann_benchmark.txt}
: > "$out"
for alg in brute-force partitioned hashed tree-ah; do
    scann_rust_amd/host/ann_benchmark --algorithm $alg --distance squared-l2 --k 10 --synthetic-train 10000 \
        --synthetic-test 200 --dim 64 --seed 42 >> "$out" 2>&1 || exit 1
done
# a larger run where the batch path matters: 1M x 128, 1000 leaves, 2000 queries
for alg in brute-force partitioned tree-ah; do
    scann_rust_amd/host/ann_benchmark --algorithm $alg --k 10 --synthetic-train 1000000 --synthetic-test 2000 \
        --dim 128 --num-partitions 1000 --partitions-to-search 20 --num-blocks 16 >> "$out" 2>&1 || exit 1
done
grep -E "^(dataset|algorithm|build_seconds|search_seconds|qps|recall|batched_qps)" "$out"
