#!/bin/bash
# the three single-GPU workloads timed in batches of 1, 100 and 1024 queries.
out=${1:-gpurun_out/r01_f_batch_sweep.jsonl}
: > "$out"
for w in ah bf_dot "txh --dist clustered --partitions-to-search 10 --pre-reorder-k 1000"; do
  for b in 1 100 1024; do
    timeout -k 10 280 python3 bench.py --workload $w --batch $b --steps 50 --warmup 5 --no-cpu-baseline >> "$out" 2>> gpurun_out/batch_sweep.err || exit 1
  done
done
python3 - "$out" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("%-28s batch %5d  %10.0f QPS  %8.3f ms/step  recall %s  check %s" % (
        d["config"]["workload"][:28], d["config"].get("batch", 0), d["value"], d["ms_per_step"],
        d["config"].get("recall10@10"), d.get("oracle_check")))
PY
