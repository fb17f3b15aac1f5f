#!/bin/bash
# Single-query latency of the wide few-query pipeline against the pipelines it replaces (SCANN_HIP_WIDE=0), one caller
# stream: flat hasher 1M x 128 at m = 5000 and Tree-X-Hybrid 1M x 128 (clustered, P = 10, m = 1000), batches 1, 2, 4.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SCANN_BENCH_STREAMS=1
for wl in "ah" "txh --dist clustered --partitions-to-search 10 --pre-reorder-k 1000"; do
  for b in ${BATCHES:-1 4}; do
    for wide in ${WIDES:-1 0}; do
      r=$(SCANN_HIP_WIDE=$wide timeout -k 10 200 python3 bench.py --workload $wl --batch $b --steps 300 --warmup 20 --no-cpu-baseline --no-recall --no-batch-sweep 2>>gpurun_out/kstat_wide.err | grep -o '"ms_per_step": [0-9.]*') || exit 1
      echo "$wl batch $b wide=$wide $r"
    done
  done
done
for wl in "ah" "txh --dist clustered --partitions-to-search 10 --pre-reorder-k 1000"; do
O=gpurun_out/ksw; rm -rf $O; mkdir -p $O
SCANN_HIP_WIDE=${PROFWIDE:-1} rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --workload $wl --batch 1 --steps 200 --no-cpu-baseline --no-recall --no-batch-sweep > $O/ks.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
f=glob.glob("$O/ks/**/*kernel_stats.csv", recursive=True)[0]
print("== $wl batch 1")
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) >= 200:
        print("   %-28s x%-3d %6.1f us" % (r["Name"].split("(")[0].split("::")[-1][:28], int(r["Calls"]) // 200, float(r["AverageNs"]) / 1e3))
PY
done
