#!/bin/bash
# rocprofv3 kernel stats of one ann_benchmark run.  Usage: bash tools/prof_cli.sh <tag> <ann_benchmark args...>
tag=$1; shift
root=$(pwd)
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag -o prof -- $root/scann_rust_amd/host/ann_benchmark "$@" > $root/gpurun_out/$tag/stdout.txt 2>&1
cd $root
f=$(find gpurun_out/$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-60s calls %6s total %10.3f ms avg %9.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
          float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
grep -E "^(qps|batched_qps|recall)" gpurun_out/$tag/stdout.txt
