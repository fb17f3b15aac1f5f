cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 30 1000; do
O=gpurun_out/kss; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --workload txh --partitions-to-search 10 --pre-reorder-k $m --batch 1 --steps 200 --no-cpu-baseline --no-recall --no-batch-sweep > $O/ks.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/ks/**/*kernel_stats.csv", recursive=True)[0]
print("m=$m")
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) >= 200 and ("small_" in r["Name"] or "select_leaves" in r["Name"]):
        print("   %-28s %6.1f us" % (r["Name"].split("(")[0].split("::")[-1][:28], float(r["AverageNs"]) / 1e3))
PY
done
