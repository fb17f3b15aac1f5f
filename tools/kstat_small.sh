# per-kernel averages at batch 1 for the AH and TXH workloads
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wl in "ah" "txh --partitions-to-search 10 --pre-reorder-k 1000"; do
O=gpurun_out/kss; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --workload $wl --batch ${1:-1} --steps 200 --no-cpu-baseline --no-recall --no-batch-sweep > $O/ks.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/ks/**/*kernel_stats.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if int(r["Calls"])>=200]
tot=sum(float(r["AverageNs"])*int(r["Calls"])/205 for r in rows)
print("== $wl batch ${1:-1}: sum of kernel time per step %.1f us, %d kernels" % (tot/1e3, len(rows)))
for r in rows: print("   %-40s x%-4s %7.1f us" % (r["Name"].split("(")[0].split("::")[-1][:40], int(r["Calls"])//205, float(r["AverageNs"])/1e3))
PY
tail -c 600 $O/ks.log | grep -o '"ms_per_step": [0-9.]*'
done
