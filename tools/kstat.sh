# average time of the kernels of the default workload for one library build: tools/kstat.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ks_$1; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 30 --no-cpu-baseline --no-recall --no-batch-sweep > $O/ks.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/ks/**/*kernel_stats.csv", recursive=True)[0]
print("== $1", " ".join("%s %.0f" % (r["Name"].split("(")[0].split("::")[-1][:22], float(r["AverageNs"])/1e3) for r in csv.DictReader(open(f)) if float(r["AverageNs"])>20000 and "encode" not in r["Name"]))
PY
rm -rf $O
