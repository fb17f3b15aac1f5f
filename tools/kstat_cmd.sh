# kernel split (rocprofv3 --kernel-trace --stats) of one bench.py command line: tools/kstat_cmd.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
O=gpurun_out/ks_$tag; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py "$@" --steps 30 --no-cpu-baseline --no-recall --no-batch-sweep > $O/ks.log 2>&1 || { tail -5 $O/ks.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$O/ks/**/*kernel_stats.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 28 <= int(r["Calls"]) <= 80]
rows.sort(key=lambda r:-float(r["AverageNs"]))
print("== $tag")
for r in rows[:16]:
    print("  %-44s %8.1f us x %s" % (r["Name"].split("(")[0].replace("void ","").replace("scann::","")[:44], float(r["AverageNs"])/1e3, r["Calls"]))
PY
tail -c 300 $O/ks.log | tr ',' '\n' | grep -E '"value"|ms_per_step' 
rm -rf $O/ks
