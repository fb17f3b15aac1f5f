# C4 (10M x 128, 1000 leaves, P = 10): the default scan choice against forced prefilter forms at large pre_reorder_k,
# one process per line; then the kernel split of the default path at m = 8192 (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${MODES:-unset 2 3}; do
  if [ $v = unset ]; then unset SCANN_HIP_MFMA; else export SCANN_HIP_MFMA=$v; fi
  echo "== SCANN_HIP_MFMA=$v"
  timeout -k 10 250 python3 tools/sweep_txh.py --num-points 10000000 --leaves 1000 --Ps 10 --ms 4000,8192 --steps 40 2>&1 | grep "P=\|Error" | cut -c1-130
done
unset SCANN_HIP_MFMA
O=gpurun_out/ks_c4; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 tools/sweep_txh.py --num-points 10000000 --leaves 1000 --Ps 10 --ms 8192 --steps 40 > $O/ks.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/ks/**/*kernel_stats.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 38 <= int(r["Calls"]) <= 130]
rows.sort(key=lambda r:-float(r["AverageNs"]))
for r in rows[:16]:
    print("  %-44s %8.1f us x %s" % (r["Name"].split("(")[0].replace("void ","").replace("scann::","")[:44], float(r["AverageNs"])/1e3, r["Calls"]))
import shutil
shutil.copy(f, "gpurun_out/r03_txh_10m_m8192_kernel_stats.csv")
PY
rm -rf $O/ks
