# rerank_short_kernel after a workspace growth (m = 1000 then 8192) against the fresh case (m = 8192 only):
# vector instructions and fetched bytes per launch tell more work from slower memory
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/spmc; rm -rf $O; mkdir -p $O
for tag in grow fresh; do
  ms=8192; [ $tag = grow ] && ms=1000,8192
  rocprofv3 --pmc SQ_INSTS_VALU FETCH_SIZE --output-format csv -d $O/$tag -- python3 tools/sweep_txh.py --num-points 10000000 --dim 128 --S 32 --leaves 1000 --Ps 10 --ms $ms --steps 6 > $O/$tag.log 2>&1
  grep "P=" $O/$tag.log | cut -c1-80
  python3 - "$O/$tag" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "rerank_short_kernel" in r["Kernel_Name"] or "rerank_i8_kernel" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-22:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("   %-24s %-14s last-6 mean %.4g  (n=%d)" % (k[0], k[1], sum(v[-6:]) / len(v[-6:]), len(v)))
PY
done
rm -rf $O/grow $O/fresh
