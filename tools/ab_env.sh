# A/B of one environment switch on the default workload inside ONE gpurun call: tools/ab_env.sh NAME v1 v2 ...
name=$1; shift
for v in "$@" "$@"; do
  export $name=$v
  echo "== $name=$v"
  timeout -k 10 200 python3 tools/sweep_mfma.py 2>/dev/null | tail -1 || exit 1
  bash tools/kstat.sh x || exit 1
done
