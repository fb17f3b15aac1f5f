# A/B of library variants on the default workload inside ONE gpurun call (same device): bash tools/ab_libs2.sh name1 name2 ...
# ("base" = libscann_hip.so, else libscann_hip_<name>.so); extra env through ENVX="A=1 B=2"
for v in "$@"; do
  if [ $v = base ]; then unset SCANN_HIP_LIB; else export SCANN_HIP_LIB=$PWD/scann_rust_amd/libscann_hip_$v.so; fi
  for rep in 1 2; do
  env $ENVX SCANN_BENCH_STREAMS=${STREAMS:-1} timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-batch-sweep --no-recall --steps 300 $BARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %9.0f QPS  %.3f ms/step  %s %.4f ms' % ('$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms']))" || exit 1
  done
done
