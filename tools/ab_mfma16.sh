# Tree-X-Hybrid 10M x 128: the default scan choice against forced choices, one fresh process per line
# (bench.py: 200 timed steps), inside ONE gpurun call.  SCANN_HIP_MFMA: unset = heuristic, 0 = f32 gather scan,
# 2 / 3 = the 32- / 16-column integer-MFMA prefilter.
run() {
  v=$1; P=$2; m=$3
  if [ $v = unset ]; then unset SCANN_HIP_MFMA; else export SCANN_HIP_MFMA=$v; fi
  timeout -k 10 400 python3 bench.py --workload txh --dist clustered --num-points 10000000 --leaves 1000 --partitions-to-search $P \
      --pre-reorder-k $m --no-cpu-baseline --no-batch-sweep --no-recall 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('MFMA=%-5s P=%3d m=%5d  %8.0f QPS  %.3f ms/step  %s %.3f ms' % ('$v', $P, $m, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms']))" || exit 1
}
for cfg in "10 1000" "10 8192" "25 8192"; do
  set -- $cfg
  run unset $1 $2
  run 0 $1 $2
  if [ $1 = 10 ]; then run 3 $1 $2; else run 2 $1 $2; fi
done
