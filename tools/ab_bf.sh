# A/B of library builds on the brute-force DotProduct workload (C2) inside one gpurun call
for v in "$@"; do
  export SCANN_HIP_LIB=$GRAFT_REPO_ROOT/scann_rust_amd/libscann_hip$v.so
  timeout -k 10 200 python3 bench.py --workload bf_dot --steps 50 --no-cpu-baseline --no-batch-sweep --no-recall 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'QPS %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], d['roofline']['kernel'], '%.3f ms' % d['roofline']['kernel_ms'], 'check', d.get('oracle_check'))" || exit 1
done
