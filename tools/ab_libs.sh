# A/B of library builds on the default workload inside ONE gpurun call (device-to-device variance is
# larger than most kernel changes): tools/ab_libs.sh "" _prev ""  -> step time + kernel split per build
for v in "$@"; do
  export SCANN_HIP_LIB=$GRAFT_REPO_ROOT/scann_rust_amd/libscann_hip$v.so
  timeout -k 10 200 python3 tools/sweep_mfma.py SCANN_HIP_MFMA=2 2>/dev/null | tail -1 || exit 1
  bash tools/kstat.sh x$v || exit 1
done
