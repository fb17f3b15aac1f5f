# the -m gpu suite under every path-forcing environment (each must be green): tools/full_matrix.sh
run() {
  echo "== $*"
  env "$@" timeout -k 10 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/fm.log 2>&1 || { tail -15 gpurun_out/fm.log; exit 1; }
  tail -1 gpurun_out/fm.log
}
run SCANN_HIP_MFMA=0
run SCANN_HIP_MFMA=2
run SCANN_HIP_MFMA=3
run SCANN_HIP_RERANK_I8=2 SCANN_HIP_RERANK_I8_MIN=1
run SCANN_HIP_RERANK_I8=2 SCANN_HIP_RERANK_I8_MIN=1 SCANN_HIP_RERANK_STORE=fp8
run SCANN_HIP_SMALL=0
run SCANN_HIP_FUSED=0
