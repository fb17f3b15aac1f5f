cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/grbm; mkdir -p $O
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-recall --no-batch-sweep > $O/log 2>&1
f=$(find $O/p -name "*counter_collection.csv" | head -1)
grep -E "adc_mfma_kernel" $f | awk -F, '{print $(NF-3), $(NF-2), $NF-$(NF-1)}' | head -12
rm -rf $O/p
