cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu -k "fused or bit_identical" > gpurun_out/s10_fused.log 2>&1; echo "fused rc $?"; tail -n 3 gpurun_out/s10_fused.log
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/s10_all_exact.log 2>&1; echo "exact rc $?"
ROBIR_PRECISION=split timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/s10_all_split.log 2>&1; echo "split rc $?"
tail -n 4 gpurun_out/s10_all_exact.log; tail -n 4 gpurun_out/s10_all_split.log
