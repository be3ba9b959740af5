cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/s17_all_exact.log 2>&1; echo "exact rc $?"
ROBIR_PRECISION=split timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/s17_all_split.log 2>&1; echo "split rc $?"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s17_smoke.log 2>&1; echo "smoke rc $?"
tail -n 4 gpurun_out/s17_all_exact.log; tail -n 4 gpurun_out/s17_all_split.log; tail -n 1 gpurun_out/s17_smoke.log
