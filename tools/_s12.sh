cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python tools/ab_ovis.py 625 2>/dev/null | grep compact
timeout 900 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -k "config5" > gpurun_out/s12_c5.log 2>&1; echo "c5 rc $?"; tail -n 3 gpurun_out/s12_c5.log
