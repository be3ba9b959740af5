cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/s4_all_exact.log 2>&1; echo "exact rc $?"
ROBIR_PRECISION=split timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/s4_all_split.log 2>&1; echo "split rc $?"
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/s4_bench.json 2> gpurun_out/s4_bench.err; echo "bench rc $?"
tail -n 6 gpurun_out/s4_all_exact.log; tail -n 6 gpurun_out/s4_all_split.log; tail -n 3 gpurun_out/s4_bench.err
