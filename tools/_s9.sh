cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_mlp_gpu.py tests/test_octree_gpu.py -x -q -m gpu > gpurun_out/s9_t.log 2>&1; echo "tests rc $?"; tail -n 2 gpurun_out/s9_t.log
for c in 1 2 3; do timeout 300 python bench.py --config $c --precision exact --steps 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['metric'][:40], '%.4g'%d['value'], '%.2f ms'%d['ms_per_step'], '%.2f'%d['roofline']['frac'])"; done
