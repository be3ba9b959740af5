cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_octree_vis_gpu.py tests/test_mlp_gpu.py -x -q -m gpu > gpurun_out/s11_t.log 2>&1; echo "tests rc $?"; tail -n 3 gpurun_out/s11_t.log
timeout 600 python bench.py --vis octree --steps 2 --warmup 1 --no-cpu-baseline --no-legs --no-configs > gpurun_out/s11_octree.json 2> gpurun_out/s11_octree.err; echo "octree bench rc $?"; tail -c 900 gpurun_out/s11_octree.json
ROBIR_OVIS_COMPACT=0 timeout 600 python bench.py --vis octree --steps 2 --warmup 1 --no-cpu-baseline --no-legs --no-configs > gpurun_out/s11_octree_plain.json 2>/dev/null; echo; tail -c 600 gpurun_out/s11_octree_plain.json
timeout 900 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -k "config5" > gpurun_out/s11_c5.log 2>&1; echo "c5 rc $?"; tail -n 3 gpurun_out/s11_c5.log
