cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_octree_vis_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python tools/ab_ovis.py 625 2>/dev/null | grep compact
