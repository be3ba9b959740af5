timeout 600 python -m pytest tests/test_octree_gpu.py -m gpu -x -q 2>&1 | tail -2
for l in 1 4 16; do echo LPR $l; ROBIR_CAST_LPR=$l python tools/ab_cast.py 2>&1 | grep "True"; done
