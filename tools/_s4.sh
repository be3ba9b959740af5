timeout 900 python -m pytest tests/test_mlp_gpu.py -m gpu -x -q 2>&1 | tail -2
ROBIR_PRECISION=split timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
ROBIR_PRECISION=split python tools/prof_perchunk.py 2>/dev/null | grep per-chunk
timeout 900 python bench.py --config 5 --precision split --steps 1 --config5-chunks 125 2>/dev/null | tail -1 | cut -c1-330
