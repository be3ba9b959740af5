timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_cesr_gpu.py -m gpu -q 2>&1 | grep "^FAILED\|passed\|failed" | cut -c1-200
