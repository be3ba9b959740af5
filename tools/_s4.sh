timeout 1500 python -m pytest tests/test_mlp_gpu.py tests/test_edge_cases_gpu.py tests/test_renderer_gpu.py -m gpu -q 2>&1 | grep "^FAILED\|passed\|failed" | cut -c1-200
ROBIR_PRECISION=split timeout 1500 python -m pytest tests/test_mlp_gpu.py tests/test_renderer_gpu.py -m gpu -q 2>&1 | grep "^FAILED\|passed\|failed" | cut -c1-200
