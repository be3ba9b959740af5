timeout 900 python -m pytest tests/test_renderer_gpu.py -m gpu -q 2>&1 | grep "^FAILED\|passed\|failed\|Error" | cut -c1-250
