timeout 900 python -m pytest tests/test_mlp_gpu.py -m gpu -x -q -k "exact_operand" 2>&1 | tail -2
