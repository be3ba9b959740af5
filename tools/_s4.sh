timeout 900 python -m pytest tests/test_edge_cases_gpu.py -m gpu -x -q 2>&1 | tail -15
