timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
ROBIR_PRECISION=split timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for C in 2 3 5; do python bench.py --config $C --precision split --steps 2 2>/dev/null | tail -1 | cut -c1-400; done
