timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for C in 2 3; do python bench.py --config $C --precision exact --steps 2 2>/dev/null | tail -1 | cut -c1-330; done
