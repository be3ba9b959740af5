timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --config 2 --precision exact --steps 2 2>/dev/null | tail -1 | cut -c1-260
