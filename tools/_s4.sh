timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep "^FAILED\|passed\|failed" | cut -c1-200
python bench.py --config 5 --precision exact --steps 1 --config5-chunks 125 2>/dev/null | tail -1 | cut -c1-300
