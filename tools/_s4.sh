timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep "^FAILED\|passed\|failed" | cut -c1-200
python tools/prof_perchunk.py 2>/dev/null | grep per-chunk
