timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/prof_perchunk.py 2>/dev/null | grep per-chunk
