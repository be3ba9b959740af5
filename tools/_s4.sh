ROBIR_PRECISION=split timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q -k "renderer or cesr or hooks or deferred or relight or smoke or edge" 2>&1 | tail -2
ROBIR_PRECISION=split python tools/prof_perchunk.py 2>/dev/null | grep per-chunk
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
