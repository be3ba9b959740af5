timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/prof_perchunk.py 2>/dev/null | grep per-chunk
for C in 1 2 3; do python bench.py --config $C --precision exact --steps 2 2>/dev/null | tail -1 | cut -c1-300; done
python bench.py --config 5 --precision exact --steps 1 --config5-chunks 125 2>/dev/null | tail -1 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
