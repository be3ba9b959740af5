timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep "^FAILED\|passed\|failed" | cut -c1-200
ROBIR_PRECISION=split timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep "^FAILED\|passed\|failed" | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
