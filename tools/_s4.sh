timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep "^FAILED\|passed\|failed" | cut -c1-200
ROBIR_PRECISION=split timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep "^FAILED\|passed\|failed" | cut -c1-200
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print({k:(v['value'],v['ms_per_step']) for k,v in d['legs'].items()}); print([(e['config'], round(e['ms'],1)) for e in d['configs']['exact']], [(e['config'], round(e['ms'],1)) for e in d['configs']['split']])"
