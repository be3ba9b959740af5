cd $GRAFT_REPO_ROOT
for t in base nobar base nobar; do timeout 200 python tools/ab_dvis.py $t robir_amd/librobir_hip_$t.so 32 2>/dev/null | tail -1; done
ROBIR_PRECISION=split timeout 900 python bench.py --gpus 8 --steps 1 --warmup 0 --precision split 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('8 ranks on one GPU:', d['value'], d['ms_per_step'], d['n_gpus'], d['config']['collective_backend'], d['weak_views']['value'])"
