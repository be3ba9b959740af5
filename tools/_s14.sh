cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python bench.py --vis octree --steps 3 --warmup 1 --no-cpu-baseline --no-legs --no-configs --precision split > gpurun_out/s14_octree_split.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/s14_octree_split.json')); print('split', d['value'], d['ms_per_step'], d['roofline'])"
timeout 600 python bench.py --vis octree --steps 3 --warmup 1 --no-cpu-baseline --no-legs --no-configs > gpurun_out/s14_octree_exact.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/s14_octree_exact.json')); print('exact', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
