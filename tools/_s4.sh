python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs 2>gpurun_out/e.txt | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['headline_image']); print({k:v.get('distance_from_fp32_mfma_image') for k,v in d['legs'].items()})"
tail -3 gpurun_out/e.txt
