# bench lines (default split precision, exact fp32) + rocprofv3 kernel trace of the default -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cat gpurun_out/bench_final.json
python bench.py --vis-precision fp32 --no-cpu-baseline > gpurun_out/bench_fp32.json 2> gpurun_out/bench_fp32.err
cat gpurun_out/bench_fp32.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact > gpurun_out/prof/bench.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof -name '*.db' | head -1) > gpurun_out/prof/summary.md
find gpurun_out/prof -name '*.db' -delete
rocprofv3 --kernel-trace --stats -d gpurun_out/prof32 -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact --vis-precision fp32 > gpurun_out/prof/bench32.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof32 -name '*.db' | head -1) > gpurun_out/prof/summary_fp32.md
rm -rf gpurun_out/prof32
head -24 gpurun_out/prof/summary.md
python tools/bench_configs.py 2>&1 | grep -E "^config"
