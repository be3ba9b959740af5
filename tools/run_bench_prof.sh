# bench line + rocprofv3 kernel trace of the same command -> gpurun_out/bench_final.json, gpurun_out/prof/summary.md
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cat gpurun_out/bench_final.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof/bench.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof -name '*.db' | head -1) > gpurun_out/prof/summary.md
find gpurun_out/prof -name '*.db' -delete
head -28 gpurun_out/prof/summary.md
