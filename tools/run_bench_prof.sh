set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof4
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -3 gpurun_out/bench_final.err
cat gpurun_out/bench_final.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof4 -o r01d -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof4/bench.log 2>&1
ls -R gpurun_out/prof4 | head
python tools/rocpd_summary.py gpurun_out/prof4/*/r01d_results.db > gpurun_out/prof4/summary.md 2>&1 || python tools/rocpd_summary.py $(find gpurun_out/prof4 -name '*.db' | head -1) > gpurun_out/prof4/summary.md
head -30 gpurun_out/prof4/summary.md
