cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact > gpurun_out/prof/bench.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof -name '*.db' | head -1) --gaps > gpurun_out/prof/summary_gaps.md
find gpurun_out/prof -name '*.db' -delete
tail -40 gpurun_out/prof/summary_gaps.md
