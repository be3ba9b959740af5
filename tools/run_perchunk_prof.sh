# kernel trace + idle-gap analysis of the per-chunk forward() path -> gpurun_out/prof_chunk/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_chunk
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_chunk -o trace -- python tools/prof_perchunk.py > gpurun_out/prof_chunk/run.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_chunk -name '*.db' | head -1) --gaps > gpurun_out/prof_chunk/summary.md
find gpurun_out/prof_chunk -name '*.db' -delete
grep "per-chunk" gpurun_out/prof_chunk/run.log
grep -A40 "one step" gpurun_out/prof_chunk/summary.md
