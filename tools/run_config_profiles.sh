# rocprofv3 kernel-trace summaries of BASELINE configs 2, 3, 5 (tools/bench_configs.py) and of the per-chunk forward()
# -> gpurun_out/cfgprof/config{N}_kernel_stats.md ; usage: bash tools/run_config_profiles.sh [tag]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
mkdir -p gpurun_out/cfgprof
python tools/bench_configs.py 1 2 3 5 2>/dev/null | grep "^config" > gpurun_out/cfgprof/${TAG}_config_rates.txt
cat gpurun_out/cfgprof/${TAG}_config_rates.txt
for C in 2 3 5; do
  RB_CONFIG_REPS=1 rocprofv3 --kernel-trace --stats -d gpurun_out/cfgprof/c$C -o trace -- python tools/bench_configs.py $C > gpurun_out/cfgprof/c$C.log 2>&1
  python tools/rocpd_summary.py $(find gpurun_out/cfgprof/c$C -name "*.db" | head -1) > gpurun_out/cfgprof/${TAG}_config${C}_kernel_stats.md
  rm -rf gpurun_out/cfgprof/c$C
  head -12 gpurun_out/cfgprof/${TAG}_config${C}_kernel_stats.md
done
