# Round-6 measurement session -> gpurun_out/final/ (copied to profiles/ by hand).  The default (exact) policy only -- the split policy left the
# per-round legs (VERDICT r5 task 8) -- plus the labelled f16 throughput mode, whose CESR nets got their one-product kernel this round.
# usage: bash tools/run_r06_profiles.sh [tag]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
O=gpurun_out/final
mkdir -p $O
prof() {  # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/p -o trace -- "$@" > $O/p.log 2>&1
  python tools/rocpd_summary.py $(find $O/p -name '*.db' | head -1) > $O/${TAG}_${name}_kernel_stats.md; rm -rf $O/p
}
pmc() {   # name, grep pattern, command... (two SQ passes + FETCH_SIZE + WRITE_SIZE, each its own run: --pmc with --kernel-trace only)
  local name=$1 pat=$2; shift 2
  : > $O/${TAG}_${name}_pmc.md
  for CS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    rocprofv3 --pmc $CS --kernel-trace -d $O/p -o p -- "$@" > $O/p.log 2>&1
    python tools/rocpd_summary.py $(find $O/p -name "*.db" | head -1) | grep -i "$pat" >> $O/${TAG}_${name}_pmc.md; rm -rf $O/p
  done
}
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/bench.err
python bench.py --vis octree --steps 3 --no-legs --no-configs --no-cpu-baseline > $O/${TAG}_bench_octree_vis.json 2>> $O/bench.err
python bench.py --scene nonconvex --steps 3 --no-legs --no-configs --no-cpu-baseline > $O/${TAG}_bench_nonconvex.json 2>> $O/bench.err
python bench.py --gpus 8 --steps 1 --warmup 1 > $O/${TAG}_bench_8rank_shared_gpu.json 2>> $O/bench.err
python bench.py --precision f16 --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_f16.json 2>> $O/bench.err
python bench.py --config 5 --precision f16 --steps 3 > $O/${TAG}_config5_f16.json 2>> $O/bench.err
python bench.py --config 5 --steps 2 > $O/${TAG}_config5_exact.json 2>> $O/bench.err
prof bench python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-legs --no-configs
for C in 2 3 5; do RB_CONFIG_REPS=1 prof config${C}_exact python tools/bench_configs.py $C; done
ROBIR_PRECISION=f16 RB_CONFIG_REPS=1 prof config5_f16 python tools/bench_configs.py 5
python tools/prof_perchunk.py 2>/dev/null | grep per-chunk | sed 's/^/exact policy: /' > $O/${TAG}_perchunk_rate.txt
python tools/prof_deferred.py 1024 128 2>/dev/null | grep -v "BOX\|boxes" > $O/${TAG}_deferred_rates.txt
pmc dvis_x6t "dvis_x6t" python tools/prof_dvis.py f16x6 32
ROBIR_PRECISION=f16 RB_CONFIG_REPS=1 pmc config5_f16 "cesr_f16\|dvis" python tools/bench_configs.py 5
rm -f $O/p.log
cat $O/${TAG}_perchunk_rate.txt $O/${TAG}_deferred_rates.txt; tail -c 300 $O/${TAG}_bench.json; echo; grep -c . $O/bench.err
