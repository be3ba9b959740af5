# Round-end measurement session -> gpurun_out/final/ (copied to profiles/ by hand): bench lines, rocprofv3 kernel traces of the
# bench, of BASELINE configs 2/3/5, of the per-chunk forward loop and of the SDF value+gradient op, PMC passes of the SDF ring
# (value pass) and backward kernels, the deferred runner loop.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
O=gpurun_out/final
mkdir -p $O
python bench.py > $O/${TAG}_bench_f16x3.json 2> $O/bench.err
python bench.py --vis-precision fp32 --no-cpu-baseline --steps 2 > $O/${TAG}_bench_fp32.json 2>> $O/bench.err
python bench.py --vis octree --steps 2 --no-cpu-baseline > $O/${TAG}_bench_octree_vis.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/p -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact > $O/p.log 2>&1
python tools/rocpd_summary.py $(find $O/p -name '*.db' | head -1) > $O/${TAG}_bench_f16x3_kernel_stats.md; rm -rf $O/p
rocprofv3 --kernel-trace --stats -d $O/p -o trace -- python bench.py --vis octree --steps 2 --warmup 1 --no-cpu-baseline > $O/p.log 2>&1
python tools/rocpd_summary.py $(find $O/p -name '*.db' | head -1) > $O/${TAG}_bench_octree_vis_kernel_stats.md; rm -rf $O/p
python tools/bench_configs.py 1 2 3 5 2>/dev/null | grep "^config" > $O/${TAG}_config_rates.txt
for C in 2 3 5; do
  RB_CONFIG_REPS=1 rocprofv3 --kernel-trace --stats -d $O/p -o trace -- python tools/bench_configs.py $C > $O/p.log 2>&1
  python tools/rocpd_summary.py $(find $O/p -name "*.db" | head -1) > $O/${TAG}_config${C}_kernel_stats.md; rm -rf $O/p
done
rocprofv3 --kernel-trace --stats -d $O/p -o trace -- python tools/prof_perchunk.py > $O/${TAG}_perchunk.log 2>&1
python tools/rocpd_summary.py $(find $O/p -name '*.db' | head -1) > $O/${TAG}_perchunk_kernel_stats.md; rm -rf $O/p
python tools/prof_perchunk.py 2>/dev/null | grep per-chunk > $O/${TAG}_perchunk_rate.txt
python tools/prof_deferred.py 1024 128 2>/dev/null | grep -v "BOX\|boxes" > $O/${TAG}_deferred_rates.txt
rocprofv3 --kernel-trace --stats -d $O/p -o trace -- python tools/prof_sdf_grad.py > $O/${TAG}_sdf_grad_rate.txt 2>/dev/null
python tools/rocpd_summary.py $(find $O/p -name '*.db' | head -1) > $O/${TAG}_sdf_grad_kernel_stats.md; rm -rf $O/p
: > $O/${TAG}_sdf_ring_pmc.md
for CS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  RB_CONFIG_REPS=1 rocprofv3 --pmc $CS --kernel-trace -d $O/p -o p -- python tools/bench_configs.py 2 > $O/p.log 2>&1
  python tools/rocpd_summary.py $(find $O/p -name "*.db" | head -1) | grep -i "sdf_ring<5>\|sdf_back" >> $O/${TAG}_sdf_ring_pmc.md; rm -rf $O/p
done
rm -f $O/p.log
cat $O/${TAG}_config_rates.txt $O/${TAG}_perchunk_rate.txt $O/${TAG}_deferred_rates.txt; grep "mode:" $O/${TAG}_sdf_grad_rate.txt; tail -c 400 $O/${TAG}_bench_f16x3.json
