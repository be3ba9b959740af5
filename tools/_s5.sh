cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s5
export ROBIR_PRECISION=split
timeout 300 python -m pytest tests/test_nonconvex_gpu.py -q -m gpu -k forward_material 2>&1 | tail -2
RB_CONFIG_REPS=1 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/s5/p -o trace -- python tools/bench_configs.py 2 > gpurun_out/s5/c2.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/s5/p -name "*.db" | head -1) > gpurun_out/s5/config2_kernel_stats.md; rm -rf gpurun_out/s5/p
head -20 gpurun_out/s5/config2_kernel_stats.md
: > gpurun_out/s5/ring8_pmc.md
for CS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  RB_CONFIG_REPS=1 timeout 600 rocprofv3 --pmc $CS --kernel-trace -d gpurun_out/s5/p -o p -- python tools/bench_configs.py 2 > gpurun_out/s5/p.log 2>&1
  python tools/rocpd_summary.py $(find gpurun_out/s5/p -name "*.db" | head -1) | grep -i "sdf_ring8\|sdf_back\|color_mlp" >> gpurun_out/s5/ring8_pmc.md; rm -rf gpurun_out/s5/p
done
cat gpurun_out/s5/ring8_pmc.md
