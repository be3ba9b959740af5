# PMC passes (no sys/hip trace flags) for the fused light-visibility kernel; usage: bash tools/run_pmc_dvis.sh [precision] [chunks]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
P=${1:-f16x3}; N=${2:-16}
mkdir -p gpurun_out/pmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace -d gpurun_out/pmc/a -o p -- python tools/prof_dvis.py $P $N > gpurun_out/pmc/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_LDS_DATA_FIFO_FULL --kernel-trace -d gpurun_out/pmc/b -o p -- python tools/prof_dvis.py $P $N > gpurun_out/pmc/b.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace -d gpurun_out/pmc/c -o p -- python tools/prof_dvis.py $P $N > gpurun_out/pmc/c.log 2>&1
for d in a b c; do python tools/rocpd_summary.py $(find gpurun_out/pmc/$d -name "*.db" | head -1) | grep -i "dvis_fused" > gpurun_out/pmc/$d.md 2>&1; done
tail -2 gpurun_out/pmc/a.log; cat gpurun_out/pmc/a.md gpurun_out/pmc/b.md gpurun_out/pmc/c.md
