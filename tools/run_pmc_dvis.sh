# PMC passes (separate runs, kernel trace only -- no sys/hip trace flags) for the fused light-visibility kernel.
# usage: bash tools/run_pmc_dvis.sh [precision] [chunks]   -> gpurun_out/pmc/summary.md
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
P=${1:-f16x3-v2}; N=${2:-16}
mkdir -p gpurun_out/pmc
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/pmc/$name -o p -- python tools/prof_dvis.py $P $N > gpurun_out/pmc/$name.log 2>&1
  python tools/rocpd_summary.py $(find gpurun_out/pmc/$name -name "*.db" | head -1) | grep -i "dvis" >> gpurun_out/pmc/summary.md
  rm -rf gpurun_out/pmc/$name
}
: > gpurun_out/pmc/summary.md
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
run b SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU
run c FETCH_SIZE
run d WRITE_SIZE
run e TCC_HIT_sum TCC_MISS_sum
tail -1 gpurun_out/pmc/a.log; cat gpurun_out/pmc/summary.md
