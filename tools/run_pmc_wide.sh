# PMC passes (separate runs, kernel trace only) for the chunk-stream colour / 512-wide kernels.
# usage: bash tools/run_pmc_wide.sh [which=decoder] [lib]   -> gpurun_out/pmcw/summary.md
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
W=${1:-decoder}; L=${2:-robir_amd/librobir_hip.so}
mkdir -p gpurun_out/pmcw
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/pmcw/$name -o p -- python tools/ab_wide.py pmc $L $W > gpurun_out/pmcw/$name.log 2>&1
  python tools/rocpd_summary.py $(find gpurun_out/pmcw/$name -name "*.db" | head -1) | grep -i "ring" >> gpurun_out/pmcw/summary.md
  rm -rf gpurun_out/pmcw/$name
}
: > gpurun_out/pmcw/summary.md
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
run b SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU
run c SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT
run d GRBM_GUI_ACTIVE
tail -1 gpurun_out/pmcw/a.log; cat gpurun_out/pmcw/summary.md
