#!/bin/bash
# The labelled f16 throughput mode: bench line, rocprofv3 kernel trace of the same command, PMC passes of the point-block kernel
# (separate passes per counter group, kernel trace only) -> gpurun_out/f16/ (copied to profiles/r05_bench_f16*.{json,md}, r05_dvis_f16p_pmc.md)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f16
timeout 900 python bench.py --precision f16 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/f16/bench_f16.json 2> gpurun_out/f16/bench_f16.err
tail -1 gpurun_out/f16/bench_f16.json | cut -c1-1500
rocprofv3 --kernel-trace --stats -d gpurun_out/f16/prof -o trace -- python bench.py --precision f16 --steps 2 --warmup 1 --no-cpu-baseline --no-legs --no-configs > gpurun_out/f16/prof.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/f16/prof -name '*.db' | head -1) > gpurun_out/f16/kernel_stats.md
rm -rf gpurun_out/f16/prof
head -14 gpurun_out/f16/kernel_stats.md
P=f16x1; N=64
run() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/f16/$name -o p -- python tools/prof_dvis.py $P $N > gpurun_out/f16/$name.log 2>&1
  python tools/rocpd_summary.py $(find gpurun_out/f16/$name -name "*.db" | head -1) | grep -i "dvis_f16p\|dvis_pb" >> gpurun_out/f16/pmc.md
  rm -rf gpurun_out/f16/$name
}
: > gpurun_out/f16/pmc.md
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
run b SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU
run c FETCH_SIZE
run d WRITE_SIZE
run g GRBM_GUI_ACTIVE
cat gpurun_out/f16/pmc.md
