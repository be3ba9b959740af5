cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/s2_all_exact.log 2>&1; echo "exact rc $?"
ROBIR_PRECISION=split timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/s2_all_split.log 2>&1; echo "split rc $?"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s2_smoke.log 2>&1; echo "smoke rc $?"
timeout 900 bash tools/run_pmc_dvis.sh f16x6 32 > gpurun_out/s2_pmc.log 2>&1; cp gpurun_out/pmc/summary.md gpurun_out/s2_x6_pmc.md
timeout 300 python tools/prof_dvis.py variants 32 f16x6,f16x3-v2 > gpurun_out/s2_variants.log 2>&1
tail -n 4 gpurun_out/s2_all_exact.log; tail -n 4 gpurun_out/s2_all_split.log; tail -n 2 gpurun_out/s2_smoke.log; cat gpurun_out/s2_x6_pmc.md; tail -n 3 gpurun_out/s2_variants.log
