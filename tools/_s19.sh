cd /root/repo
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) > gpurun_out/t_exact.log 2>&1
( time ROBIR_PRECISION=split timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) > gpurun_out/t_split.log 2>&1
( time ROBIR_MLP_PRECISION=fp32 timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) > gpurun_out/t_fp32.log 2>&1
( python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > gpurun_out/t_smoke.log 2>&1
tail -n 8 gpurun_out/t_*.log
