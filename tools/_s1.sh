cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_sg_gpu.py -x -q -m gpu > gpurun_out/s1_sg.log 2>&1; echo "sg rc $?" 
timeout 300 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu -k "nan or value_grad" > gpurun_out/s1_nan.log 2>&1; echo "nan rc $?"
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err; echo "bench rc $?"
timeout 300 tools/ubench/mfma_power.bin > gpurun_out/s1_mfma_power.txt 2>&1; echo "ubench rc $?"
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/s1_all.log 2>&1; echo "all rc $?"
tail -3 gpurun_out/s1_sg.log gpurun_out/s1_nan.log gpurun_out/s1_all.log; cat gpurun_out/s1_mfma_power.txt; head -c 1500 gpurun_out/s1_bench.json; tail -5 gpurun_out/s1_bench.err
