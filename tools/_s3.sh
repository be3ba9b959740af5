cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for t in base vform voff; do timeout 200 python tools/ab_dvis.py $t robir_amd/librobir_hip_$t.so 32 2>/dev/null | tail -1; done
for t in nounscale all3; do ROBIR_X6_SCALE_LOG2=0 timeout 200 python tools/ab_dvis.py $t robir_amd/librobir_hip_$t.so 32 2>/dev/null | tail -1; done
done
for t in vform voff nounscale all3; do python tools/ab_dvis.py compare base $t; done
timeout 300 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu -k "reverse_mode_for_large" 2>&1 | tail -2
