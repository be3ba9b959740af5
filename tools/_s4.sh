export AB_PRECISION=f16x6
for r in 1 2; do
python tools/ab_dvis.py blk robir_amd/librobir_hip.so 64 2>&1 | tail -2
python tools/ab_dvis.py single robir_amd/librobir_hip_x6single.so 64 2>&1 | tail -2
done
python tools/ab_dvis.py compare blk single
