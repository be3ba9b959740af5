python tools/ab_sdf.py base robir_amd/librobir_hip.so 2>&1 | grep points
python tools/ab_sdf.py okcred robir_amd/librobir_hip_okcred.so 2>&1 | grep "points\|checksums"
python tools/ab_sdf.py base robir_amd/librobir_hip.so 2>&1 | grep "points\|checksums"
