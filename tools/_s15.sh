cd $GRAFT_REPO_ROOT
for t in base nodma nolds both base; do timeout 200 python tools/ab_dvis.py $t robir_amd/librobir_hip_$t.so 32 2>/dev/null | tail -1; done
