cd $GRAFT_REPO_ROOT
for t in base regstage base regstage; do timeout 120 python tools/ab_sdf.py $t robir_amd/librobir_hip_$t.so 2>/dev/null | tail -2; done
