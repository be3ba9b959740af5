cd $GRAFT_REPO_ROOT
for t in base nobar nodma nolds noval nodmalds allabl base; do timeout 120 python tools/ab_sdf.py $t robir_amd/librobir_hip_$t.so 2>/dev/null | tail -1; done
