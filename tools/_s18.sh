cd /root/repo
for t in sx_base sx_noread sx_nodma sx_nobar sx_noepi sx_nord_nodma sx_none sx_base; do
  timeout 300 python tools/ab_sdf_x6.py $t robir_amd/librobir_hip_$t.so 2>&1 | tail -1
done
