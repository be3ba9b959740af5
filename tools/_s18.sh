cd /root/repo
for t in sx_base sx_bias1 sx_nosb sx_bs1db2 sx_bs1db3 sx_bs2db1b sx_base sx_bias1; do
  timeout 300 python tools/ab_sdf_x6.py $t robir_amd/librobir_hip_$t.so grad 2>&1 | tail -1
done
