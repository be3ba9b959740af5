for t in base dNOVAL dNOBAR dNOLDS dNODMA dNODMALDS dALL dDB3 base; do
  l=robir_amd/librobir_hip_$t.so; [ $t = base ] && l=robir_amd/librobir_hip.so
  python tools/ab_wide.py $t $l decoder
done
