# Samples package power and shader clock (rocm-smi) while bench.py's visibility kernel runs: evidence for / against the
# power-limit reading of profiles/r02_dvis_pmc.md.  usage: bash tools/power_probe.sh  -> gpurun_out/power_probe.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/power_probe.txt
: > $O
echo "== idle" >> $O
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i "power\|sclk\|mclk" >> $O
python bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-exact > gpurun_out/power_probe_bench.json 2>/dev/null &
BP=$!
sleep 12      # model build + warm-up
echo "== under the visibility kernel (0.5 s apart)" >> $O
for i in 1 2 3 4 5 6 7 8 9 10; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | tr '\n' ' ' >> $O
  echo >> $O
  sleep 0.5
done
wait $BP
tail -c 300 gpurun_out/power_probe_bench.json >> $O
cat $O
