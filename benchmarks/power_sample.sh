# Samples the card's clocks (rocm-smi) and, where the box lets an ordinary user read it, its power (hwmon / amd-smi) twice a
# second while bench.py runs the headline step, to check the reading that the step's GEMMs sit at the power limit rather
# than at the MFMA issue limit.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/power_sample.txt
rocm-smi --showmaxpower --showclocks > $OUT 2>&1
ls /sys/class/drm/card*/device/hwmon/*/ 2>/dev/null | head -40 >> $OUT
(amd-smi metric --power --clock 2>&1 | head -60) >> $OUT
timeout -k 10 300 python bench.py --steps 40 --warmup 3 > gpurun_out/power_sample_bench.txt 2>&1 &
BP=$!
for i in $(seq 1 60); do
  sleep 0.5
  kill -0 $BP 2>/dev/null || break
  echo "--- t=$i" >> $OUT
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk" >> $OUT
  cat /sys/class/drm/card*/device/hwmon/*/power1_average /sys/class/drm/card*/device/hwmon/*/power1_input 2>/dev/null | tr '\n' ' ' >> $OUT
  (amd-smi metric --power 2>/dev/null | grep -i -E "SOCKET_POWER|CURRENT" | tr '\n' ' ') >> $OUT
  echo >> $OUT
done
wait $BP
echo "bench rc=$?" >> $OUT
tail -1 gpurun_out/power_sample_bench.txt | cut -c1-200 >> $OUT
