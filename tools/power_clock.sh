# Socket power and shader clock while the smoke training step runs back to back (rocm-smi, 0.5 s samples) -> gpurun_out/power_clock.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/power_clock.txt
( rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -iE "power|sclk|max" | head -8 ) > $O
echo "--- during python bench.py --steps 300 (0.5 s samples: average socket power W, sclk MHz)" >> $O
python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-extras > /tmp/pc_bench.log 2>&1 &
BP=$!
sleep 6
for i in $(seq 1 14); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ' ' >> $O
  echo >> $O
  sleep 0.5
done
wait $BP
tail -1 /tmp/pc_bench.log | cut -c1-200 >> $O
cat $O
