#!/bin/bash
# what the box does while the default chain runs: engine / memory clocks, power and temperature sampled beside 30 s of back-to-back launches
cd $GRAFT_REPO_ROOT
rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -v "^=\|^$" | head -20
echo "---- under load"
python bench.py --no-cpu --steps 150000 --warmup 100 > gpurun_out/clock_probe_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|mclk\|fclk\|power\|Temperature (Sensor junction)\|Temperature (Sensor memory)" | tr '\n' ';' | cut -c1-600; echo
  sleep 3
done
wait $BP
python -c "import json; j=json.loads(open('gpurun_out/clock_probe_bench.json').readlines()[-1]); print('launch_us', j['roofline']['launch_us'])"
