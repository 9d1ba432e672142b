#!/bin/bash
# Socket power / clocks / cap while the bench loop runs (rocm-smi sampled every ~0.2 s): tools/power_probe.sh > profiles/<tag>_power.txt
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | head -12
echo "--- samples while bench.py --steps 300 runs (bf16, side streams on) ---"
python bench.py --steps 300 --warmup 5 --no-cpu-baseline --profile-steps 0 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 6
for i in $(seq 1 25); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket|sclk" | tr '\n' ' ' | sed -E 's/ +/ /g'; echo
  sleep 0.2
done
wait $BP
tail -1 /tmp/b.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['ms_per_step'],3), 'ms/step')"
