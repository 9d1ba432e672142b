#!/bin/bash
# final records of this build on one box: the driver's default command, config 2 alone, config 2 as a HIP graph, kernel stats
O=$PWD/gpurun_out; R=$PWD; T=${1:-r04i}
python bench.py --out $O/${T}_bench.json > $O/${T}_bench.line 2>$O/${T}_bench.err
C2="--no-cpu-baseline --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
python bench.py --steps 60 --warmup 16 $C2 --out $O/${T}_cfg2_bench.json > /dev/null 2>&1
python bench.py --steps 60 --warmup 16 $C2 --graph 1 --extra 0 --profile-steps 0 --out $O/${T}_cfg2_bench_graph.json > /dev/null 2>&1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --extra 0 --precision bf16 --out $O/${T}_bench_bf16.json > /dev/null 2>&1
python - <<EOF2
import json
for n in ('bench', 'bench_bf16', 'cfg2_bench', 'cfg2_bench_graph'):
    d = json.load(open('$O/${T}_%s.json' % n)); r = d.get('roofline') or {}
    print(n, round(d['value'], 1), round(d['ms_per_step'], 3), 'median', d.get('ms_per_step_median_hipevent'), 'frac', round(r.get('frac', 0), 3))
d = json.load(open('$O/${T}_bench.json'))
print(json.dumps(d.get('summary'))[:1200])
EOF2
wc -c $O/${T}_bench.line
