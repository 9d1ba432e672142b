#!/bin/bash
# stream-role A/Bs on the final config-2 kernel mix
mkdir -p gpurun_out/r04k
C2="--steps 40 --warmup 8 --no-cpu-baseline --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
run() { env $1 python bench.py $C2 --out gpurun_out/r04k/role_$2.json > /dev/null 2>&1; python - <<PY
import json
d = json.load(open('gpurun_out/r04k/role_$2.json'))
print('$2', round(d['value']), round(d['ms_per_step'], 4), 'median', round(d['ms_per_step_median_hipevent'], 4))
PY
}
for r in 1 2; do
run "A=1" base$r
run "KP_NO_ROLES=chain" nochain$r
run "KP_NO_ROLES=aux" noaux$r
run "KP_SMALL_SIDE=1" smallside$r
run "KP_NO_ROLES=wgrad" nowgrad$r
done
python bench.py $C2 --graph 1 --out gpurun_out/r04k/role_graph.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('gpurun_out/r04k/role_graph.json')); print('graph', round(d['value']), round(d['ms_per_step'],4))"
