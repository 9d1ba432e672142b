C2="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 --extra 0"
for rep in 1 2; do for g in 0 1; do
python bench.py $C2 --graph $g 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph=$g step', round(d['ms_per_step'],3), round(d['ms_per_step_median_hipevent'],3), round(d['value']))"
done; done
for g in 0 1; do python bench.py $C2 --graph $g --streams 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams0 graph=$g step', round(d['ms_per_step'],3), round(d['value']))"; done
