# config 2 (Pong-class net, 256 pairs) on one box: the library's current build, three repetitions (KP_CFG via --lib for A/B)
P="--no-cpu-baseline --profile-steps 0 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --steps 50 --warmup 10"
for rep in 1 2 3; do for l in "$@"; do echo -n "$l: "; python bench.py $P --lib $l 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['value']), d['config']['loss'])"; done; done
