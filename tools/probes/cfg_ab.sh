run() { for s in 0 1; do echo -n "streams=$s $* : "; python bench.py --no-cpu-baseline --profile-steps 0 --streams $s "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['value'],1))"; done; }
run --steps 30 --warmup 5 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256
run --steps 10 --warmup 3 --net keynet
run --steps 5 --warmup 2 --size 256 --model_keypoints 30 --batch 256 --precision fp16
run --steps 20 --warmup 5 --precision fp16
run --steps 5 --warmup 2 --precision fp32
