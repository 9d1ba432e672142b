steps profiled: 2 (+2 warm-up); command: bench.py --steps 2 --warmup 2 --profile-steps 0 --streams 1 --no-cpu-baseline --extra 0 --size 256 --model_keypoints 30 --batch 256 --precision fp16
