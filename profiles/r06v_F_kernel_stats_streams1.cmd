steps profiled: 4 (+2 warm-up); command: bench.py --steps 4 --warmup 2 --profile-steps 0 --streams 1 --no-cpu-baseline --extra 0 
