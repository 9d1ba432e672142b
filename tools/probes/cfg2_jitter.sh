# where does the gap between the wall mean and the per-step median of the Pong-class step come from?
P="--no-cpu-baseline --profile-steps 0 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --steps 30 --warmup 5"
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'mean', round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3), 'min', round(d['ms_per_step_min_hipevent'],3), 'max', round(d['ms_per_step_max_hipevent'],3))"; }
for r in 1 2; do
python bench.py $P 2>/dev/null | show freeze
python -c "import gc, runpy, sys; gc.disable(); sys.argv=['bench.py']+'$P'.split(); runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | show nogc
python bench.py $P --streams 0 2>/dev/null | show streams0
python bench.py $P --graph 1 2>/dev/null | show graph
done
