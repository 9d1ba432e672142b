run() { echo -n "$* : "; env "${@:2}" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --lib $1 2>/tmp/err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['config']['loss'])"; }
for rep in 1 2; do
run build_ab/base3.so KP_WG_ORDER=1
run keypoints_amd/csrc/libkp_hip.so KP_WG_ORDER=1
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --streams 0 --lib build_ab/base3.so | tail -1 | cut -c90-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --streams 0 | tail -1 | cut -c90-200
