#!/bin/bash
# Round records on ONE box: tools/records.sh <tag>   -> gpurun_out/<tag>_*.json / .csv / .txt  (copy what is quoted into profiles/)
T=${1:-rXX}; O=$PWD/gpurun_out; R=$PWD
# the driver's command (default dtype fp16 since round 3; the line also carries extra.bf16 and the extra_configs legs)
python bench.py --steps 20 --warmup 5 --out $O/${T}_bench.json > /dev/null 2>$O/${T}_bench.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --extra 0 --streams 0 --out $O/${T}_bench_streams0.json > /dev/null 2>&1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --extra 0 --precision bf16 --out $O/${T}_bench_bf16.json > /dev/null 2>&1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --out $O/${T}_cfg2_bench.json > /dev/null 2>&1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --precision bf16 --fuse-bn 0 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 --out $O/${T}_cfg2_bench_nofuse.json > /dev/null 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision bf16 --net keynet --out $O/${T}_cfg3_bench.json > /dev/null 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --size 256 --model_keypoints 30 --batch 256 --precision fp16 --out $O/${T}_cfg5_bench.json > /dev/null 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision fp32 --extra 0 --out $O/${T}_fp32_bench.json > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for s in 1 0; do
  rm -rf /tmp/ks$s
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks$s -o k -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 --extra 0 --streams $s > /dev/null 2>&1
  cp $(find /tmp/ks$s -name "*kernel_stats.csv" | head -1) $O/${T}_kernel_stats_streams$s.csv
  python $R/tools/overlap_report.py $(find /tmp/ks$s -name "*kernel_trace.csv" | head -1) 0.4 > $O/${T}_overlap_streams$s.txt
done
rm -rf /tmp/k2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k2 -o k -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --profile-steps 0 --streams 0 --precision bf16 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256 > /dev/null 2>&1
cp $(find /tmp/k2 -name "*kernel_stats.csv" | head -1) $O/${T}_cfg2_kernel_stats.csv
cd $R
(python tools/bench_infer.py; python tools/bench_infer.py --batch 32; python tools/bench_infer.py --batch 256 --frames 50) 2>&1 | grep -v amdgpu.ids > $O/${T}_infer.txt
python - <<EOF
import json
for n in ('bench', 'bench_streams0', 'bench_bf16', 'cfg2_bench', 'cfg2_bench_nofuse', 'cfg3_bench', 'cfg5_bench', 'fp32_bench'):
    d = json.load(open('$O/${T}_%s.json' % n)); r = d['roofline']
    print(n, round(d['value'], 1), round(d['ms_per_step'], 3), 'frac', round(r['frac'], 3), 'conc', round(r.get('concurrent', {}).get('frac', 0), 3), 'step', round(r['step_frac'], 3))
EOF
