#!/bin/bash
# kernel sequence of one config-2 / model-F step (to locate stray copies): bash tools/probes/seq.sh [F]
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/sq
ARGS="--model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
[ "$1" = F ] && ARGS="--batch 128"
rocprofv3 --kernel-trace --output-format csv -d /tmp/sq -o k -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --profile-steps 0 --streams 0 --extra 0 --precision bf16 $ARGS > /dev/null 2>&1
python - <<'PY'
import csv, glob, re
f = glob.glob('/tmp/sq/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [re.sub(r'\(anonymous namespace\)::|void |\(.*|v4::', '', r['Kernel_Name'])[:40] for r in rows]
# last step: from the last l2_loss backwards to previous adam
last = max(i for i, n in enumerate(names) if n.startswith('adam_k'))
prev = max(i for i, n in enumerate(names[:last - 5]) if n.startswith('adam_k'))
seq = names[prev + 1:last + 1]
print(len(seq), 'launches in the step')
for i, n in enumerate(seq):
    if 'copyBuffer' in n or 'Fill' in n or 'elementwise' in n:
        print(i, '...', seq[max(0, i - 2):i], '->', n, '->', seq[i + 1:i + 3])
PY
