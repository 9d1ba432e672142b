#!/bin/bash
# Round-6 GPU run helper (one file instead of a script per run): `bash tools/r06.sh <subcommand>`; output under gpurun_out/r06*/
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
sub=${1:-}; shift || true
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
case "$sub" in
stall_env)   # config-2 host stalls under different HIP-runtime pool sizes
  o=gpurun_out/r06b; mkdir -p $o
  run() { tag=$1; shift; env "$@" python tools/cfg2_stall.py --no-f-legs --steps 300 --repeat 2 2>&1 | grep -v "$F" > $o/stall_$tag.txt; }
  run base KP_NONE=1
  run kernarg16m HSA_KERNARG_POOL_SIZE=16777216
  run signal4k ROC_SIGNAL_POOL_SIZE=4096
  run batch DEBUG_CLR_MAX_BATCH_SIZE=100000
  run cpusync DEBUG_CLR_BATCH_CPU_SYNC_SIZE=100000
  run cmdbuf GPU_MAX_COMMAND_BUFFERS=64
  ( cd /tmp && rocprofv3 --hip-runtime-trace --output-format csv -d /tmp/hiptrace -o t -- python $OLDPWD/tools/cfg2_stall.py --no-f-legs --steps 200 --repeat 1 > $OLDPWD/$o/stall_hiptrace.txt 2>&1 )
  python - <<'PY' > $o/hiptrace_slow.txt 2>&1
import csv, glob
fs = glob.glob('/tmp/hiptrace/**/*hip_api_trace.csv', recursive=True)
print(fs)
rows = []
for f in fs:
    for r in csv.DictReader(open(f)):
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        rows.append((d, r['Function'], int(r['Start_Timestamp'])))
rows.sort(reverse=True)
for d, fn, t in rows[:40]:
    print(f'{d:9.3f} ms  {fn}  @{t}')
import collections
tot = collections.Counter(); cnt = collections.Counter()
for d, fn, t in rows:
    tot[fn] += d; cnt[fn] += 1
for fn, v in tot.most_common(12):
    print(f'{fn:40s} {cnt[fn]:8d} calls {v:10.1f} ms')
PY
  ;;
stall_hunt)   # many config-2 steps: catch the rare GPU-side stall and see what the host did meanwhile; then the same under a kernel trace
  o=gpurun_out/r06c; mkdir -p $o
  python tools/cfg2_stall.py --no-f-legs --steps 1500 --repeat 3 2>&1 | grep -v "$F" > $o/hunt_base.txt
  HSA_KERNARG_POOL_SIZE=16777216 python tools/cfg2_stall.py --no-f-legs --steps 1500 --repeat 3 2>&1 | grep -v "$F" > $o/hunt_kernarg.txt
  HSA_KERNARG_POOL_SIZE=16777216 python tools/cfg2_stall.py --no-f-legs --steps 1500 --repeat 3 --reserve 1 2>&1 | grep -v "$F" > $o/hunt_kernarg_reserve.txt
  ( cd /tmp && rm -rf /tmp/kt && HSA_KERNARG_POOL_SIZE=16777216 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o t -- python $OLDPWD/tools/cfg2_stall.py --no-f-legs --steps 1500 --repeat 2 2>&1 | grep -v "$F" > $OLDPWD/$o/hunt_ktrace.txt )
  python - <<'PY' > $o/ktrace_gaps.txt 2>&1
import csv, glob
fs = glob.glob('/tmp/kt/**/*kernel_trace.csv', recursive=True)
rows = []
for f in fs:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
rows.sort()
print(len(rows), 'kernels')
end, last = 0, None
gaps = []
for i, r in enumerate(rows):
    if end and r[0] - end > 3_000_000:
        gaps.append((r[0] - end, i))
    if r[1] > end:
        end, last = r[1], i
for g, i in gaps[:30]:
    print(f'gap {g / 1e6:8.3f} ms before kernel #{i}')
    for j in range(max(0, i - 4), min(len(rows), i + 3)):
        print('     ', '>>' if j == i else '  ', f'{(rows[j][0] - rows[i][0]) / 1e6:9.3f} ms  dur {(rows[j][1] - rows[j][0]) / 1e3:8.1f} us  q{rows[j][3]} s{rows[j][4]}  {rows[j][2]}')
PY
  ;;
ab)   # alternating A/B of the model-F step on ONE box: bash tools/r06.sh ab <out> "<env A>" "<env B>" ["<env C>" ...]  (env = KP_* of keypoints_amd/_probe.py, or '-')
  o=gpurun_out/$1; mkdir -p $o; shift
  : > $o/ab.txt
  for rep in 1 2 3; do
    for e in "$@"; do
      ee=$e; [ "$e" = "-" ] && ee="KP_NONE=1"
      env $ee python bench.py --steps 20 --warmup 8 --no-cpu-baseline --profile-steps 0 --extra 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep  %-40s' % '$e', round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median_hipevent'],3), round(d['value'],1))" >> $o/ab.txt
    done
  done
  cat $o/ab.txt
  ;;
*) echo "unknown subcommand $sub"; exit 2;;
esac
