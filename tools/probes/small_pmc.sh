#!/bin/bash
# SQ counters of the Pong-class band kernels on one launch shape (rocprofv3 --pmc, one pass per counter group):
#   bash tools/probes/small_pmc.sh            -> gpurun_out/small_pmc.txt
cd "$(dirname "$0")/../.." || exit 1
ROOT=$(pwd); export TMPDIR=/tmp
cat > /tmp/small_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['KP_ROOT'])
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
N, H = 256, 84; bf = torch.bfloat16; st = torch.cuda.current_stream().cuda_stream
for C, K in ((16, 32), (32, 32)):
    a = (torch.randn(N, H + 2, H + 2, C, device='cuda') * 0.5).to(bf); wf = (torch.randn(K, 3, 3, C, device='cuda') * 0.1).to(bf)
    bias = torch.zeros(K, device='cuda'); y = torch.empty(N, H, H, K, device='cuda', dtype=bf); stats = torch.empty(N * H * H // 64 + 1, 2, K, device='cuda')
    dy = torch.zeros(N, H + 4, H + 4, K, device='cuda', dtype=bf); dy[:, 2:-2, 2:-2] = torch.randn(N, H, H, K, device='cuda').to(bf)
    parts = _lib.lib().kp_conv_wgrad_partials(N, H, H, C, K, 3, 2, 1)
    scratch = torch.empty(parts, K, 3, 3, C, device='cuda')
    for _ in range(4):
        call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, H + 2, C, K, 3, 1, st)
        call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, H, C, K, 3, 2, 1, st)
    torch.cuda.synchronize()
PY
export KP_ROOT=$ROOT
CGRPS=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC"
 "GRBM_GUI_ACTIVE FETCH_SIZE"
 "WRITE_SIZE"
)
rm -rf /tmp/spmc; i=0
for g in "${CGRPS[@]}"; do
  (cd /tmp && rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/spmc/g$i -- python /tmp/small_one.py > /tmp/spmc_$i.log 2>&1)
  i=$((i+1))
done
python - <<'PY' > $ROOT/gpurun_out/small_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('/tmp/spmc/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        import re
        m = re.search(r'(\w*small\w*<[^>]*>)', r['Kernel_Name'])
        if not m: continue
        k = m.group(1)
        a = acc[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
        if r['Counter_Name'] in ('SQ_WAVES', 'FETCH_SIZE'):
            d = dur[k]; d[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); d[1] += 1
for k, cs in acc.items():
    print(k, 'avg_us', round(dur[k][0] / max(1, dur[k][1]) / 1e3, 1))
    for c, (v, n) in sorted(cs.items()):
        print(f'   {c:28s} {v / n:16.0f}')
PY
cat $ROOT/gpurun_out/small_pmc.txt; grep -v "^[EWI]2026" /tmp/spmc_0.log | tail -15; ls /tmp/spmc/g0/*/ | head
