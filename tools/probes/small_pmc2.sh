#!/bin/bash
# SQ counters of the BatchNorm-input band kernels (32 -> 16, 32 -> 32, 16 -> 16) at config-2 shape: which unit bounds them
cd "$(dirname "$0")/../.." || exit 1
ROOT=$(pwd); export TMPDIR=/tmp
cat > /tmp/small_two.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['KP_ROOT'])
import torch
from keypoints_amd import _lib
from keypoints_amd._lib import call, ptr
N, H = 256, 84; bf = torch.bfloat16; st = torch.cuda.current_stream().cuda_stream
for C, K in ((32, 16), (32, 32), (16, 16)):
    yp = (torch.randn(N, H, H, C, device='cuda') * 0.5).to(bf); wf = (torch.randn(K, 3, 3, C, device='cuda') * 0.1).to(bf)
    bn = [torch.randn(C, device='cuda') * 0.1, torch.rand(C, device='cuda') + 0.5, torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.1]
    bias = torch.zeros(K, device='cuda'); y = torch.empty(N, H, H, K, device='cuda', dtype=bf); stats = torch.empty(N * H * H // 64 + 1, 2, K, device='cuda')
    for _ in range(4):
        call('kp_conv_fwd_bnin', ptr(yp), *[ptr(t) for t in bn], 0.01, ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H, H, C, K, 1, st)
    torch.cuda.synchronize()
PY
export KP_ROOT=$ROOT
CGRPS=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC"
 "GRBM_GUI_ACTIVE"
)
rm -rf /tmp/spmc2; i=0
for g in "${CGRPS[@]}"; do
  (cd /tmp && rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/spmc2/g$i -- python /tmp/small_two.py > /tmp/spmc2_$i.log 2>&1)
  i=$((i+1))
done
python - <<'PY' > $ROOT/gpurun_out/small_pmc2.txt
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('/tmp/spmc2/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(\w*small\w*<[^>]*>)', r['Kernel_Name'])
        if not m: continue
        k = m.group(1)
        a = acc[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
        if r['Counter_Name'] in ('SQ_WAVES', 'GRBM_GUI_ACTIVE'):
            d = dur[k]; d[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); d[1] += 1
for k, cs in acc.items():
    print(k, 'avg_us', round(dur[k][0] / max(1, dur[k][1]) / 1e3, 1))
    for c, (v, n) in sorted(cs.items()):
        print(f'   {c:28s} {v / n:16.0f}')
PY
cat $ROOT/gpurun_out/small_pmc2.txt
