"""Clock stamps of the Pong-class forward band kernel (a -DKP_FB_TRACE build of kp_conv_small.hip): per workgroup
[hw id, start, then per band: compute done / barrier passed / staged, ..., end].  python tools/probes/small_trace.py lib.so [C K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from keypoints_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from keypoints_amd._lib import call, ptr
C, K = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16, 16)
N, H = 256, 84; bf = torch.bfloat16; st = torch.cuda.current_stream().cuda_stream
a = (torch.randn(N, H + 2, H + 2, C, device='cuda') * 0.5).to(bf); wf = (torch.randn(K, 3, 3, C, device='cuda') * 0.1).to(bf)
bias = torch.zeros(K, device='cuda'); y = torch.empty(N, H, H, K, device='cuda', dtype=bf)
stats = torch.zeros(N * H * H // 64 + 1, 2, K, device='cuda')
for _ in range(3):
    call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, H + 2, C, K, 3, 1, st)
torch.cuda.synchronize()
raw = stats.view(-1)[8192 * 2 * K:].cpu().numpy().view(np.uint64)[:512 * 32].reshape(512, 32)
ids = raw[:, 1]; t = np.concatenate([raw[:, :1], raw[:, 2:]], 1).astype(np.int64)
nst = int((t[0] > 0).sum()); t = t[:, :nst]; t0 = t[:, 0].min()
t = t - t0
xcc = (ids >> 32).astype(int); cu = ((ids >> 8) & 0xf).astype(int); se = ((ids >> 13) & 0x7).astype(int); sh = ((ids >> 12) & 1).astype(int)
print('stamps per workgroup', nst, ' clock ticks: start min/max', t[:, 0].min(), t[:, 0].max(), ' end min/mean/max', t[:, -1].min(), int(t[:, -1].mean()), t[:, -1].max())
d = np.diff(t, axis=1)
names = ['prologue', 'issue0'] + ['wait_barrier', 'commit', 'issue+items'] * ((nst - 3) // 3) 
ph = {}
for i in range(d.shape[1]):
    nm = names[i] if i < len(names) else 'tail'
    ph.setdefault(nm, []).append(d[:, i])
for k, v in ph.items():
    v = np.stack(v, 1); print(f'{k:14s} mean {v.mean():9.0f}  min {v.min():8d}  max {v.max():8d}  per band means', np.round(v.mean(0)).astype(int))
import collections
cnt = collections.Counter(zip(xcc, se, sh, cu))
print('distinct (xcc, se, sh, cu):', len(cnt), ' workgroups per CU histogram:', collections.Counter(cnt.values()))
print('per-xcc end mean', [int(t[xcc == i, -1].mean()) for i in range(8)])
# do the two workgroups of a CU run at the same time?  (same CU = same counter)
byc = collections.defaultdict(list)
for i in range(512): byc[(xcc[i], se[i], sh[i], cu[i])].append((t[i, 0] + t0, t[i, -1] + t0, i))
ov = []
for k, v in byc.items():
    if len(v) == 2:
        (a0, a1, i), (b0, b1, j) = sorted(v)
        ov.append((b0 - a0, a1 - a0, b1 - a0, i, j))
ov = np.array(ov)
print('second workgroup of a CU starts after (ticks) mean/min/max', int(ov[:, 0].mean()), ov[:, 0].min(), ov[:, 0].max(), ' first ends at mean', int(ov[:, 1].mean()), ' second ends at mean', int(ov[:, 2].mean()))
print('examples', ov[:6].tolist())
for k in range(8):
    m = xcc == k
    st_, en_ = t[m, 0], t[m, -1]
    print(f'xcc {k}: {m.sum()} workgroups, entry spread {st_.max() - st_.min()} ticks, first entry -> last end {en_.max() - st_.min()} ticks, last start -> first end {en_.min() - st_.max()}')
