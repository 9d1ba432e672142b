"""Clock stamps (100 MHz) of the Pong-class weight-gradient band kernel, -DKP_FB_TRACE build: python tools/probes/wgrad_trace.py lib.so [C K]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from keypoints_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from keypoints_amd._lib import call, ptr
C, K = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16, 32)
N, H = 256, 84; bf = torch.bfloat16; st = torch.cuda.current_stream().cuda_stream
a = (torch.randn(N, H + 2, H + 2, C, device='cuda') * 0.5).to(bf)
dy = torch.zeros(N, H + 4, H + 4, K, device='cuda', dtype=bf); dy[:, 2:-2, 2:-2] = torch.randn(N, H, H, K, device='cuda').to(bf)
parts = _lib.lib().kp_conv_wgrad_partials(N, H, H, C, K, 3, 2, 1)
scratch = torch.empty(parts, K, 3, 3, C, device='cuda')
for _ in range(3):
    call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, H, C, K, 3, 2, 1, st)
torch.cuda.synchronize()
buf = np.zeros(512 * 64, dtype=np.uint64)
f = ctypes.CDLL(_lib.LIB_PATH).kp_debug_wgrad_trace
f.argtypes = [ctypes.c_void_p]; print('copy rc', f(buf.ctypes.data))
t = buf.reshape(512, 64)[:parts].astype(np.int64)
nst = int((t[0] > 0).sum()); t = t[:, :nst]; t0 = t[:, 0].min(); t = t - t0
print('workgroups', parts, 'stamps', nst, ' start spread', t[:, 0].max(), ' end min/mean/max (x10 ns)', t[:, -1].min(), int(t[:, -1].mean()), t[:, -1].max())
d = np.diff(t, axis=1)
names = ['issue0', 'commit0+barrier'] + ['issue_next', 'multiply', 'commit', 'barrier'] * 32
agg = {}
for i in range(d.shape[1]): agg.setdefault(names[i], []).append(d[:, i])
for k, v in agg.items():
    v = np.stack(v, 1); print(f'{k:16s} mean {v.mean():8.1f} min {v.min():6d} max {v.max():6d}  x{v.shape[1]} per workgroup, total mean {v.sum(1).mean():8.1f}')
