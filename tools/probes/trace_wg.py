"""s_memtime stamps of workgroup 0 / wave 0 of the tap-reuse weight-gradient kernel: per stage
[DMA issue | MFMA + transpose reads | vmcnt wait | barrier].    python tools/probes/trace_wg.py C K H"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from keypoints_amd import _lib as L
from keypoints_amd._lib import call, ptr
N = 128; st = torch.cuda.current_stream().cuda_stream; bf = torch.bfloat16
C, K, H = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (512, 512, 32)
a = (torch.randn(N, H + 2, H + 2, C, device='cuda') * 0.5).to(bf)
dy = torch.zeros(N, H + 4, H + 4, K, device='cuda', dtype=bf); dy[:, 2:-2, 2:-2] = (torch.randn(N, H, H, K, device='cuda') * 0.5).to(bf)
scratch = torch.zeros(L.lib().kp_conv_wgrad_partials(N, H, H, C, K, 3, 2, 1), K, 3, 3, C, device='cuda')
tr = torch.zeros(512, device='cuda', dtype=torch.int64)
for i in range(3):
    call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, H, C, K, 3, 2, 1, st)
call('kp_config_ptr', 2, ptr(tr))
call('kp_conv_wgrad', ptr(a), ptr(dy), ptr(scratch), N, H, H, C, K, 3, 2, 1, st)
torch.cuda.synchronize()
call('kp_config_ptr', 2, None)
t = [v for v in tr.cpu().tolist() if v]
d = [t[i + 1] - t[i] for i in range(len(t) - 1)]
print('stamps', len(t), 'prologue', d[0], 'total', t[-1] - t[0], 'ideal MFMA cycles/stage (2 waves/SIMD)', 2 * 72 * 32)
body = d[1:]
for i in range(0, min(len(body), 4 * 12), 4):
    print('stage %2d: dma-issue %5d  mfma %5d  vmwait %5d  barrier %5d   sum %5d' % (i // 4, *body[i:i + 4], sum(body[i:i + 4])))
n = len(body) // 4
tot = [sum(body[j] for j in range(k, 4 * n, 4)) / n for k in range(4)]
print('mean per stage: dma-issue %.0f  mfma %.0f  vmwait %.0f  barrier %.0f   sum %.0f' % (*tot, sum(tot)))
