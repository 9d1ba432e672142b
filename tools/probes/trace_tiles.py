"""Per-tile (s_memtime, s_memrealtime) stamps of workgroup 0 / wave 0 of the patch conv kernel over a whole launch:
cycles per tile and the shader clock they ran at (s_memrealtime ticks at 100 MHz).
    python tools/probes/trace_tiles.py C K H"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from keypoints_amd._lib import call, ptr
N = 128; st = torch.cuda.current_stream().cuda_stream; bf = torch.bfloat16
C, K, H = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (256, 128, 128)
a = (torch.randn(N, H + 2, H + 2, C, device='cuda') * 0.5).to(bf); wf = (torch.randn(K, 3, 3, C, device='cuda') * 0.05).to(bf)
bias = torch.zeros(K, device='cuda'); y = torch.empty(N, H, H, K, device='cuda', dtype=bf)
stats = torch.empty(N * H * H // 64, 2, K, device='cuda')
tr = torch.zeros(512, device='cuda', dtype=torch.int64)
for i in range(3):
    call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, H + 2, C, K, 3, 1, st)
call('kp_config', 1, 32)
call('kp_config_ptr', 2, ptr(tr))
for i in range(3):   # the last launch's stamps stay in the buffer
    call('kp_conv_fwd', ptr(a), ptr(wf), ptr(bias), ptr(y), ptr(stats), N, H + 2, H + 2, C, K, 3, 1, st)
torch.cuda.synchronize()
call('kp_config_ptr', 2, None); call('kp_config', 1, 0)
t = tr.cpu().tolist()
cyc = [v for v in t[0::2] if v]; real = [v for v in t[1::2] if v]
n = min(len(cyc), len(real))
dc = [cyc[i + 1] - cyc[i] for i in range(n - 1)]; dr = [real[i + 1] - real[i] for i in range(n - 1)]
print('tiles', n, 'ideal MFMA cycles/tile', 9 * (C // 64) * 1024)
print('cycles/tile:', ' '.join(str(v) for v in dc))
print('GHz       :', ' '.join(f'{c / (r * 10.0):.2f}' for c, r in zip(dc, dr)))
print('mean cycles/tile %.0f   mean GHz %.2f' % (sum(dc) / len(dc), sum(dc) / (sum(dr) * 10.0)))
