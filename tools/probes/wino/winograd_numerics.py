#!/usr/bin/env python3
"""Numerics of Winograd F(2x2, 3x3) with 16-bit transformed operands, against the direct 16-bit convolution the library runs
("what comes next"): both use the SAME bf16 / fp16 inputs and weights and accumulate in fp32; Winograd
additionally rounds the transformed input tile (B^T d B) and the transformed filter (G g G^T) to 16 bits before the products.
CPU, float64 reference.   python tools/probes/winograd_numerics.py"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd(x, w, dt):
    """x [N,C,H,W] (H, W even), w [K,C,3,3] already rounded to dt; transformed operands rounded to dt; fp32-like accumulate
    (emulated in float64 here: the accumulation error is not what is being studied)."""
    N, C, H, W = x.shape
    K = w.shape[0]
    xp = F.pad(x, (1, 1, 1, 1), mode='replicate').double()
    U = torch.einsum('ij,kcjl,ml->kcim', G, w.double(), G).to(dt).double()            # [K,C,4,4]
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                        # [N,C,H/2,W/2,4,4]
    V = torch.einsum('ij,ncpqjl,ml->ncpqim', Bt, tiles, Bt).to(dt).double()
    M = torch.einsum('kcim,ncpqim->nkpqim', U, V)
    Y = torch.einsum('ij,nkpqjl,ml->nkpqim', At, M, At)                              # [N,K,H/2,W/2,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)


for dt, name, ulp in ((torch.bfloat16, 'bf16', 2.0 ** -8), (torch.float16, 'fp16', 2.0 ** -11)):
    for C, K in ((64, 64), (256, 256), (512, 512)):
        x = F.leaky_relu(torch.randn(2, C, 16, 16), 0.01).to(dt)                    # a post-BatchNorm + LeakyReLU activation
        w = (torch.randn(K, C, 3, 3) * (2.0 / (9 * K)) ** 0.5).to(dt)               # kaiming fan_out
        ref = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode='replicate'), w.double())
        wg = winograd(x, w, dt)
        scale = ref.abs().mean()
        err = (wg - ref)
        direct_out = ref.to(dt).double()                                           # what the direct kernel stores (fp32 accumulate ~ exact here)
        wino_out = wg.to(dt).double()
        bad = ((wino_out - direct_out).abs() > 0).double().mean()
        print(f'{name} C={C:3d} K={K:3d}: Winograd error before the output rounding: rel-L2 {err.norm() / ref.norm():.2e}, '
              f'max {err.abs().max() / scale:.2e} of mean|y| (one {name} ulp of mean|y| = {ulp:.1e}); '
              f'{100 * bad:.0f} % of the stored outputs differ from the direct kernel\'s, '
              f'stored rel-L2 {(wino_out - ref).norm() / ref.norm():.2e} vs direct {(direct_out - ref).norm() / ref.norm():.2e}')
