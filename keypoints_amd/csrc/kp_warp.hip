// Input-side augmentation of the KeyNet config: thin-plate-spline warp and per-sample rotation, each a bilinear
// resampling with zero padding (keypoints/tps.py:10-87,128-131,154-166; keypoints/data_augments.py:13-38).
// The reference builds an [N][H][W][2] sampling grid on the CPU (two bmm's over an [N][H*W][T] radial-basis
// tensor), ships it to the device and calls F.grid_sample; here the sampling position is evaluated per output pixel
// in registers (T is 4: closed form) and the four taps are gathered straight from the NCHW fp32 image -- one pass,
// no grid tensor.  HBM-bound: read + write of the image.
// grid_sample / affine_grid semantics are those of the torch defaults (bilinear, align_corners=False).
#include "kp_common.h"

namespace {

constexpr int TPB = 256;

// bilinear tap of grid_sample(padding_mode='zeros', align_corners=False) at normalised position (sx, sy)
struct Taps {
    int x0, y0;
    float w00, w01, w10, w11;   // (y0,x0) (y0,x0+1) (y0+1,x0) (y0+1,x0+1)
    bool in00, in01, in10, in11;
};
__device__ __forceinline__ Taps make_taps(float sx, float sy, int H, int W) {
    const float ix = ((sx + 1.f) * (float)W - 1.f) * 0.5f;
    const float iy = ((sy + 1.f) * (float)H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    Taps t;
    t.x0 = (int)fx; t.y0 = (int)fy;
    const float ex = (fx + 1.f) - ix, ey = (fy + 1.f) - iy;   // weights of the west / north taps
    const float wx = ix - fx, wy = iy - fy;
    t.w00 = ex * ey; t.w01 = wx * ey; t.w10 = ex * wy; t.w11 = wx * wy;
    const bool xa = t.x0 >= 0 && t.x0 < W, xb = t.x0 + 1 >= 0 && t.x0 + 1 < W;
    const bool ya = t.y0 >= 0 && t.y0 < H, yb = t.y0 + 1 >= 0 && t.y0 + 1 < H;
    t.in00 = xa && ya; t.in01 = xb && ya; t.in10 = xa && yb; t.in11 = xb && yb;
    return t;
}
__device__ __forceinline__ float sample(const float* __restrict__ img, const Taps& t, int W) {
    const float* p = img + (long long)t.y0 * W + t.x0;
    float v = 0.f;
    if (t.in00) v += p[0] * t.w00;
    if (t.in01) v += p[1] * t.w01;
    if (t.in10) v += p[W] * t.w10;
    if (t.in11) v += p[W + 1] * t.w11;
    return v;
}

__global__ __launch_bounds__(TPB) void warp_tps_k(const float* __restrict__ x, const float* __restrict__ theta,
                                                  const float* __restrict__ ctrl, float* __restrict__ out, int N, int C,
                                                  int H, int W, int T, int reduced, int ctrl_batched) {
    const long long pix = (long long)blockIdx.x * TPB + threadIdx.x;
    if (pix >= (long long)N * H * W) return;
    const int j = (int)(pix % W);
    const int i = (int)((pix / W) % H);
    const int n = (int)(pix / ((long long)W * H));
    const float gx = ruler_at(j, W), gy = ruler_at(i, H);
    const int P = reduced ? T + 2 : T + 3;
    const float* th = theta + (long long)n * P * 2;
    const float* ct = ctrl + (ctrl_batched ? (long long)n * T * 2 : 0);
    const float* w = th;                 // [T][2] (reduced: [T-1][2], the first weight is minus their sum)
    const float* a = th + (P - 3) * 2;   // [3][2]
    float bx = 0.f, by = 0.f, w0x = 0.f, w0y = 0.f;
    if (reduced)
        for (int t = 0; t < T - 1; ++t) { w0x -= w[t * 2]; w0y -= w[t * 2 + 1]; }
    for (int t = 0; t < T; ++t) {
        const float dx = gx - ct[t * 2], dy = gy - ct[t * 2 + 1];
        const float D = sqrtf(dx * dx + dy * dy);
        const float U = (D * D) * logf(D + 1e-6f);
        const float wx = reduced ? (t == 0 ? w0x : w[(t - 1) * 2]) : w[t * 2];
        const float wy = reduced ? (t == 0 ? w0y : w[(t - 1) * 2 + 1]) : w[t * 2 + 1];
        bx += U * wx; by += U * wy;
    }
    const float zx = (a[0] + gx * a[2] + gy * a[4]) + bx;
    const float zy = (a[1] + gx * a[3] + gy * a[5]) + by;
    const Taps tp = make_taps((gx + zx) * 2.f - 1.f, (gy + zy) * 2.f - 1.f, H, W);
    const long long hw = (long long)H * W, o = (long long)i * W + j;
    for (int c = 0; c < C; ++c) out[((long long)n * C + c) * hw + o] = sample(x + ((long long)n * C + c) * hw, tp, W);
}

__global__ __launch_bounds__(TPB) void warp_affine_k(const float* __restrict__ x, const float* __restrict__ mat,
                                                     float* __restrict__ out, int N, int C, int H, int W) {
    const long long pix = (long long)blockIdx.x * TPB + threadIdx.x;
    if (pix >= (long long)N * H * W) return;
    const int j = (int)(pix % W);
    const int i = (int)((pix / W) % H);
    const int n = (int)(pix / ((long long)W * H));
    // affine_grid(align_corners=False): base = linspace(-1, 1, S) * (S - 1) / S
    const float xb = (W > 1 ? (-1.f + 2.f * (float)j / (float)(W - 1)) : 0.f) * ((float)(W - 1) / (float)W);
    const float yb = (H > 1 ? (-1.f + 2.f * (float)i / (float)(H - 1)) : 0.f) * ((float)(H - 1) / (float)H);
    const float* m = mat + (long long)n * 6;
    const Taps tp = make_taps(m[0] * xb + m[1] * yb + m[2], m[3] * xb + m[4] * yb + m[5], H, W);
    const long long hw = (long long)H * W, o = (long long)i * W + j;
    for (int c = 0; c < C; ++c) out[((long long)n * C + c) * hw + o] = sample(x + ((long long)n * C + c) * hw, tp, W);
}

// the same with the matrix of a pure rotation built per sample on the device: [[cos a, sin a, 0], [-sin a, cos a, 0]]
// (tps.py:154-166 builds it on the CPU from the drawn angles and ships it)
__global__ __launch_bounds__(TPB) void warp_rotate_k(const float* __restrict__ x, const float* __restrict__ angle,
                                                     float* __restrict__ out, int N, int C, int H, int W) {
    const long long pix = (long long)blockIdx.x * TPB + threadIdx.x;
    if (pix >= (long long)N * H * W) return;
    const int j = (int)(pix % W);
    const int i = (int)((pix / W) % H);
    const int n = (int)(pix / ((long long)W * H));
    const float xb = (W > 1 ? (-1.f + 2.f * (float)j / (float)(W - 1)) : 0.f) * ((float)(W - 1) / (float)W);
    const float yb = (H > 1 ? (-1.f + 2.f * (float)i / (float)(H - 1)) : 0.f) * ((float)(H - 1) / (float)H);
    const float a = angle[n];
    const float ca = cosf(a), sa = sinf(a);
    const Taps tp = make_taps(ca * xb + sa * yb + 0.f, -sa * xb + ca * yb + 0.f, H, W);
    const long long hw = (long long)H * W, o = (long long)i * W + j;
    for (int c = 0; c < C; ++c) out[((long long)n * C + c) * hw + o] = sample(x + ((long long)n * C + c) * hw, tp, W);
}

}  // namespace

extern "C" {

int kp_warp_tps(const float* x, const float* theta, const float* ctrl, float* out, int N, int C, int H, int W, int T,
                int reduced, int ctrl_batched, void* stream) {
    KP_REQUIRE(x && theta && ctrl && out, "kp_warp_tps: null pointer");
    KP_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && T >= 1 && T <= 4096, "kp_warp_tps: bad dims");
    kp_count_bytes(8.0 * N * C * H * W);
    const long long pix = (long long)N * H * W;
    warp_tps_k<<<cdiv(pix, TPB), TPB, 0, (hipStream_t)stream>>>(x, theta, ctrl, out, N, C, H, W, T, reduced, ctrl_batched);
    return kp_check_launch("warp_tps");
}

int kp_warp_affine(const float* x, const float* mat, float* out, int N, int C, int H, int W, void* stream) {
    KP_REQUIRE(x && mat && out, "kp_warp_affine: null pointer");
    KP_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "kp_warp_affine: bad dims");
    kp_count_bytes(8.0 * N * C * H * W);
    const long long pix = (long long)N * H * W;
    warp_affine_k<<<cdiv(pix, TPB), TPB, 0, (hipStream_t)stream>>>(x, mat, out, N, C, H, W);
    return kp_check_launch("warp_affine");
}

int kp_warp_rotate(const float* x, const float* angle, float* out, int N, int C, int H, int W, void* stream) {
    KP_REQUIRE(x && angle && out, "kp_warp_rotate: null pointer");
    KP_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "kp_warp_rotate: bad dims");
    kp_count_bytes(8.0 * N * C * H * W);
    const long long pix = (long long)N * H * W;
    warp_rotate_k<<<cdiv(pix, TPB), TPB, 0, (hipStream_t)stream>>>(x, angle, out, N, C, H, W);
    return kp_check_launch("warp_rotate");
}

}  // extern "C"
