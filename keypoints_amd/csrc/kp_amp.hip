// Dynamic loss scaling for the fp16 storage mode -- the role of apex `amp.scale_loss` + the O2 optimizer wrapper in the
// reference's train step (transporter.py:51,84-89; keypoints.py:48,79-84): the loss gradient is multiplied by a scale so
// that fp16 gradients do not underflow; before the update the gradients are checked for inf / nan, un-scaled inside
// the Adam kernel, and the scale adapts (halve on overflow and skip the step, double after `growth_interval` clean
// steps).  Everything, including Adam's step counter and bias corrections, lives in a small DEVICE state block, so a
// train step never synchronises with the host.
//
//   state (8 doubles): [0] scale  [1] clean steps since the last change  [2] found_inf flag of this step
//                      [3] Adam step count t (skipped steps do not count)  [4] 1 - b1^t  [5] sqrt(1 - b2^t)
//                      [6] steps skipped so far  [7] 1 / scale
#include "kp_common.h"

namespace {

constexpr int TPB = 256;

__global__ __launch_bounds__(TPB) void amp_check_k(const float* __restrict__ g, size_t n, double* __restrict__ state) {
    bool bad = false;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (size_t)gridDim.x * TPB) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        // x - x is 0 for finite x and nan for inf / nan
        const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
        bad |= !(t == 0.f);
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * TPB) bad |= !(g[i] - g[i] == 0.f);
    if (__any(bad) && (threadIdx.x & 63) == 0) state[2] = 1.0;   // benign race: every writer stores the same value
}

__global__ void amp_prepare_k(double* __restrict__ state, double b1, double b2) {
    if (state[2] == 0.0) {
        const double t = state[3] + 1.0;
        state[3] = t;
        state[4] = 1.0 - pow(b1, t);
        state[5] = sqrt(1.0 - pow(b2, t));
    }
    state[7] = 1.0 / state[0];
}

__global__ void amp_update_k(double* __restrict__ state, double growth, double backoff, double interval, double max_scale) {
    if (state[2] != 0.0) {
        state[0] = fmax(state[0] * backoff, 1.0);
        state[1] = 0.0;
        state[6] += 1.0;
    } else {
        state[1] += 1.0;
        if (state[1] >= interval) {
            state[0] = fmin(state[0] * growth, max_scale);
            state[1] = 0.0;
        }
    }
    state[2] = 0.0;
}

// torch.optim.Adam on un-scaled gradients; the whole launch is a no-op when this step overflowed
__global__ __launch_bounds__(TPB) void adam_scaled_k(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                     float b1, float b2, float eps, const double* __restrict__ state) {
    if (state[2] != 0.0) return;
    const float inv_scale = (float)state[7], bc1 = (float)state[4], sbc2 = (float)state[5];
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * inv_scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sbc2 + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

// y = x * (float)scale, the scale read from the state block: amp.scale_loss's multiplication, forward (the loss) and backward (its
// incoming gradient), without a host read or an fp64 -> fp32 conversion launch in front
__global__ void amp_scale_k(const float* __restrict__ x, const double* __restrict__ state, float* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * (float)state[0];
}

}  // namespace

extern "C" {

int kp_amp_scale(const float* x, const double* state, float* y, size_t n, void* stream) {
    KP_REQUIRE(x && state && y && n > 0, "kp_amp_scale: bad args");
    amp_scale_k<<<cdiv(n, TPB), TPB, 0, (hipStream_t)stream>>>(x, state, y, n);
    return kp_check_launch("amp_scale");
}

int kp_amp_init(double* state, double init_scale, void* stream) {
    KP_REQUIRE(state && init_scale >= 1.0, "kp_amp_init: bad args");
    const double h[8] = {init_scale, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0, 1.0 / init_scale};
    if (hipMemcpyAsync(state, h, sizeof(h), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {   // h is a stack buffer: one-off setup call
        kp_set_error("kp_amp_init: copy failed");
        return KP_E_LAUNCH;
    }
    return KP_OK;
}

int kp_amp_check(const float* g, size_t n, double* state, void* stream) {
    KP_REQUIRE(g && state && n > 0, "kp_amp_check: bad args");
    KP_REQUIRE(((size_t)g & 15) == 0, "kp_amp_check: gradient buffer must be 16-byte aligned");
    kp_count_bytes(4.0 * n);
    unsigned grid = cdiv(n / 4 + 1, TPB * 4);
    if (grid > 2048) grid = 2048;
    amp_check_k<<<grid, TPB, 0, (hipStream_t)stream>>>(g, n, state);
    return kp_check_launch("amp_check");
}

int kp_amp_prepare(double* state, float b1, float b2, void* stream) {
    KP_REQUIRE(state, "kp_amp_prepare: null state");
    amp_prepare_k<<<1, 1, 0, (hipStream_t)stream>>>(state, (double)b1, (double)b2);
    return kp_check_launch("amp_prepare");
}

int kp_adam_step_scaled(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                        const double* state, void* stream) {
    KP_REQUIRE(p && g && m && v && state && n > 0, "kp_adam_step_scaled: bad args");
    kp_count_bytes(28.0 * n);
    adam_scaled_k<<<cdiv(n, TPB), TPB, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, b1, b2, eps, state);
    return kp_check_launch("adam_step_scaled");
}

int kp_amp_update(double* state, float growth_factor, float backoff_factor, int growth_interval, float max_scale,
                  void* stream) {
    KP_REQUIRE(state && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval > 0,
               "kp_amp_update: bad args");
    amp_update_k<<<1, 1, 0, (hipStream_t)stream>>>(state, (double)growth_factor, (double)backoff_factor,
                                                   (double)growth_interval, (double)max_scale);
    return kp_check_launch("amp_update");
}

}  // extern "C"
