/*
 * kp_hip.h -- C ABI of libkp_hip.so: the MI355X (gfx950) kernels behind the
 * Transporter / KeyNet train-step hot path of DuaneNielsen/keypoints.
 *
 * The reference has no FFI: its path is Python calling torch.nn / torch.autograd
 * (SURVEY.md section 8b).  Each entry point below therefore cites the reference
 * Python statement(s) whose arithmetic it replaces; keypoints_amd/ binds them with
 * ctypes (keypoints_amd/_lib.py) and INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer; the caller owns all memory (torch's
 *     caching allocator in our host code) including workspaces;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and
 *     the call never synchronises;
 *   - return value: 0 = ok, <0 = error (KP_E_*); kp_last_error() gives text;
 *   - dtype: KP_F32 (0), KP_BF16 (1) or KP_F16 (2) = storage type of activations /
 *     packed weights ("T" below).  Accumulation is always fp32, statistics fp64;
 *   - activation images are NHWC with an optional physical border:
 *     img[N][H+2*pad][W+2*pad][C]; (H, W) always name the interior size.
 */
#ifndef KP_HIP_H
#define KP_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KP_F32 0
#define KP_BF16 1
#define KP_F16 2   /* IEEE half storage (the reference's apex-O2 type, transporter.py:51): same kernels, same MFMA rate,
                    * 3 more mantissa bits than bf16, 5-bit exponent -> train with loss scaling (kp_amp_*) */

#define KP_OK 0
#define KP_E_ARG (-1)         /* bad argument / unsupported shape */
#define KP_E_LAUNCH (-2)      /* hipGetLastError() != hipSuccess after a launch */
#define KP_E_UNSUPPORTED (-3)

/* BN+activation output modes (vgg.py:24-38: what follows the activation) */
#define KP_TAIL_NONE 0
#define KP_TAIL_MAXPOOL2 1    /* nn.MaxPool2d(2,2)            vgg.py:24 */
#define KP_TAIL_UPSAMPLE2 2   /* nn.UpsamplingBilinear2d(x2)  vgg.py:26 */

int kp_version(void);
const char* kp_last_error(void);
/* How many leading rows of the partial-sum buffer a kp_conv_fwd (stats != NULL) / kp_bn_act_bwd_reduce call of this
 * shape writes: a pure function of the arguments and the device (the persistent kernels write a few rows per workgroup).
 * Only those rows are defined; pass the value on to kp_bn_finalize / kp_bn_bwd_finalize.  Never more than the
 * documented buffer size (ceil(N*Ho*Wo/64) and N*H rows).  Nothing is launched.  < 0: KP_E_*. */
long long kp_conv_fwd_stats_rows(int N, int Hi, int Wi, int C, int K, int R, int dtype);
long long kp_bn_bwd_rows(int gp, int tail, int N, int H, int W, int C, int dtype);
/* Legacy side channel (thread-local): the row count of this thread's most recent statistics-producing call.  Equal to
 * what the two functions above return for that call's shape; new code should use those. */
long long kp_last_rows(void);
/* number of CUs of the current device (for grid sizing by host code) */
int kp_device_cus(void);

/* Tuning switches for A/B measurements (not needed in normal use). */
#define KP_CFG_CONV_PATCH 0   /* 1 (default): LDS-patch 3x3 kernel where eligible; 2: its pair-tile generation first (kp_conv_pair.hip: measured equal, opt-in); 0: linear implicit-GEMM kernel */
#define KP_CFG_DEBUG 1        /* ablation mask for kernel measurements (0 = normal operation) */
int kp_config(int key, int value);
/* the value kp_config(key, .) last stored (the default if never set): KP_CFG_CUS, KP_CFG_CONV_PATCH, KP_CFG_DEBUG, KP_CFG_WALK, KP_CFG_BN_FAST,
 * KP_CFG_BRED_PATCH, KP_CFG_FIRST_ROWS; a caller that changes a process-global knob for its own lifetime (keypoints_amd.dist.GradReducer:
 * the reduced persistent grid of the overlap mode) restores what it found.  KP_E_ARG (negative) for any other key. */
int kp_config_get(int key);
#define KP_CFG_WGRAD_PATCH 3  /* 1 (default): tap-reuse weight-gradient kernel where eligible; 0: per-tap kernel */
#define KP_CFG_BN_FAST 4      /* 1 (default): multi-row unrolled bf16 BatchNorm/activation kernels; 0: general kernels */
#define KP_CFG_WALK 5         /* memory walk order of the passes, a mask of KP_WALK_* (default KP_WALK_DEFAULT; 0 = every pass front to back,
                               * plain loads).  Changes no result except the summation order of the fp32 BatchNorm partials. */
#define KP_WALK_CONV_LINEAR 1     /* persistent 3x3 kernels take their tiles in ONE front over the batch (step i of the grid covers
                                   * tile ids [i*grid, (i+1)*grid)) instead of eight per-XCD ranges */
#define KP_WALK_FWD_REVERSE 2     /* kp_bn_act_fwd sweeps the conv output back to front (last-written images first) */
#define KP_WALK_REDUCE_REVERSE 4  /* kp_bn_act_bwd_reduce sweeps the incoming gradient back to front; the apply pass then runs front to back */
#define KP_WALK_APPLY_REVERSE 8   /* kp_bn_act_bwd_apply back to front (measurement only) */
/* 4 instead of 8 channels per thread (8-byte lanes, about half the registers, twice the waves per SIMD) for one class of
 * BatchNorm passes each: */
#define KP_WALK_V4_FWD_NONE 0x200
#define KP_WALK_V4_FWD_POOL 0x400
#define KP_WALK_V4_FWD_UP 0x800
#define KP_WALK_V4_REDUCE 0x1000
#define KP_WALK_V4_APPLY 0x2000
#define KP_WALK_V4_POOL_REDUCE 0x4000
#define KP_WALK_V4_POOL_APPLY 0x8000
#define KP_WALK_STREAM_LOADS 128  /* the BatchNorm passes read with non-temporal loads (nothing re-reads their inputs soon) */
#define KP_WALK_V4_ALL 0xfe00
#define KP_WALK_DEFAULT (KP_WALK_CONV_LINEAR | KP_WALK_FWD_REVERSE | KP_WALK_REDUCE_REVERSE | KP_WALK_STREAM_LOADS | KP_WALK_V4_ALL)
#define KP_CFG_BN_FLAT 8      /* 1 (default): BatchNorm-backward passes of narrow, few-channel maps keep their loads in flight across image rows (bn_bwd_flat_k); 0: row kernel */
#define KP_CFG_BRED_PATCH 9   /* kp_conv_dgrad_bnred for the matrix-core layers (C, K multiples of 64, 16x16-tileable maps): 1 (default) where the data
                                 gradient has 64 output channels per tile (C % 128 != 0: a gain); 2: for 128-channel tiles too (measured slower: register
                                 spills); 0: only for the 16/32-channel layers */
#define KP_CFG_SMALL_BAND 7   /* 1 (default): LDS-staged weight gradient of the 16/32-channel layers where a band of rows fits; 0: gather kernel */
#define KP_CFG_FIN_WIDE 11      /* 1: the 1024-thread statistics finalize for every layer (measurement); 0 (default): the 4-wave kernel for C <= 32 */
#define KP_CFG_DISABLE_MFMA 12  /* 1: every convolution on the generic VALU kernels (measurement / debugging); 0 (default) */
#define KP_CFG_WG16 13          /* bit mask (default 7): which small-channel weight gradients take the 16 x 16 x 32 matrix-core kernel */
#define KP_CFG_FIRST_ROWS 14   /* single-channel first layers with 16 / 32 filters: 1 (default) the row kernel (a lane = one pixel x eight channels, weights in registers), 0: conv3_first_fwd_k, > 1: workgroups per launch of the row kernel (measurement) */
#define KP_CFG_FIRST_KTN 10  /* kp_conv_first_fwd_bnact: 2 (default) one wavefront computes all 64 channels of its groups (66 us at 128 x 128 x 128: the plain convolution's time); 1: a pair of wavefronts, 32 channels each (75 us) */
#define KP_CFG_STATS_STORE 18  /* 0 (default) / 1 (measurement): BatchNorm sums of the 128-channel-tile forward kernel taken at the store stage from the STORED (rounded) values in 16 registers instead of 64 -- the kernel then leaves room on a CU for one wave per SIMD of another stream's pass kernels; statistics differ by the storage rounding (docs/experiments/r06.md section 10) */
#define KP_CFG_FIN_FOLD 16     /* 1 (default): kp_conv_fwd_fin runs the BatchNorm finalize in the tail of the persistent patch convolution (last workgroup of each output-channel tile); 0: always as the kp_bn_finalize launch behind it (A/B) */
#define KP_CFG_FIN_NT 17       /* 256 (default since round 6) / 1024: threads per workgroup of the one-launch statistics / backward-sums finalize of the > 32-channel layers.  A 16-wave workgroup needs four free wave slots on every SIMD of one CU and waited 100-430 us for them beside the other streams' persistent kernels (profiles/r05v_F_timeline_streams1.txt:32); four waves of ~20 registers are resident at once.  Model F step -0.07 ms (5 of 6 alternations on two boxes, profiles/r06_fin_fold_ab.txt) */
#define KP_CFG_CUS 6          /* CU budget the persistent kernels size their grids to (launches on a CU-masked stream); 0 = whole device */
#define KP_CFG_TRACE 2        /* device buffer of 512 int64: s_memtime stamps of workgroup 0 of the patch conv kernel */
int kp_config_ptr(int key, void* value);

/* ---- measurement (bench.py roofline leg; not used by the model code) ------- */
#define KP_PROF_CONV_GENERIC 0   /* fp32-accumulate VALU implicit-GEMM conv (fwd + dgrad) */
#define KP_PROF_WGRAD_GENERIC 1
#define KP_PROF_CONV_MFMA 2      /* bf16 MFMA implicit-GEMM conv (fwd + dgrad) */
#define KP_PROF_WGRAD_MFMA 3
#define KP_PROF_CONV_FIRST 4     /* Cin <= 4 first-layer VALU kernels (fwd + wgrad), HBM-bound */
#define KP_PROF_CONV_SMALL 5     /* C, K in {16, 32} small-channel kernels, forward / data gradient (incl. the BatchNorm-input form) */
#define KP_PROF_WGRAD_SMALL 9    /* ... their weight gradient (reported apart since round 3: one wave per SIMD, a different bound) */
#define KP_PROF_CONV_F32MFMA 7   /* strict-fp32 conv on v_mfma_f32_32x32x2_f32 (fwd + dgrad) */
#define KP_PROF_WGRAD_F32MFMA 8
#define KP_PROF_BOTTLENECK 6     /* fused keypoint bottleneck (kp_keypoint_bottleneck_fwd / _bwd), HBM-bound; flops = 0 */
/* Start (on=1, resets) / stop (on=0) recording a HIP-event pair around every convolution launch,
 * on the launch stream. */
int kp_prof_enable(int on);
/* Synchronise the recorded events of one kernel family and return summed time (ms), algorithmic
 * FLOPs and launch count since kp_prof_enable(1).  Host-synchronising. */
int kp_prof_collect(int family, double* ms, double* flops, long long* launches);
/* Summed algorithmic HBM bytes (operands + result, each counted once) of the recorded launches of a family. */
int kp_prof_bytes(int family, double* bytes);
/* kp_prof_enable(2): no events -- every kp_* entry point adds the algorithmic HBM bytes (operands + results, each once) of the
 * launch it makes to one running total; kp_prof_step_bytes returns the total and the number of launches since then (bench.py: the
 * step-level HBM fraction of the HBM-bound configurations = bytes per step / step time / peak).  kp_prof_enable(0) ends it. */
int kp_prof_step_bytes(double* bytes, long long* launches);

/* ---- layout ------------------------------------------------------------ */
/* NCHW fp32 -> NHWC T with a replicated border of `pad` pixels.
 * Replaces nn.ReplicationPad2d(1) in front of the first conv of a Unit
 * (knn.py:115) plus the layout change into our internal format.
 * Cp >= C is the channel count of dst: channels C..Cp-1 are written as zeros (Cp > C pads an odd width such as the
 * KeyNet decoder's z+K = 74 input channels, keynet.py:55, up to a multiple of 64 so that the MFMA kernels take the
 * layer; kp_pack_weights / kp_wgrad_unpack / kp_nhwc_grad_to_nchw take the same Cp). */
int kp_nchw_to_nhwc(const float* src, void* dst, int N, int C, int Cp, int H, int W, int pad, int dtype, void* stream);
/* Adjoint of the above: gsrc is the gradient w.r.t. the padded image
 * [N][H+2pad][W+2pad][C] (T); border contributions are folded back onto the
 * edge pixels and the result is written NCHW fp32. */
int kp_nhwc_grad_to_nchw(const void* gsrc, float* dst, int N, int C, int Cp, int H, int W, int pad, int dtype,
                         void* stream);

/* The two above for an input that the reference forms with torch.cat((z, m), dim=1) in front of a Unit -- KeyNet.forward,
 * keynet.py:29: the encoder's z [N][C0][H][W] and the Gaussian maps m [N][C1][H][W], decoder input width C0 + C1 = 74
 * (keynet.py:55) -- so that the concatenated fp32 tensor is never written: channels [0, C0) of dst come from src0, [C0, C0 + C1)
 * from src1, [C0 + C1, Cp) are zeros; same values as kp_nchw_to_nhwc of the concatenation.  The adjoint splits the folded gradient
 * back into the two inputs' (dst0 / dst1 NULL: that input needs no gradient); same sums as kp_nhwc_grad_to_nchw. */
int kp_nchw2_to_nhwc(const float* src0, int C0, const float* src1, int C1, void* dst, int N, int Cp, int H, int W, int pad, int dtype,
                     void* stream);
int kp_nhwc_grad_to_nchw2(const void* gsrc, float* dst0, int C0, float* dst1, int C1, int N, int Cp, int H, int W, int pad, int dtype,
                          void* stream);

/* ---- the image-sized first layer with its conv output recomputed, not stored ---------------------------------------------------
 * A Unit's first layer -- ReplicationPad2d(1) + Conv2d(Cin <= 3 -> 64, 3x3) + BatchNorm2d (train) + LeakyReLU, knn.py:115-119 -- is bound
 * by streaming its [N][H][W][64] output, which the plain entry points move five times per call (kp_conv_fwd writes y; kp_bn_act_fwd
 * reads it and writes the next layer's input; both BatchNorm-backward passes read it again; the weight gradient reads dy).  The input
 * image is 48 x smaller and 27 x 64 MACs per pixel are four MFMAs per 32 pixels: these entry points RECOMPUTE y wherever it is needed,
 * and neither y nor dy ever exists in memory.  x [N][H+2][W+2][C] is the padded image in the storage type (kp_nchw_to_nhwc), w the
 * forward pack [64][3][3][C] (kp_pack_weights); every call streams at most one 64-channel tensor.
 *   kp_conv_first_stats      == the statistics of kp_conv_fwd(x, w, bias, stats): rows = kp_conv_first_fused_rows, each [2][64]
 *   kp_conv_first_fwd_bnact  == kp_conv_fwd + kp_bn_act_fwd(tail none, out_pad 1): a_out [N][H+2][W+2][64] with its replicate border;
 *                               bit-identical to the two calls given the same mean / invstd
 *   kp_conv_first_bwd_reduce == kp_bn_act_bwd_reduce(g, gp = 1, tail none, y = conv(x)): rows = kp_conv_first_fused_rows of [2][64]
 *                               for kp_bn_bwd_finalize; g = dL/da over the padded domain [N][H+2][W+2][64] (kp_conv_dgrad of the layer above)
 *   kp_conv_first_bwd_wgrad  == kp_bn_act_bwd_apply + kp_conv_wgrad: scratch = kp_conv_first_fused_rows partial gradients [64][3][3][C]
 *                               for kp_wgrad_unpack; sums = kp_bn_bwd_finalize's [2][64].  (The image needs no gradient: no data gradient.)
 * Same arithmetic on the same rounded values as the calls they replace (y and dy rounded to the storage type where those store
 * them); the partition of the sums differs: equal up to the order of fp32 additions.  16-bit storage, K == 64, C <= 3, W % 32 == 0;
 * otherwise kp_conv_first_fused_supported is 0 and the calls return KP_E_UNSUPPORTED without launching. */
int kp_conv_first_fused_supported(int N, int H, int W, int C, int K, int dtype);
long long kp_conv_first_fused_rows(int N, int H, int W, int C, int K, int dtype);
int kp_conv_first_stats(const void* x, const void* w, const float* bias, float* stats, int N, int H, int W, int C, int K, int dtype,
                        void* stream);
int kp_conv_first_fwd_bnact(const void* x, const void* w, const float* bias, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, float slope, void* a_out, int N, int H, int W, int C, int K, int dtype, void* stream);
int kp_conv_first_bwd_reduce(const void* g, const void* x, const void* w, const float* bias, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, float slope, float* partials, int N, int H, int W, int C, int K,
                             int dtype, void* stream);
int kp_conv_first_bwd_wgrad(const void* g, const void* x, const void* w, const float* bias, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, float slope, const float* sums, float* scratch, int N, int H, int W,
                            int C, int K, int dtype, void* stream);

/* [N][Hs][Ws][C] -> the top-left corner of [N][Hd][Wd][C], zeros elsewhere (C * element size a multiple of 16).  With
 * kp_conv_wgrad_pad_hw: the weight gradient of a matrix-core layer (vgg.py:30-33, autograd) on a map that is not a multiple of the
 * tap-reuse kernel's 8 x 16 pixel block -- shapes no BASELINE network has, e.g. 100 x 100 inputs -- runs on zero-padded copies of
 * its two operands (the zero gradient there makes every added pixel contribute exactly 0): 46-57 -> ~900 TFLOP/s. */
int kp_pad_hw(const void* src, void* dst, int N, int Hs, int Ws, int Hd, int Wd, int C, int dtype, void* stream);

/* ---- inference: Conv2d(3x3) + eval-mode BatchNorm2d + (Leaky)ReLU of a matrix-core layer as ONE kernel (widened) -----
 * Replaces, for a network in eval() mode, the statements vgg.py:30-38 turns into (nn.Conv2d -> nn.BatchNorm2d with its running
 * statistics -> activation) -- the downstream consumers of a trained net run exactly that (atari_demo.py:35-36).  The BatchNorm is a
 * per-channel affine map there: out = act(conv(x, w) * scale[k] + shift[k]) with scale = gamma / sqrt(running_var + eps),
 * shift = beta + (conv_bias - running_mean) * scale (the caller forms the 2 K floats; no BatchNorm: scale = 1, shift = bias), applied
 * to the fp32 accumulators in the convolution's epilogue, ONE rounding to the storage type: no statistics, no BatchNorm pass, no conv
 * output in memory.  x: [N][H+2][W+2][C] T with its replicate border; w: kp_pack_weights' w_fwd [K][3][3][C]; out: the interior of
 * [N][H+2p][W+2p][K] T (p = out_pad; the ring is left untouched: kp_pad_border fills it).  16-bit storage, C and K multiples of 64,
 * H and W multiples of 16 (kp_conv_fwd_act_supported); KP_E_UNSUPPORTED otherwise -- the caller runs kp_conv_fwd + kp_bn_act_fwd.
 * kp_pad_border: the replicate border of a padded NHWC tensor from its own interior (nn.ReplicationPad2d(pad), vgg.py:30). */
int kp_conv_fwd_act_supported(int N, int H, int W, int C, int K, int dtype);
int kp_conv_fwd_act(const void* x, const void* w, const float* scale, const float* shift, float slope, void* out, int out_pad, int N,
                    int H, int W, int C, int K, int dtype, void* stream);
int kp_pad_border(void* buf, int N, int H, int W, int C, int pad, int dtype, void* stream);
int kp_conv_wgrad_pad_hw(int N, int Ho, int Wo, int C, int K, int dy_pad, int dtype, int* Hp, int* Wp);

/* ---- weights ----------------------------------------------------------- */
/* fp32 [K][C][R][R] (nn.Conv2d.weight) -> T [K][R][R][Cp] (forward operand) and,
 * if w_dgrad != NULL, T [Cp][R][R][K] with the taps flipped (dgrad operand); channels C..Cp-1 are zero. */
int kp_pack_weights(const float* w, void* w_fwd, void* w_dgrad, int K, int C, int Cp, int R, int dtype, void* stream);
/* The same for n <= 24 3x3 layers in ONE launch (all layers of a knn.Unit): host arrays of the per-layer arguments;
 * w_dgrad[i] may be NULL. */
int kp_pack_weights_batch(int n, const float* const* w, void* const* w_fwd, void* const* w_dgrad, const int* K, const int* C,
                          const int* Cp, int dtype, void* stream);
/* dw[K][C][R][R] += sum_p transpose(scratch[p][K][R][R][Cp])[.., :C]   (fp32; p < partials summed in a fixed order) */
int kp_wgrad_unpack(const float* scratch, int partials, float* dw, int K, int C, int Cp, int R, void* stream);

/* ---- convolution ------------------------------------------------------- */
/* "Valid" RxR correlation (R = 1 or 3) over an already padded image:
 *   y[n][p][q][k] = bias[k] + sum_{r,s,c} x[n][p+r][q+s][c] * w[k][r][s][c]
 * x: [N][Hi][Wi][C] T, w: [K][R][R][C] T, y: [N][Hi-R+1][Wi-R+1][K] T.
 * Forward of nn.Conv2d(k=3) after ReplicationPad2d (vgg.py:30-33, knn.py:115);
 * with x = zero-padded dy and w = the flipped/transposed pack it is the
 * data-gradient of the same layer.  If stats != NULL it receives the BatchNorm batch statistics
 * (vgg.py:35) as deterministic partial sums over disjoint groups of output pixels,
 *   stats[row][0][k] = sum_{m in row} y[m][k],  stats[row][1][k] = sum y^2   (fp32, no atomics).
 * The buffer must hold ceil(N*Ho*Wo / 64) rows; kp_last_rows() tells how many the kernel used (the persistent
 * MFMA kernel keeps running sums per workgroup and writes 4 rows per workgroup); kp_bn_finalize reduces them in fp64. */
int kp_conv_fwd(const void* x, const void* w, const float* bias, void* y, float* stats,
                int N, int Hi, int Wi, int C, int K, int R, int dtype, void* stream);
/* Data-gradient of the same layer (autograd of vgg.py:30-33): gx[N][H+2][W+2][C] (gradient over the replicate-padded
 * input domain; the consumer folds the border ring onto the edge pixels) from dy[N][H+4][W+4][K], which MUST carry
 * a zero border of 2 pixels (as written by kp_bn_act_bwd_apply with dp = 2), and w_dgrad = the flipped/transposed
 * pack [C][3][3][K] of kp_pack_weights.  Same result as kp_conv_fwd(dy, w_dgrad, NULL, gx, NULL, N, H+4, W+4, K, C, 3);
 * knowing the zero border lets the border ring be computed with 3 taps per pixel. */
int kp_conv_dgrad(const void* dy, const void* w_dgrad, void* gx, int N, int H, int W, int K, int C, int dtype,
                  void* stream);
/* Weight gradient: sum_p scratch[p][k][r][s][c] (+)= sum_{n,p,q} dy[n][p][q][k] * x[n][p+r][q+s][c]
 * x: [N][Ho+R-1][Wo+R-1][C] T; dy: interior of [N][Ho+2*dy_pad][Wo+2*dy_pad][K] T.
 * P = kp_conv_wgrad_partials(same shape arguments) tells the scratch protocol of this shape:
 *   P == 1: scratch fp32 [K][R][R][C], ZEROED by the caller (the per-tap MFMA fallback accumulates into it with fp32
 *           atomics; a single-split launch of any other kernel overwrites it);
 *   P  > 1: scratch fp32 [P][K][R][R][C], no zeroing needed: every partial is fully written with plain stores by the
 *           workgroups of one pixel split (3x3 MFMA tap-reuse kernel, generic kernel) or by one workgroup (first-layer and
 *           small-channel kernels), and kp_wgrad_unpack sums them in a fixed order (bitwise reproducible).  Every shape
 *           of the BASELINE networks, in every precision, takes this protocol. */
int kp_conv_wgrad_partials(int N, int Ho, int Wo, int C, int K, int R, int dy_pad, int dtype);
int kp_conv_wgrad(const void* x, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int R,
                  int dy_pad, int dtype, void* stream);

/* ---- 3x3 layer fed by another 3x3 layer: BatchNorm-apply + activation + replicate pad fused into the consumer -----------
 * vgg.py:30-38 builds Pad(1) -> Conv3x3 -> BatchNorm2d -> act per layer; between two such layers the reference materialises
 * act(bn(y_prev)) and its padded copy.  These entry points read the PREVIOUS layer's conv output yprev [N][H][W][C] (no
 * border) and apply  a[i][j] = act(bn(yprev[clamp(i-1)][clamp(j-1)]))  over the padded domain while staging their input,
 * so that tensor never exists.  mean/invstd/gamma/beta: the previous layer's BatchNorm (batch statistics in training, running
 * statistics in eval; all NULL = batch_norm=False, activation only), slope: its LeakyReLU slope (0 = ReLU).
 *   kp_conv_fwd_bnin  == kp_bn_act_fwd(yprev -> a, KP_TAIL_NONE, out_pad 1) + kp_conv_fwd(a, ..., N, H+2, W+2, C, K, 3)
 *   kp_conv_wgrad_bnin == the same + kp_conv_wgrad(a, dy, ...)
 * bit for bit (same expression, same rounding to the storage type, same kernels behind the staging).  Statistics rows /
 * partial-gradient protocol: kp_conv_fwd_stats_rows(N, H+2, W+2, C, K, 3) / kp_conv_wgrad_partials(N, H, W, C, K, 3, dy_pad).
 * kp_conv_bnin_supported: 1 if both fused forms exist for the shape (today: C, K in {16, 32}, 16-bit storage: the Pong-class
 * nets of vgg.py:53,66, whose layers are HBM-bound); otherwise the calls return KP_E_UNSUPPORTED and launch nothing. */
int kp_conv_bnin_supported(int N, int H, int W, int C, int K, int dtype);
int kp_conv_fwd_bnin(const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     float slope, const void* w, const float* bias, void* y, float* stats, int N, int H, int W, int C, int K,
                     int dtype, void* stream);
int kp_conv_wgrad_bnin(const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta,
                       float slope, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int dy_pad, int dtype,
                       void* stream);

/* The data gradient of a 3x3 layer L that also leaves the first pass of layer L - 1's BatchNorm backward (the layer whose padded
 * activation was L's input; vgg.py:30-38 pairs).  gx: exactly kp_conv_dgrad's output [N][H+2][W+2][C] (padded domain, bit for
 * bit).  partials: what kp_bn_act_bwd_reduce(gx, gp = 1, tail none, yred, ...) would leave -- rows x [2][C] sums of
 * gh = fold(gx) * act'(z) and gh * xhat over yred = layer L - 1's conv output [N][H][W][C] with ITS BatchNorm (mean / invstd /
 * gamma / beta, all NULL without BatchNorm) and activation slope -- computed in the padded domain (folding is the adjoint of
 * the replicate pad: every element of gx meets the factors of its clamped source pixel), equal up to the order of the fp32
 * additions; kp_bn_bwd_finalize takes the rows.  Row count: kp_conv_dgrad_bnred_rows (a pure function of the shape and the
 * device; the caller sizes `partials` with it).  kp_conv_dgrad_bnred_supported: 1 where the fused kernel exists (16-bit storage;
 * C, K in {16, 32} and a band of rows fits the LDS: kp_conv_small.hip; or C, K multiples of 64 on 16x16-tileable maps: the
 * persistent LDS-patch kernel's epilogue + the 3-tap ring kernel's, kp_conv_patch.hip / kp_conv_mfma.hip, where every workgroup
 * keeps one output-channel tile); otherwise KP_E_UNSUPPORTED, nothing launched. */
int kp_conv_dgrad_bnred_supported(int N, int H, int W, int K, int C, int dtype);
long long kp_conv_dgrad_bnred_rows(int N, int H, int W, int K, int C, int dtype);
int kp_conv_dgrad_bnred(const void* dy, const void* w_dgrad, void* gx, const void* yred, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, float slope, float* partials, int N, int H, int W, int K, int C,
                        int dtype, void* stream);

/* ---- BatchNorm(train) + activation (+ pool / upsample) ----------------- */
/* fp64 workspace size (in doubles) needed by kp_bn_finalize / kp_bn_bwd_finalize for C channels. */
int kp_bn_ws_doubles(int C);
/* stats rows (from kp_conv_fwd; rows = kp_last_rows()) -> batch mean / 1/sqrt(var+eps); updates running stats
 * with momentum (unbiased var) like nn.BatchNorm2d in train mode.  count = N*H*W.
 * Deterministic fp64 reduction (one fused launch for rows <= 8192 -- a wide kernel above 128 rows, and for C <= 32 at every
 * row count --, else two-stage through `ws`, kp_bn_ws_doubles(C) doubles). */
int kp_bn_finalize(const float* stats, long long rows, double* ws, float* save_mean, float* save_invstd, float* running_mean,
                   float* running_var, long long* num_batches_tracked, float momentum, float eps, long long count,
                   int C, void* stream);
/* kp_conv_fwd(..., stats) + kp_bn_finalize(stats, kp_last_rows(), ...) as ONE call: the pair vgg.py:33-35 (Conv2d, BatchNorm2d in training
 * mode) turns into.  Where the convolution runs on the persistent patch kernel with one statistics row per workgroup (model F's layers at
 * the BASELINE batches), the finalize runs in the TAIL of the convolution -- the last workgroup of each output-channel tile to arrive
 * (a device-scope ticket per tile) sums that tile's rows in a fixed order in fp64 and writes mean / invstd / running statistics for
 * its channels -- so the dependent kp_bn_finalize launch (5-7 us of work that waits 100-400 us for a CU beside the other streams'
 * persistent kernels, profiles/r05v_F_timeline_streams1.txt) does not exist; anywhere else the call launches kp_bn_finalize itself.
 * Same numbers either way up to the order of the fp64 additions (<= 1 ulp of the fp32 results); deterministic.
 * tickets: KP_FIN_TICKETS (64) zero-initialised 32-bit counters the call may use and leaves zero (one buffer per stream: launches on
 * one stream may share it); NULL: never fold.  ws: as kp_bn_finalize.  kp_conv_fwd_fin_folds: 1 if this shape folds (pure query). */
#define KP_FIN_TICKETS 64
int kp_conv_fwd_fin(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi, int C, int K, int R,
                    int dtype, double* ws, float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                    long long* num_batches_tracked, float momentum, float eps, long long count, unsigned* tickets, void* stream);
int kp_conv_fwd_fin_folds(int N, int Hi, int Wi, int C, int K, int R, int dtype);
/* out = tail(act((y - mean) * invstd * gamma + beta)); act = LeakyReLU(slope) (slope 0 = ReLU).
 * y: [N][H][W][C] T.  out: [N][Ho+2*out_pad][Wo+2*out_pad][C] T with replicated border where
 * (Ho,Wo) = (H,W) | (H/2,W/2) | (2H,2W) for tail none | maxpool2 | upsample2.
 * argmax (maxpool only): u8 [N][H/2][W/2][C] = winning position 0..3 (row-major in the window).
 * If mean == NULL the affine step is skipped (conv + activation without BN). */
int kp_bn_act_fwd(const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  float slope, int tail, void* out, int out_pad, unsigned char* argmax,
                  int N, int H, int W, int C, int dtype, void* stream);
/* Backward, pass 1: partial sums over disjoint groups of pixels: partials[r][0][c] = sum g*act'(z),
 * partials[r][1][c] = sum g*act'(z)*xhat  (fp32, buffer of [N*H][2][C], no atomics; kp_last_rows() rows are used).
 * The incoming gradient g w.r.t. the activation output of this layer is read through `gsrc`:
 *   tail none    : gsrc = [N][H+2gp][W+2gp][C] T, gp in {0,1}; gp=1 folds the border (adjoint of replicate pad)
 *   tail maxpool2: gsrc = [N][H/2+2gp][W/2+2gp][C] T routed through argmax
 *   (upsample: run kp_upsample2_bwd first, then tail none / gp = 0). */
int kp_bn_act_bwd_reduce(const void* gsrc, int gp, int tail, const unsigned char* argmax, const void* y,
                         const float* mean, const float* invstd, const float* gamma, const float* beta,
                         float slope, float* partials, int N, int H, int W, int C, int dtype, void* stream);
/* partials -> sums[0..C) = dbeta, sums[C..2C) = dgamma (fp32, overwritten); dbeta[] += , dgamma[] += if non-NULL. */
int kp_bn_bwd_finalize(const float* partials, double* ws, long long rows, int C, float* sums, float* dgamma,
                       float* dbeta, void* stream);
/* Backward, pass 2: dy = gamma*invstd*(ghat - dbeta/M - xhat*dgamma/M) written into
 * [N][H+2dp][W+2dp][C] T with a ZERO border of dp pixels. */
int kp_bn_act_bwd_apply(const void* gsrc, int gp, int tail, const unsigned char* argmax, const void* y,
                        const float* mean, const float* invstd, const float* gamma, const float* beta,
                        float slope, const float* sums, void* dy, int dp,
                        int N, int H, int W, int C, int dtype, void* stream);
/* Adjoint of bilinear x2 (align_corners=True): gsrc = [N][2H+2gp][2W+2gp][C] T (folded if gp=1)
 * -> g [N][H][W][C] T. */
int kp_upsample2_bwd(const void* gsrc, int gp, void* g, int N, int H, int W, int C, int dtype, void* stream);

/* ---- 1x1 head: Conv2d(C->K, 1) + LeakyReLU (knn.py:123-125) ------------- */
/* x: [N][H][W][C] T (no border); w fp32 [K][C]; out fp32 NCHW [N][K][H][W]. */
int kp_head_fwd(const void* x, const float* w, const float* bias, float slope, float* out,
                int N, int H, int W, int C, int K, int dtype, void* stream);
/* gout/out fp32 NCHW; dx [N][H][W][C] T (may be NULL); dw [K][C] +=, db [K] += (fp32).
 * ws: fp32 workspace of kp_head_bwd_ws_floats(...) floats (contents irrelevant) -> the weight / bias gradient is reduced
 * in two stages with a fixed summation order (one partial gradient per workgroup, then their sum): bitwise reproducible.
 * ws == NULL: one stage with fp32 atomics (faster by a few microseconds, run-to-run differences in the last bits). */
long long kp_head_bwd_ws_floats(int N, int H, int W, int C, int K, int dtype);
int kp_head_bwd(const float* gout, const float* out, const void* x, const float* w, float slope,
                void* dx, float* dw, float* db, float* ws, int N, int H, int W, int C, int K, int dtype, void* stream);

/* The head of a Unit fed by its last 3x3 layer's convolution OUTPUT (knn.py:110-130: core ... -> out_block): BatchNorm2d
 * (train: batch statistics) + (Leaky)ReLU of that layer (vgg.py:30-38) run on the head's load path -- the layer's activation
 * tensor is never written.  yprev [N][H][W][C] T (no border); mean / invstd / gamma / beta fp32 [C] (all NULL: no BatchNorm);
 * slope_in: the producing layer's activation slope, 0 <= slope_in <= 1.
 *   kp_head_fwd_bnin == kp_bn_act_fwd(yprev, ..., slope_in, tail none, a, out_pad 0) + kp_head_fwd(a, ...)
 *   kp_head_bwd_bnin == the same + kp_head_bwd(gout, out, a, ...)
 * bit for bit (same expression, same rounding to the storage type).  kp_head_bnin_supported: 1 if both exist for the shape
 * (16-bit storage, and the kernel kp_head_fwd picks for the shape has the fused form: the 32-channel matrix-core head --
 * kp_head_mfma.hip -- or the K <= 16 kernels with C a power of two <= 512 -- kp_head_fast.hip; the wide matrix-core heads
 * do not); otherwise the calls return KP_E_UNSUPPORTED and launch nothing.
 * Workspace: kp_head_bwd_ws_floats of the same shape. */
int kp_head_bnin_supported(int N, int H, int W, int C, int K, int dtype);
int kp_head_fwd_bnin(const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     float slope_in, const float* w, const float* bias, float slope, float* out, int N, int H, int W, int C,
                     int K, int dtype, void* stream);
int kp_head_bwd_bnin(const float* gout, const float* out, const void* yprev, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, float slope_in, const float* w, float slope, void* dx, float* dw,
                     float* db, float* ws, int N, int H, int W, int C, int K, int dtype, void* stream);

/* ---- inference-only Unit forward as one kernel -------- */
/* `net.keypoint(x)` of the reference's consumers (atari_demo.py:35-36, tests/test_gl_hack.py:454-456): a Unit
 * (keypoints/models/knn.py:110-130) in eval mode whose core has no pooling / upsampling (the Pong-class tables of
 * keypoints/models/vgg.py:64-70): L x [ReplicationPad2d(1), Conv2d 3x3, BatchNorm2d(eval), (Leaky)ReLU], then
 * Conv2d 1x1 + LeakyReLU.  fp32 throughout.
 * x fp32 NCHW [B][Cin][H][W] -> heat fp32 NCHW [B][K][H][W].
 * widths (HOST) [L+1]: output channels of the 3x3 layers (multiples of 8), then K.  slopes (HOST) [L+1]: negative slope
 * of each activation (0 = ReLU).
 * params (DEVICE fp32, kp_unit_infer_param_floats(...) floats): the folded, kernel-ordered weights, written layer by
 * layer by kp_unit_infer_pack -- layer = 0..L-1 for the 3x3 layers, L for the head -- from the tensors of the Unit's
 * state_dict as they are: w = Conv2d.weight [K][C][3][3] ([K][C] for the head), b = Conv2d.bias or NULL, and the
 * BatchNorm2d weight / bias / running_mean / running_var [K] (var == NULL: the layer has no BatchNorm).  The fold is
 * w * gamma / sqrt(var + eps), (b - mean) * gamma / sqrt(var + eps) + beta in fp32.  Pack again after the parameters change.
 * KP_E_UNSUPPORTED: L outside 1..6, a 3x3 width that is not a multiple of 8, a last 3x3 width other than 8 / 16 / 32, or
 * a chain too wide for one workgroup's LDS. */
long long kp_unit_infer_param_floats(int Cin, int L, const int* widths);
int kp_unit_infer_pack(float* params, int layer, int Cin, int L, const int* widths, const float* w, const float* b,
                       const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                       void* stream);
int kp_unit_infer_fwd(const float* x, const float* params, float* heat, int B, int Cin, int H, int W, int L,
                      const int* widths, const float* slopes, void* stream);
/* The consumer's whole call, `spacial_logsoftmax(net.keypoint(x))` (atari_demo.py:35-36): kp_unit_infer_fwd followed by
 * kp_spatial_softmax_fwd(log_domain = 1) on the same stream; k [B][K][2] (y, x), p [B][K][H], q [B][K][W]. */
int kp_unit_infer_keypoints_fwd(const float* x, const float* params, float* heat, float* k, float* p, float* q, int B,
                                int Cin, int H, int W, int L, const int* widths, const float* slopes, void* stream);

/* ---- keypoint bottleneck (keypoints/models/functional.py) --------------- */
/* heat fp32 [BK][H][W] -> k [BK][2] (y,x), p [BK][H], q [BK][W].
 * log_domain=1: spacial_logsoftmax (functional.py:37-44); 0: spacial_softmax (:27-34). */
int kp_spatial_softmax_fwd(const float* heat, float* k, float* p, float* q, int BK, int H, int W,
                           int log_domain, void* stream);
int kp_spatial_softmax_bwd(const float* gk, const float* k, const float* p, const float* q, float* dheat,
                           int BK, int H, int W, void* stream);
/* m[bk][i][j] = exp(-sqrt((r_i-ky)^2 + (c_j-kx)^2 + eps) / (2 sigma^2))   functional.py:56-63 */
int kp_gaussian_fwd(const float* k, float* m, int BK, int H, int W, float sigma, float eps, void* stream);
int kp_gaussian_bwd(const float* gm, const float* k, float* dk, int BK, int H, int W, float sigma, float eps,
                    void* stream);
/* -( (r_i-ky)^2 + (c_j-kx)^2 )   functional.py:66-72 */
int kp_point_map_fwd(const float* k, float* m, int BK, int H, int W, void* stream);
/* Transport, combine_method='max' (transporter.py:57-60):
 *   Ms = max_k m_s, Mt = max_k m_t, phi = phi_s*(1-Ms)*(1-Mt) + phi_t*Mt.   All fp32 NCHW.
 * arg_s/arg_t: u8 [B][H][W] argmax over K (first index wins, like torch.max). */
int kp_transport_max_fwd(const float* phi_s, const float* phi_t, const float* m_s, const float* m_t,
                         float* phi, float* mask_s, float* mask_t, unsigned char* arg_s, unsigned char* arg_t,
                         int B, int C, int K, int H, int W, void* stream);
/* Any of the d_* outputs may be NULL (not needed). d_m_* are fully written (zeros off the argmax). */
int kp_transport_max_bwd(const float* gphi, const float* phi_s, const float* phi_t, const float* mask_s,
                         const float* mask_t, const unsigned char* arg_s, const unsigned char* arg_t,
                         float* d_phi_s, float* d_phi_t, float* d_m_s, float* d_m_t,
                         int B, int C, int K, int H, int W, void* stream);

/* The other combine modes of TransporterNet.forward (transporter.py:41-55), forward and adjoint, all fp32 NCHW:
 *   KP_COMBINE_SUM_AND_CLAMP  mask = clamp(sum_k m, 0, 1)                                   transporter.py:47-50
 *   KP_COMBINE_PRETRAINED     mask = m (K == 1: the MaskMaker output, transporter.py:83-92)  transporter.py:52-55
 *   KP_COMBINE_LOOP           phi <- phi (1 - m_s,i)(1 - m_t,i) + phi_t m_t,i, i = 0..K-1     transporter.py:41-45
 * then (modes 1, 2) phi = phi_s (1 - mask_s)(1 - mask_t) + phi_t mask_t.  mask_s / mask_t [B][H][W] are what forward()
 * returns as mask_xs / mask_xt (the LAST per-keypoint slices in 'loop' mode, as in the reference).
 * Backward: any d_* may be NULL; d_m_* are [B][K][H][W] (clamp passes the gradient where 0 <= sum <= 1, like torch). */
#define KP_COMBINE_SUM_AND_CLAMP 1
#define KP_COMBINE_PRETRAINED 2
#define KP_COMBINE_LOOP 3
int kp_transport_fwd(int mode, const float* phi_s, const float* phi_t, const float* m_s, const float* m_t, float* phi,
                     float* mask_s, float* mask_t, int B, int C, int K, int H, int W, void* stream);
int kp_transport_bwd(int mode, const float* gphi, const float* phi_s, const float* phi_t, const float* m_s,
                     const float* m_t, float* d_phi_s, float* d_phi_t, float* d_m_s, float* d_m_t, int B, int C, int K,
                     int H, int W, void* stream);

/* The whole bottleneck of TransporterNet.forward for combine_method='max' in ONE launch (SURVEY.md 8b
 * keypoint_bottleneck_fwd): spacial_logsoftmax of both branches' heat-maps (functional.py:37-44), gaussian_like_function
 * of both keypoint sets (functional.py:56-63; key2map, transporter.py:31), max over K and the blend
 * (transporter.py:57-60).  All fp32 NCHW.  heat_s/heat_t [B][K][H][W], phi_s/phi_t [B][C][H][W]; outputs: k_t [B][K][2]
 * (y,x), p_t [B][K][H], q_t [B][K][W], m_t [B][K][H][W], phi [B][C][H][W], mask_s/mask_t [B][H][W], arg_t u8 [B][H][W]
 * (argmax_k m_t, first index wins).  The source branch's maps are consumed inside the kernel and never written
 * (TransporterNet.forward does not return them).  Same results as the piecewise entry points above, bit for bit.
 * H, W <= 256. */
int kp_keypoint_bottleneck_fwd(const float* heat_s, const float* heat_t, const float* phi_s, const float* phi_t,
                               float* k_t, float* p_t, float* q_t, float* m_t, float* phi, float* mask_s, float* mask_t,
                               unsigned char* arg_t, int B, int C, int K, int H, int W, float sigma, float eps,
                               void* stream);
/* Its adjoint w.r.t. the target branch (the source branch runs under no_grad, transporter.py:36-37):
 * gphi = dL/dphi [B][C][H][W]; optional g_k = dL/dk_t [B][K][2] and g_m = dL/dm_t [B][K][H][W] (NULL = none) are
 * added where they enter the chain.  Outputs d_phi_t [B][C][H][W] and d_heat_t [B][K][H][W] (fully written).
 * Needs G*H*W <= 16384 with G = min(C, max(1, 1024 / (H*W/4))) (every BASELINE shape: 16x16, 32x32, 84x84). */
int kp_keypoint_bottleneck_bwd(const float* gphi, const float* g_k, const float* g_m, const float* phi_s,
                               const float* phi_t, const float* mask_s, const float* mask_t, const unsigned char* arg_t,
                               const float* k_t, const float* p_t, const float* q_t, float* d_phi_t, float* d_heat_t,
                               int B, int C, int K, int H, int W, float sigma, float eps, void* stream);

/* ---- 16-bit NHWC ("T") interchange between an encoder head, the keypoint bottleneck and the decoder -----------------------
 * The reference's Units exchange fp32 NCHW tensors (knn.py:127-130; TransporterNet.forward, transporter.py:34-64: feature
 * encoder -> 'max' blend -> decoder).  Inside this library a Unit computes in T NHWC; at 16 feature channels over 84 x 84 maps
 * (the Pong-class nets, vgg.py:53,66) the fp32 planes at the Unit boundaries and the two layout kernels around the decoder are a
 * measurable share of the step.  These entry points keep the same arithmetic and move the boundary tensors in T NHWC instead:
 *   kp_head_fwd_t        == kp_head_fwd / kp_head_fwd_bnin with the output rounded once to T: out_t [N][H][W][K]
 *   kp_head_bwd_t        == kp_head_bwd / kp_head_bwd_bnin reading the output gradient and the saved output as T [N][H][W][K]
 *                           (bnin != 0: x is the producing 3x3 layer's conv output and mean .. slope_in its BatchNorm / activation,
 *                           as in kp_head_fwd_bnin; bnin == 0: x is the stored activation, the five arguments are ignored)
 *   kp_keypoint_bottleneck_fwd_t == kp_keypoint_bottleneck_fwd with phi_s / phi_t T [B][H][W][C] and the blend written as the
 *                           decoder's first-layer input a [B][H+2][W+2][C] T, replicate border included (knn.py:115: ReplicationPad2d(1));
 *                           keypoints, marginals, masks, argmax and m_t are bit-identical to kp_keypoint_bottleneck_fwd on the same
 *                           values, the blend is its value rounded once to T
 *   kp_keypoint_bottleneck_bwd_t == kp_nhwc_grad_to_nchw (the fold of the replicate border) + kp_keypoint_bottleneck_bwd, with
 *                           ga [B][H+2][W+2][C] T = the decoder's data gradient over the padded domain and d_phi_t [B][H][W][C] T
 *   kp_nhwc_to_nchw      interior of a padded T image -> fp32 NCHW (TransporterNet.forward also RETURNS the blend, transporter.py:64)
 * *_supported: 1 where the form exists (16-bit storage; head: C == 32, K == 16, H*W % 8 == 0, N*H*W % 32 == 0; bottleneck:
 * C in {8, 16, 32}, H, W <= 256, H*W <= 16384); otherwise the calls return KP_E_UNSUPPORTED and launch nothing, and the caller
 * runs the fp32 NCHW entry points.  Head workspace: kp_head_bwd_ws_floats of the same shape. */
int kp_head_t_supported(int N, int H, int W, int C, int K, int dtype);
int kp_head_fwd_t(const void* x, int bnin, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  float slope_in, const float* w, const float* bias, float slope, void* out_t, int N, int H, int W, int C, int K,
                  int dtype, void* stream);
int kp_head_bwd_t(const void* gout_t, const void* out_t, const void* x, int bnin, const float* mean, const float* invstd,
                  const float* gamma, const float* beta, float slope_in, const float* w, float slope, void* dx, float* dw, float* db,
                  float* ws, int N, int H, int W, int C, int K, int dtype, void* stream);
int kp_keypoint_bottleneck_t_supported(int B, int C, int K, int H, int W, int dtype);
int kp_keypoint_bottleneck_fwd_t(const float* heat_s, const float* heat_t, const void* phi_s, const void* phi_t, float* k_t,
                                 float* p_t, float* q_t, float* m_t, void* a, float* mask_s, float* mask_t, unsigned char* arg_t,
                                 int B, int C, int K, int H, int W, float sigma, float eps, int dtype, void* stream);
int kp_keypoint_bottleneck_bwd_t(const void* ga, const float* g_k, const float* g_m, const void* phi_s, const void* phi_t,
                                 const float* mask_s, const float* mask_t, const unsigned char* arg_t, const float* k_t,
                                 const float* p_t, const float* q_t, void* d_phi_t, float* d_heat_t, int B, int C, int K, int H,
                                 int W, float sigma, float eps, int dtype, void* stream);
int kp_nhwc_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int pad, int dtype, void* stream);

/* ---- two half-batches in one launch ("pair" forms) ------------------------------------------------------------------------
 * TransporterNet.forward runs every encoder twice on the same weights: the source frame under no_grad, then the target frame
 * (transporter.py:36-39); nn.BatchNorm2d in training mode takes the batch statistics of EACH call and updates its running
 * statistics twice.  The pair forms take both half-batches as one tensor [N = 2 x N/2] (half 0 = the first call's images) and launch
 * each half exactly as a separate call of N / 2 images would be launched -- same grid, same work per workgroup, same statistics
 * rows -- in ONE grid: results are bit-identical to the two calls, there are half as many launches, each twice as long.
 *   kp_conv_fwd_pair       == kp_conv_fwd (R = 3, C <= 4: a Unit's first layer) on each half; stats = [2][rows][2][K] with
 *                             rows = kp_conv_fwd_stats_rows(N / 2, Hi, Wi, C, K, 3, dtype)
 *   kp_conv_fwd_bnin_pair  == kp_conv_fwd_bnin on each half; mean / invstd [2][C] (gamma / beta [C]: one layer);
 *                             rows = kp_conv_fwd_stats_rows(N / 2, H + 2, W + 2, C, K, 3, dtype)
 *   kp_bn_finalize_pair    == kp_bn_finalize on each half's rows, half 0 first: save_mean / save_invstd [2][C]; running statistics
 *                             and num_batches_tracked take both updates in order.  count = pixels of ONE half.
 *   kp_head_fwd_bnin_pair  == kp_head_fwd_bnin (t_out = 0: out0 / out1 fp32 NCHW [N/2][K][H][W]) or kp_head_fwd_t with bnin = 1
 *                             (t_out = 1: T [N/2][H][W][K]) on each half
 * *_pair_supported: 1 where the form exists (16-bit storage; conv: first = 1 the first-layer VALU kernel, first = 0 the C, K in
 * {16, 32} band kernel; head: C == 32 on the matrix cores; finalize: rows <= 8192, C <= 32); otherwise KP_E_UNSUPPORTED,
 * nothing launched, and the caller makes the two calls. */
int kp_conv_pair_supported(int N, int H, int W, int C, int K, int first, int dtype);
int kp_conv_fwd_pair(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi, int C, int K,
                     int dtype, void* stream);
int kp_conv_fwd_bnin_pair(const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta,
                          float slope, const void* w, const float* bias, void* y, float* stats, int N, int H, int W, int C, int K,
                          int dtype, void* stream);
int kp_bn_finalize_pair_supported(long long rows, int C);   /* 1: kp_bn_finalize_pair takes `rows` rows per half of C channels */
int kp_bn_finalize_pair(const float* stats, long long rows, float* save_mean, float* save_invstd, float* running_mean,
                        float* running_var, long long* num_batches_tracked, float momentum, float eps, long long count, int C,
                        void* stream);
int kp_head_pair_supported(int N, int H, int W, int C, int K, int t_out, int dtype);
int kp_head_fwd_bnin_pair(const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta,
                          float slope_in, const float* w, const float* bias, float slope, void* out0, void* out1, int t_out, int N,
                          int H, int W, int C, int K, int dtype, void* stream);

/* ---- input-side augmentation (keypoints/tps.py, keypoints/data_augments.py) ----------------------- */
/* out = F.grid_sample(x, tps_grid(theta, ctrl, x.shape), padding_mode='zeros')   tps.py:60-87,128-131.
 * x/out fp32 NCHW; theta [N][T+3][2] (reduced != 0: [N][T+2][2], tps.py:48-50); ctrl [N][T][2] or, with
 * ctrl_batched == 0, [T][2] shared by the batch.  The sampling grid is evaluated per pixel, never materialised. */
int kp_warp_tps(const float* x, const float* theta, const float* ctrl, float* out, int N, int C, int H, int W, int T,
                int reduced, int ctrl_batched, void* stream);
/* out = F.grid_sample(x, F.affine_grid(mat, x.shape), padding_mode='zeros'); mat [N][2][3]   tps.py:154-166 */
int kp_warp_affine(const float* x, const float* mat, float* out, int N, int C, int H, int W, void* stream);
/* the same for per-sample ROTATIONS given as angles [N] (radians) on the device: the matrix [[cos, sin, 0], [-sin, cos, 0]] of
 * rotate_affine_grid_multi (tps.py:154-166) is formed in the kernel instead of on the host */
int kp_warp_rotate(const float* x, const float* angle, float* out, int N, int C, int H, int W, void* stream);

/* ---- loss and optimiser -------------------------------------------------- */
/* l2_reconstruction_loss (transporter.py:56-60): loss_sum[0] += sum (x-t)^2*mask (fp64, zeroed by caller);
 * grad = 2*(x-t)*mask*scale  (scale = 1/numel).  mask may be NULL. */
int kp_l2_loss(const float* x, const float* t, const float* mask, double* loss_sum, float* grad, float scale,
               size_t n, void* stream);
/* The same in ONE launch including the scalar: loss[0] = (float)(sum / n) (the mean formed in fp64, rounded once: what
 * `((x - x_)**2 * mask).mean()` of transporter.py:56-60 returns), grad = 2*(x-t)*mask/n.  ws2: two doubles, zero before the FIRST
 * call and left zero by every call (the last workgroup to arrive writes the mean and clears them): a persistent per-stream
 * workspace, so that neither a zero-fill nor a division kernel sits between the forward and the backward of a step. */
int kp_l2_loss_mean(const float* x, const float* t, const float* mask, double* ws2, float* loss, float* grad, size_t n, void* stream);
/* y[i] = x[i] * s[0] with s a DEVICE scalar: chains the loss gradient (autograd hands the upstream gradient of the
 * scalar loss as a device tensor) without a host synchronisation.  x, y 16-byte aligned; y may alias x. */
int kp_scale(const float* x, const float* s, float* y, size_t n, void* stream);
/* torch.optim.Adam step (transporter.py:47,89) over a flat fp32 arena. bc1 = 1-b1^t, bc2 = 1-b2^t. */
int kp_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2,
                 float eps, float bc1, float bc2, void* stream);

/* ---- dynamic loss scaling for dtype KP_F16 ----------------------------------------------------------------------
 * The role of apex `amp.scale_loss(loss, optim)` + the O2 optimizer wrapper (transporter.py:51,84-89): the loss gradient
 * is multiplied by state[0]; kp_amp_check flags inf / nan in the (all-reduced) gradient arenas; kp_amp_prepare
 * advances Adam's step count and bias corrections unless the step overflowed; kp_adam_step_scaled un-scales and
 * updates (a no-op on overflow); kp_amp_update halves the scale after an overflow or multiplies it by growth_factor
 * after growth_interval clean steps, and clears the flag.  `state` is 8 DEVICE doubles:
 * {scale, clean steps, found_inf, Adam step t, 1-b1^t, sqrt(1-b2^t), steps skipped, 1/scale}; nothing here synchronises
 * with the host except kp_amp_init. */
int kp_amp_init(double* state, double init_scale, void* stream);
int kp_amp_check(const float* g, size_t n, double* state, void* stream);
/* y[i] = x[i] * (float)state[0]: `amp.scale_loss` itself (transporter.py:84-85) -- the loss on the way forward, its incoming gradient
 * on the way back -- with the scale read from the device state block. */
int kp_amp_scale(const float* x, const double* state, float* y, size_t n, void* stream);
int kp_amp_prepare(double* state, float b1, float b2, void* stream);
int kp_adam_step_scaled(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                        const double* state, void* stream);
int kp_amp_update(double* state, float growth_factor, float backoff_factor, int growth_interval, float max_scale,
                  void* stream);

/* ---- gradient exchange of the data-parallel step (SURVEY.md 8b "kp_comm_init/destroy/allreduce", 8e) ---------------
 * The reference has no distributed code; these calls are what a DDP wrapper around transporter.py:79-89 would do between
 * `loss.backward()` and `optim.step()`: one all-reduce (average) of each flat gradient arena over the ranks (RCCL over
 * xGMI), and a broadcast of parameters / BatchNorm buffers from one rank.  RCCL is bound at run time: kp_comm_available()
 * is 0 (and the other calls return KP_E_ARG with text) in a process that has no librccl.
 * One communicator per process and GPU: kp_comm_init binds it to the CURRENT HIP device; `id128` = the 128 bytes rank 0
 * got from kp_comm_unique_id, carried to the other ranks by the launcher (any side channel).  Collectives are enqueued on
 * `stream` in place (buf is input and output), never synchronise; every rank must issue the same calls in the same order.
 * dtype: KP_F32 / KP_BF16 / KP_F16 = element type of buf.  average != 0: sum / world (ncclAvg). */
int kp_comm_available(void);
int kp_comm_unique_id(void* id128);
int kp_comm_init(const void* id128, int rank, int world, void** comm);
int kp_comm_allreduce(void* comm, void* buf, size_t count, int dtype, int average, void* stream);
int kp_comm_broadcast(void* comm, void* buf, size_t count, int dtype, int root, void* stream);
/* number of ranks of the communicator (ncclCommCount): what bench.py prints as `dp.nccl_ranks` for --comm native */
int kp_comm_count(void* comm, int* ranks);
int kp_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* KP_HIP_H */
