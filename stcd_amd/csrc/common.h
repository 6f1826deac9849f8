// common.h -- shared declarations of the stcd HIP engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <string.h>

#include "../../include/stcd_hip.h"

namespace stcd {

typedef __bf16 bf16;

enum DType { F32 = STCD_DTYPE_F32, BF16 = STCD_DTYPE_BF16 };
inline size_t dsize(int dt) { return dt == BF16 ? 2 : 4; }

void set_error(const std::string& msg);

#define STCD_CHECK(cond, msg)                                              \
    do {                                                                   \
        if (!(cond)) {                                                     \
            ::stcd::set_error(std::string(__func__) + ": " + (msg));       \
            return 1;                                                      \
        }                                                                  \
    } while (0)

#define STCD_HIP(call)                                                                        \
    do {                                                                                      \
        hipError_t err_ = (call);                                                             \
        if (err_ != hipSuccess) {                                                             \
            ::stcd::set_error(std::string(__func__) + ": " #call ": " + hipGetErrorString(err_)); \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

// ------------------------------------------------------------------ device helpers
template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void load8<bf16>(const bf16* p, float (&v)[8]) {
    uint4 r = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. every global load AND STORE
// the wave still has in flight: a persistent kernel that stores a tile and then meets at the loop barrier pays a full
// HBM write round trip (~1.5 us) per tile for nothing.  Use where the only cross-wave traffic is through LDS.
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 t;
    t[0] = (__bf16)lo;   // plain casts: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
    t[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, t);
}
template <>
__device__ __forceinline__ void store8<bf16>(bf16* p, const float (&v)[8]) {
    uint4 r;
    r.x = pack_bf16x2(v[0], v[1]); r.y = pack_bf16x2(v[2], v[3]);
    r.z = pack_bf16x2(v[4], v[5]); r.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = r;
}
// value as the kernel's activation type would store it (bf16 rounding in bf16 mode, identity in fp32 mode)
template <typename T>
__device__ __forceinline__ float round_as(float v) { return (float)(T)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------ BatchNorm sum accumulators
// Per-channel sums (forward: sum y, sum y^2; backward: sum dz, sum dz*xhat) meet in 64-bit FIXED-POINT integer
// accumulators through global atomic adds: integer addition is exact and order-independent, so the result is
// reproducible run to run (float atomics are not), and the kernel that CONSUMES the statistics derives mean / invstd /
// scale / shift (or the backward coefficients) for its own channels from 2*C integers in its prologue -- there is no
// separate "finalize" launch between the reduction and its consumer.  Every block adds one value per (group, channel,
// sum); BN_REP replicas spread the adders (<= 1/8 of the blocks meet on one address), the consumer adds the replicas.
// Layout: int64 [BN_REP][groups][2][C], zeroed by the engine (one memset per forward / per backward for ALL layers).
// Resolution / range: forward sum(y) 2^-26 / +-1.4e11, sum(y^2) 2^-18 / 3.5e13; backward sums 2^-36 / +-1.3e8 -- each
// block's contribution is an fp32 partial sum whose own rounding is coarser than these steps for any activation scale
// a BatchNorm'd network produces.
constexpr int BN_REP = 8;
constexpr float BN_FS1 = 67108864.f;       // 2^26
constexpr float BN_FS2 = 262144.f;         // 2^18
constexpr float BN_BS = 68719476736.f;     // 2^36
inline int64_t bn_acc_bytes(int groups, int C) { return (int64_t)BN_REP * groups * 2 * C * 8; }
__device__ __forceinline__ void bn_acc_add(long long* acc, int rep, int groups, int C, int g, int which, int c, double v, float scale) {
    const long long q = (long long)llrint(v * (double)scale);
    atomicAdd(reinterpret_cast<unsigned long long*>(acc + (((int64_t)(rep & (BN_REP - 1)) * groups + g) * 2 + which) * C + c),
              (unsigned long long)q);
}
__device__ __forceinline__ double bn_acc_get(const long long* __restrict__ acc, int groups, int C, int g, int which, int c, float scale) {
    long long s = 0;
#pragma unroll
    for (int r = 0; r < BN_REP; ++r) s += acc[(((int64_t)r * groups + g) * 2 + which) * C + c];
    return (double)s * (1.0 / (double)scale);      // (the scales are powers of two: exact, and no double-precision division)
}
// 1 / sqrt(x) in double without a double-precision square root or division (software sequences of ~40 instructions each on the
// consumer blocks' critical path: every block of every kernel that applies a BatchNorm builds its table first): a float seed and two
// Newton steps -- 2^-23 -> 2^-45 -> 2^-89 relative error
__device__ __forceinline__ double bn_rsqrt(double x) {
    double y = (double)__frsqrt_rn((float)x);
    y = y * (1.5 - 0.5 * x * y * y);
    return y * (1.5 - 0.5 * x * y * y);
}

// ---- consumer-side "finalize": every block of a kernel that APPLIES a BatchNorm derives the per-channel constants of all
//      its channels from the integer accumulators (2*C*BN_REP loads per block) into an LDS table; block 0 also publishes
//      them (stat, running statistics / dgamma, dbeta) for later kernels.  Same arithmetic as torch: biased variance for
//      the normalisation, unbiased for running_var, momentum 0.1, double precision for the moments.
// tab: [groups][2][C] = scale, shift.  Thread c handles channel c for all groups IN ORDER (the shared encoder BatchNorm
// sees date 0 then date 1: /root/reference/models/SiamUnet_diff.py:99,123).
__device__ __forceinline__ void bn_fwd_table(float* tab, const long long* __restrict__ facc, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                             float* __restrict__ stat, int C, int groups, int64_t ppg, float momentum, float eps,
                                             bool publish, int cbase = 0, int CS = 0, int g_first = 0) {
    // channels [cbase, cbase + CS) only (CS == 0: all): wide layers give every block one 64-channel slab, so its prologue reads
    // 64 channels' accumulators instead of up to 2048
    if (CS == 0) CS = C;
    const double inv_n = 1.0 / (double)ppg;            // the ONE double-precision division of a thread's table entries
    for (int cl = threadIdx.x; cl < CS; cl += blockDim.x) {
        const int c = cbase + cl;
        if (facc) {
            const float gam = gamma[c], bet = beta[c];
            float rm = (publish && rmean) ? rmean[c] : 0.f, rv = (publish && rvar) ? rvar[c] : 0.f;
            for (int gi = 0; gi < groups; ++gi) {      // the running statistics see the groups in the order the reference calls the
                int g = gi + g_first;                   // BatchNorm on them: g_first, g_first + 1, ... (cyclic)
                if (g >= groups) g -= groups;
                const double s1 = bn_acc_get(facc, groups, C, g, 0, c, BN_FS1), s2 = bn_acc_get(facc, groups, C, g, 1, c, BN_FS2);
                const double mean = s1 * inv_n;
                double var = s2 * inv_n - mean * mean;
                if (var < 0.0) var = 0.0;
                const double invstd = bn_rsqrt(var + (double)eps);
                const float sc = (float)(gam * invstd), sh = (float)(bet - mean * gam * invstd);
                tab[(g * 2 + 0) * CS + cl] = sc;
                tab[(g * 2 + 1) * CS + cl] = sh;
                if (publish) {
                    float* st = stat + (int64_t)g * 4 * C;
                    st[c] = (float)mean; st[C + c] = (float)invstd; st[2 * C + c] = sc; st[3 * C + c] = sh;
                    const double unb = ppg > 1 ? var * ((double)ppg / (double)(ppg - 1)) : var;
                    rm = (float)((1.0 - momentum) * rm + momentum * mean);
                    rv = (float)((1.0 - momentum) * rv + momentum * unb);
                }
            }
            if (publish && rmean) rmean[c] = rm;
            if (publish && rvar) rvar[c] = rv;
        } else if (gamma) {      // eval mode: the running statistics (read-only), the arithmetic of k_bn_eval_prepare -- no launch for it
            const float invstd = 1.f / sqrtf(rvar[c] + eps);
            const float sc = gamma[c] * invstd, sh = beta[c] - rmean[c] * gamma[c] * invstd;
            for (int g = 0; g < groups; ++g) {
                tab[(g * 2 + 0) * CS + cl] = sc;
                tab[(g * 2 + 1) * CS + cl] = sh;
            }
        } else {
            for (int g = 0; g < groups; ++g) {
                tab[(g * 2 + 0) * CS + cl] = stat[(int64_t)g * 4 * C + 2 * C + c];
                tab[(g * 2 + 1) * CS + cl] = stat[(int64_t)g * 4 * C + 3 * C + c];
            }
        }
    }
}

// ---- "virtual activation" (round 4): the consumer of a conv -> BN -> ReLU -> Dropout2d layer reads the producer's RAW conv output
//      Y and applies scale / shift / ReLU / mask while it stages its input tile, so the activation A is never written to HBM (one
//      full-tensor pass and one launch less per layer; reference semantics: /root/reference/models/SiamUnet_diff.py:99-119).
//      The arithmetic is k_bn_act's, bit for bit: A = round_bf16(max(y * sc + sh, 0) * mk).
struct XfSrc {
    const long long* facc = nullptr;   // training forward: the producer's statistics accumulators (bn_fwd_table); nullptr otherwise
    const float* gamma = nullptr;      // forward (training / eval): affine parameters; nullptr: backward -- read the published `stat`
    const float* beta = nullptr;
    float* rmean = nullptr; float* rvar = nullptr;   // running statistics (updated by block 0 of the FORWARD consumer in training mode)
    float* stat = nullptr;             // producer's [groups][4][C] (mean, invstd, scale, shift): published by block 0 when publish != 0
    const float* mask = nullptr;       // Dropout2d factors [N][C] (0 or 1 / (1 - p)); nullptr: none
    int C = 0, groups = 1, npg = 0, publish = 0;
    long long ppg = 0;                 // values per channel and group
    float momentum = 0.1f, eps = 1e-5f;
    int on = 0;                        // 0: plain input tensor
};
// one 16-B piece (8 consecutive channels of one pixel).  sc / sh: the channel's BatchNorm scale / shift ALREADY multiplied by the
// image's Dropout2d factor (mk >= 0, so max(y * sc + sh, 0) * mk == max(y * (sc * mk) + sh * mk, 0): k_bn_act folds the same way);
// okm: 0xffffffff, or 0 for a zero-filled border piece, which must stay zero (the convolution pads A, not Y).  Per pair of
// channels: 2 unpacks, 2 FMAs, 1 v_cvt_pk_bf16_f32, 1 v_pk_max_i16 (ReLU on the packed bf16 bits: a negative bf16 is a negative
// int16; rounding and ReLU commute), 1 AND -- 3.5 VALU per element (the first version's fp32 max + mask multiply + select: 5).
__device__ __forceinline__ uint4 xf_act8(const uint4 raw, const float (&sc)[8], const float (&sh)[8], const unsigned okm) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float lo = fmaf(__uint_as_float(w[i] << 16), sc[2 * i], sh[2 * i]);
        const float hi = fmaf(__uint_as_float(w[i] & 0xffff0000u), sc[2 * i + 1], sh[2 * i + 1]);
        const s16x2 p = __builtin_bit_cast(s16x2, pack_bf16x2(lo, hi));
        o[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(p, s16x2{0, 0})) & okm;
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
// scale / shift of 8 channels folded with the Dropout2d factors of the image (identity factors when the layer has no mask)
__device__ __forceinline__ void xf_fold8(const float* __restrict__ tab_sc, const float* __restrict__ tab_sh, const float4 m0, const float4 m1,
                                         float (&sc)[8], float (&sh)[8]) {
    const float mk[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = tab_sc[j] * mk[j]; sh[j] = tab_sh[j] * mk[j]; }
}

// ------------------------------------------------------------------ host launchers (kernels_*.hip)
// All tensors NHWC with an explicit pixel stride ("ld", in elements); channel counts are multiples of 8.

// x1,x2 fp32 NCHW [B,cin,H,W] -> X [2B,H,W,8] (channels >= cin zero)
void launch_in_pack(int dt, const float* x1, const float* x2, void* X, int B, int cin, int H, int W, hipStream_t s, int dates = 2);
// g fp32 NCHW [B,L,H,W] -> G [B,H,W,8]; optional bias_acc (int64 [BN_REP][1][2][8], scale BN_BS): per-channel sums of g
void launch_gout_pack(int dt, const float* g, void* G, int B, int L, int H, int W, hipStream_t s, long long* bias_acc = nullptr);
// bias gradients that conv / pack kernels accumulated as integer sums -> fp32 gradient entries (one launch per stage)
struct BiasJob { int64_t acc_off; int64_t out_off; int C, valid; float scale; int pad_; };
void launch_bias_finish(const BiasJob* jobs_dev, int njobs, const char* ws, float* grads, hipStream_t s);

// generic tap-list convolution, reference FMA implementation.  w fp32 [ntaps][kpad][wld]; out_nchw_f32: write fp32
// NCHW [n,co,ho,wo] instead of NHWC activations (network output).
void launch_conv_ref(int dt, const stcd_conv_geom& g, const void* in, const float* w, int kpad, int wld,
                     const float* bias, void* out, bool out_nchw_f32, hipStream_t s);
// dw fp32 [ntaps][kpad][wld] += sum_pixels in(tap) * dout ; caller zeroes dw.
void launch_wgrad_ref(int dt, const stcd_conv_geom& g, const void* in, const void* dout, float* dw, int kpad, int wld,
                      hipStream_t s);
void launch_wgrad_ref_f64(int dt, const stcd_conv_geom& g, const void* in, const void* dout, double* dw, int kpad, int wld,
                          hipStream_t s);

// ---- MFMA (bf16) implementations, kernels_conv_mfma.hip
struct ConvMfmaPlan {
    int CiB = 0, nchunks = 0, KS = 0, modeB = 0, NT = 0, NTtot = 0;
    int64_t wf_elems = 0;   // bf16 elements of the fragment-order weight image
    bool ok = false;
};
ConvMfmaPlan conv_mfma_plan(const stcd_conv_geom& g);
// w fp32 [ntaps][kpad][wld] (the engine's packed layout) -> fragment-order bf16 image
void launch_pack_frag(const stcd_conv_geom& g, const ConvMfmaPlan& p, const float* w, int kpad, int wld, void* dst,
                      hipStream_t s);
int launch_conv_mfma(const stcd_conv_geom& g, const ConvMfmaPlan& p, const void* in, const void* wf, const float* bias,
                     void* out, bool out_nchw_f32, hipStream_t s);
int launch_conv_mfma_x4(const stcd_conv_geom g[4], const ConvMfmaPlan p[4], const void* in, const void* const wf[4],
                        const float* bias, void* out, hipStream_t s);
// resident-filter persistent kernel for 3x3 stride-1 convs with Ci % 32 == 0 (uses the mode-A fragment image)
struct ConvResPlan {
    int NT = 0, CW = 32, nslices = 0, P = 0, blocks = 0;
    int filt_bytes = 0, lds_bytes = 0;
    int single_halo = 0;       // one halo buffer (two barriers per step): a wider output-channel slice where LDS allows one block per CU anyway
    bool ok = false;
};
ConvResPlan conv_res_plan(const stcd_conv_geom& g, const ConvMfmaPlan& p, int groups);
// A data-gradient launch that also forms the BatchNorm-backward partial sums of the layer whose output gradient dA it writes
// (sum(dz), sum(dz * xhat), dz = dA * mask * (z > 0): what k_bn_reduce<T, 1> would compute from a second pass over dA and Y).
struct BwdSum {
    const void* Y = nullptr; int ldy = 0;          // that layer's conv output (same map, same channels as dA)
    const float* stat = nullptr;                   // its published [groups][4][C] (mean, invstd, scale, shift)
    const float* mask = nullptr;                   // its Dropout2d factors [N][C] or nullptr
    long long* acc = nullptr;                      // its backward accumulators (scale BN_BS)
    int groups = 1;
};
int launch_conv_res(const stcd_conv_geom& g, const ConvMfmaPlan& p, const ConvResPlan& rp, const void* in, const void* wf,
                    const float* bias, void* out, int groups, long long* stat_acc, int cpad, hipStream_t s, int stat_c0 = 0,
                    float s1_scale = BN_FS1, float s2_scale = BN_FS2, const XfSrc* xf = nullptr, const BwdSum* bs = nullptr);
bool conv_res_bwdsum_ok(const stcd_conv_geom& g, const ConvResPlan& rp);
// Tap-list convolutions with Ci % 64 == 0, Co % 64 == 0 as a tiled GEMM [positions x (taps * Ci)] . [(taps * Ci) x Co]: 128x128
// or 64x64 block tiles staged through LDS in (64-channel chunk, tap) steps, fused bias + BN statistics, LDS-transposed 16-B
// output stores.  1x1 convolutions of any stride, and the wide layers the resident-filter kernel cannot take.
// Uses the mode-A fragment image of conv_mfma_plan (CiB == 64).
struct ConvGemmPlan {
    int W = 0;             // fragments per wave and side: 4 -> 128x128 block tile, 2 -> 64x64
    int tiles_m = 0, tiles_n = 0, blocks = 0, lds_bytes = 0;
    int pix_chunks = 0;    // 1: the 64 "channels" of a row are 8 consecutive PIXELS of 8 channels (the 7x7 stem): bounds per pixel
    bool ok = false;
};
ConvGemmPlan conv_gemm_plan(const stcd_conv_geom& g, const ConvMfmaPlan& p, int groups);
// fused epilogue of k_conv_gemm, applied to v = conv + bias in this order:  v = relu(v) ; v = round(v) ; v = gate > 0 ? v : 0 ;
// out = alpha * v + beta * res   (gate / res: tensors of the output's geometry with their own pixel strides).  ResidualBlock of
// ChangeFormer's decoder head (ChangeFormerBaseNetworks.py:109-120) forward and backward without separate element-wise passes.
struct ConvEpi {
    int relu = 0;
    const void* gate = nullptr; int ldg = 0;
    const void* res = nullptr; int ldr = 0;
    float alpha = 1.f, beta = 1.f;
};
int launch_conv_gemm(const stcd_conv_geom& g, const ConvMfmaPlan& p, const ConvGemmPlan& gp, const void* in, const void* wf,
                   const float* bias, void* out, int groups, long long* stat_acc, int cpad, hipStream_t s, int stat_c0 = 0,
                   float s1_scale = BN_FS1, float s2_scale = BN_FS2, const ConvEpi* epi = nullptr);
// 3x3 stride-1 convolutions of wide layers on large maps (Ci % 64 == 0, Co % 128 == 0): resident 18 x 18 x 64-channel halo, the
// filter streamed per (chunk, tap) stage; 16 x 16-pixel x 128/256-channel block tiles, persistent blocks (kernels_conv_halo.hip).
// Reads the CiB = 64 fragment image (the one k_conv_gemm reads); ConvEpi as k_conv_gemm; no fused BN statistics.
struct ConvHaloPlan {
    int NH = 0;            // 128-channel halves of the block's output slice (1: 4 waves, 2: 8 waves)
    int nslices = 0, P = 0, blocks = 0, lds_bytes = 0;
    bool ok = false;
};
ConvHaloPlan conv_halo_plan(const stcd_conv_geom& g, const ConvMfmaPlan& p);
int launch_conv_halo(const stcd_conv_geom& g, const ConvMfmaPlan& p, const ConvHaloPlan& hp, const void* in, const void* wf,
                     const float* bias, void* out, hipStream_t s, const ConvEpi* epi = nullptr);

// Wide layers on large maps (Ci % 64 == 0, Co % 256 == 0, an even number >= 4 of (chunk, tap) K-tiles) as an implicit GEMM on a
// 256 x 256 block tile staged by LDS-DMA with counted vmcnt (8 waves, 4 phases per K-tile; kernels_conv_dma.hip).  Reads the
// CiB = 64 fragment image; ConvEpi as k_conv_gemm; one BatchNorm group, no fused statistics.
struct ConvDmaPlan {
    int tiles_m = 0, tiles_n = 0, blocks = 0, lds_bytes = 0;
    bool ok = false;
};
ConvDmaPlan conv_dma_plan(const stcd_conv_geom& g, const ConvMfmaPlan& p);
int launch_conv_dma(const stcd_conv_geom& g, const ConvMfmaPlan& p, const ConvDmaPlan& dp, const void* in, const void* wf,
                    const float* bias, void* out, hipStream_t s, const ConvEpi* epi = nullptr);

// small-channel persistent kernel (filter in registers, double-buffered halo, optional fused BN statistics);
// uses the mode-B fragment image of conv_mfma_plan.
bool conv_small_ok(const stcd_conv_geom& g, const ConvMfmaPlan& p);
int conv_small_blocks(const stcd_conv_geom& g, int groups);
int launch_conv_small(const stcd_conv_geom& g, const void* in, const void* wf_modeB, const float* bias, void* out,
                      bool out_nchw_f32, int groups, long long* stat_acc, int cpad, hipStream_t s, const XfSrc* xf = nullptr,
                      const BwdSum* bs = nullptr);
bool conv_small_bwdsum_ok(const stcd_conv_geom& g);      // shapes the fused form handles (16 channels out, whole 8 x 16 tiles, 3 / 5 k-steps)
struct WgradMfmaPlan {
    int WCI = 1, NTW = 1, gx = 1, gy = 1, gz = 1;
    int64_t slab_floats = 0;
    int wi_valid = 0;      // > 0: columns of the input that exist (a strided VIEW passes a virtual width: see configure_segcd)
    int gemm = 0;          // > 0: one-tap launch on k_wgrad_gemm<gemm> (32*gemm x 32*gemm channel tile), gx = position slices
    int dma = 0;           // 1: k_wgrad_dma (256 x 256 channel tile per (split, tap) block, LDS-DMA staging), gx = position splits, gy = taps
    bool ok = false;
};
WgradMfmaPlan wgrad_mfma_plan(const stcd_conv_geom& g, int kpad, int wld, bool allow_wide = false, int force_co64 = -1);   // allow_wide: 64 x 32 tile for ci >= 64; force_co64 1: the 64 x 64 tile (opt-in variant, see the plan)
int launch_wgrad_mfma(const stcd_conv_geom& g, const WgradMfmaPlan& p, const void* in, const void* dout, float* slab,
                      int kpad, int wld, hipStream_t s);
// One weight-gradient launch as the kernel sees it.  A backward stage runs ALL its launches of one kernel variant as a
// single grouped grid (device job table, blockIdx -> job by binary search): parallelism then comes from the layers
// side by side, so each block can own many tiles (few K-split slabs) and no layer pays its own launch / tail.
struct WgradJob {
    stcd_conv_geom g;
    int64_t in_off, dout_off, slab_off;    // bytes from the base pointer handed to the kernel (absolute when base == 0)
    int kpad, wld;
    int dymin, dxmin, HH, HWp;
    int tiles_x, tiles_y, ntiles;
    int xrow_bytes;                        // LDS bytes of one halo row of the X tile (incl. bank padding)
    int x_bytes, y_bytes;                  // LDS bytes of one X halo tile / one dY tile
    int co_valid;                          // channels of dout that exist in memory (round8(co))
    unsigned in_bytes, dout_bytes;         // tensor sizes (buffer-load range checks)
    int gx, gy, gz;                        // block grid of this job
    int start;                             // first block of the job inside a grouped launch
    int lds_bytes;
    int wi_valid;                          // input columns >= wi_valid read as zero (== g.wi for a plain tensor)
    int dma_L, dma_nk;                     // k_wgrad_dma: positions per split (a multiple of 64), K-tiles per block (even)
    int pad_;
    // X is a "virtual activation" (XfSrc): `in_off` addresses the producer's raw conv output, BN-affine + ReLU + Dropout2d are
    // applied while the X tile is staged (k_wgrad_group only).  xf_C == 0: plain input.
    int64_t xf_stat_off, xf_mask_off;      // producer's published [groups][4][C] table / Dropout2d factors [N][C] (-1: none); bytes from base
    int xf_C, xf_groups, xf_npg, pad2_;
};
// one-tap weight gradients with Ci, Co >= 64 (1x1 convs, the phases of 2x2 stride-2 transposed convs): dW = X^T . dY as a GEMM
// over positions, 128x128 / 64x64 channel tiles, positions split over gx blocks that each write one fp32 slab
WgradMfmaPlan wgrad_gemm_plan(const stcd_conv_geom& g, int kpad, int wld);
WgradJob wgrad_gemm_make_job(const stcd_conv_geom& g, const WgradMfmaPlan& p, int64_t in_off, int64_t dout_off, int64_t slab_off,
                             int kpad, int wld);
int launch_wgrad_gemm_group(int W, const WgradJob* jobs_dev, int njobs, int total_blocks, const char* base, hipStream_t s);
// weight gradient of wide stride-1 tap-list layers (Ci % 256 == 0, Co % 256 == 0, maps >= 64 wide) on the LDS-DMA pipeline
// (kernels_wgrad_dma.hip): block = (position split, tap, 256 x 256 channel tile); S splits -> S slabs [tap][ci][co]
WgradMfmaPlan wgrad_dma_plan(const stcd_conv_geom& g, int kpad, int wld);
void wgrad_dma_set_split(WgradMfmaPlan& p, const stcd_conv_geom& g, int S, int kpad, int wld);
WgradJob wgrad_dma_make_job(const stcd_conv_geom& g, const WgradMfmaPlan& p, int64_t in_off, int64_t dout_off, int64_t slab_off,
                            int kpad, int wld);
int launch_wgrad_dma_group(const WgradJob* jobs_dev, int njobs, int total_blocks, const char* base, hipStream_t s);
WgradJob wgrad_make_job(const stcd_conv_geom& g, const WgradMfmaPlan& p, int64_t in_off, int64_t dout_off, int64_t slab_off,
                        int kpad, int wld);
void wgrad_job_set_xf(WgradJob& a, int64_t stat_off, int64_t mask_off, int C, int groups, int npg);
int wgrad_variant_slots(int WCI, int NTW, bool t9, int lds_bytes);   // resident blocks of that kernel variant on the chip
int launch_wgrad_group(int WCI, int NTW, bool t9, const WgradJob* jobs_dev, int njobs, int total_blocks, int lds_bytes,
                       const char* base, hipStream_t s, bool xf = false);
struct PackSpec;
// Batched slab reduction: ONE launch per backward stage sums every weight-gradient launch's K-split slabs straight
// into the reference-layout gradient tensors (device job table built at configure time).
struct ReduceJob {
    int64_t start, count;      // range in the launch's global index space (count = outputs * 16 lanes)
    int64_t slab_off;          // byte offset of this launch's slabs in the workspace
    int64_t out_off;           // float offset of the filter in the flat gradient buffer
    int64_t slab_stride;       // floats between consecutive slabs
    int gx, ntaps, K, N, kpad, wld;
    int ks, kn_major;          // reference index map of the launch's taps (as PackSpec)
    int8_t ky[9], kx[9];
    int8_t tiled, parts, pad_[4]; // tiled 1: block = (32 co x TK ci x taps tile), all slabs, LDS-transposed stores; 0: thread = (output, part),
                               // `parts` (power of two <= 64) threads of one block per output, summed in index order (deterministic)
};
void launch_reduce_jobs(const ReduceJob* jobs_dev, int njobs, int64_t total, const char* ws, float* grads, hipStream_t s);
// out = sum of the gx slabs; ps == nullptr: engine layout [tap][kpad][wld], else scattered into the reference layout
void launch_reduce_dw(const float* slab, int gx, const stcd_conv_geom& g, int K, int N, int kpad, int wld,
                      const PackSpec* ps, float* out, hipStream_t s);

// weight (un)packing between the reference layouts and the engine layout [tap][k][wld]
struct PackSpec {
    int ks;          // source kernel size (1,2,3)
    int kn_major;    // 1: src[k][n][ks][ks] ; 0: src[n][k][ks][ks]
    int K, N, wld;   // reduction channels, output channels, padded row length of the packed matrix
    int kpad;        // packed rows per tap (>= K; rows >= K are zero)
    int ntaps;
    int8_t ky[9], kx[9];
};
void launch_pack_w(const PackSpec& ps, const float* src, float* dst, hipStream_t s);          // dst[t][k][n]
void launch_unpack_dw(const PackSpec& ps, const double* dwe, float* gsrc, hipStream_t s);      // gsrc[src idx] = dwe[t][k][n]

// One launch repacks every filter of the network straight from the reference-layout parameters:
// job kind 0 -> fp32 [tap][kpad][wld] (reference kernels), kind 1 -> bf16 MFMA-fragment image.
struct PackJob {
    int64_t start, count;     // range of this job in the launch's global element index space
    int64_t src_off;          // float offset of the filter in the flat parameter buffer
    int64_t dst_off;          // byte offset of the image in the workspace
    int kind;
    PackSpec ps;              // reference index map restricted to the launch's taps
    int Ci, Co, CiB, nchunks, KS, NTtot, modeB;   // fragment-image parameters (kind 1)
    int aux;                  // kind 2 (7x7 stem as a 7-tap GEMM over 8-pixel rows): real input channels
};
void launch_pack_jobs(const PackJob* jobs_dev, int njobs, int64_t total, const float* params, char* ws, hipStream_t s);

// batch-norm (train): per-(group, channel) sums of y and y^2 over a group's pixels, added into acc (see BN accumulators)
int bn_stats_chunks(int64_t pixels_per_group, int C);
void launch_bn_stats(int dt, const void* Y, int ld, int C, int groups, int64_t pixels_per_group, long long* acc, hipStream_t s);
// eval mode: stat fp32 [groups][4][C] = mean, invstd, scale, shift from the running statistics
void launch_bn_finalize(const XfSrc& x, hipStream_t s);
void launch_bn_eval_prepare(int C, int groups, const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float* stat, float eps, hipStream_t s);
// A = relu(Y*scale+shift) * mask ; optional fused 2x2 max-pool output P (floor).
// Images [g*npg, (g+1)*npg) belong to group g; A for group g starts at A_base + g*a_group_off (elements).
// Training (facc != nullptr): every block derives scale / shift of all channels from the accumulators in its prologue;
// block 0 also writes stat [groups][4][C] (kept for the backward) and updates the running statistics (group 0 then
// group 1: the shared encoder BatchNorm sees date 0 then date 1).  Eval (facc == nullptr): stat is read.
// extra destinations of an activation (dense concatenation without copy kernels: the producer also writes its output into
// the channel slice of every consumer's concat buffer) and extra sources of a gradient (the consumers' concat-gradient
// slices are summed by the producer's BatchNorm-backward reduction instead of by scatter-add kernels).
// Element (g, image-in-group n, pixel, c) of view k lives at p + (both ? g*goff : 0) + (n*HW + pixel)*ld + c; a view with
// gmask == 2 exists for group 1 only (a tensor of the date-1 images alone), gmask == 3 for both groups (7: three groups);
// a view of several groups strides by goff per group, a single-group view points at its group's data.
constexpr int MAX_VIEWS = 6;
struct SliceViews {
    int n = 0;
    void* p[MAX_VIEWS]; int ld[MAX_VIEWS]; int64_t goff[MAX_VIEWS]; int gmask[MAX_VIEWS];
};
struct BnActArgs {
    const void* Y; int ldy;
    void* A; int lda; int64_t a_group_off;
    void* P; int ldp;                       // nullable
    float* stat;                            // [groups][4][C]
    const float* mask;                      // nullable, [groups*npg][C]
    int C, groups, npg, H, W;
    int relu;                               // 0: affine only
    const void* res = nullptr; int ldres = 0; // optional residual added before the ReLU (plain [N,HW,ldres] tensor)
    const long long* facc = nullptr;        // training: forward accumulators of this layer
    const float* gamma = nullptr; const float* beta = nullptr;
    float* running_mean = nullptr; float* running_var = nullptr;
    float momentum = 0.1f, eps = 1e-5f;
    SliceViews extra;                       // extra destinations of A
    int g_first = 0;                        // group whose statistics update the running ones first (then cyclic)
};
void launch_bn_act(int dt, const BnActArgs& a, hipStream_t s);
// skip layers (groups == 2, ReLU, no residual): also writes the bi-temporal fusion F = |a1-a2| (fmode 0) / a2-a1 (1)
void launch_bn_act_pair(int dt, const BnActArgs& a, void* F, int ldf, int fmode, hipStream_t s);
void launch_maxpool(int dt, const void* A, int lda, void* P, int ldp, int N, int H, int W, int C, hipStream_t s);

// Grouped views: element (g, n_in_group, pix, c) of an activation lives at p + g*goff + (n_in_group*HW + pix)*ld + c.
// A plain [groups*npg, HW, C] tensor has goff = npg*HW*ld; the conc variant keeps T1/T2 skips as channel slices of
// the decoder's concat buffer (goff = C).
// skip fusion: D[n][.., 0:C] = |a1-a2| (mode 0) or a2-a1 (mode 1); a1 = group 0 image n, a2 = group 1 image n
void launch_fuse(int dt, int mode, const void* A, int lda, int64_t a_goff, void* D, int ldd, int B, int64_t HW, int C,
                 hipStream_t s);
void launch_fuse_bwd(int dt, int mode, const void* A, int lda, int64_t a_goff, const void* dD, int ldd, void* dA, int ldda,
                     int64_t da_goff, int B, int64_t HW, int C, hipStream_t s);
// replication pad of the last row/col of a channel slice (odd sizes): rows [h0,H) copy row h0-1, cols likewise
void launch_rep_pad(int dt, void* D, int ld, int N, int H, int W, int h0, int w0, int C, hipStream_t s);
void launch_rep_pad_bwd(int dt, void* dD, int ld, int N, int H, int W, int h0, int w0, int C, hipStream_t s);

// backward of bn_act (train): dz = dA*mask*(z>0); sums of dz and dz*xhat per (group, channel), added into bacc
void launch_bn_bwd_reduce(int dt, const void* dA, int ldda, int64_t da_goff, const void* Y, int ldy, const float* stat,
                          const float* mask, int C, int groups, int npg, int64_t HW, int relu, long long* bacc,
                          hipStream_t s, const void* res = nullptr, int ldres = 0, const SliceViews* extra_src = nullptr,
                          int base_valid = 1, void* dA_sum = nullptr);
// (extra_src: dA := [base_valid ? dA : 0] + sum of the views, rounded once and written to dA_sum -- a plain view like dA)
// dY = scale*(dz - k1 - xhat*k2) = scale*dz + b*(y-mean) + c with (b, c) derived per block from bacc and stat in the
// kernel's prologue (no finalize launch); block 0 also writes dgamma / dbeta (summed over the groups).  dY is a plain
// tensor and may alias dA when dA is plain too.
// optional: res (residual inside the ReLU gate), dZout (the gated dz is also written: the residual branch's gradient),
// extra (added to dY: gradient arriving through a residual branch)
void launch_bn_bwd_apply(int dt, const void* dA, int ldda, int64_t da_goff, void* dY, int lddy, const void* Y, int ldy,
                         const float* stat, const long long* bacc, float* dgamma, float* dbeta, const float* mask, int C, int groups,
                         int npg, int64_t HW, int relu, hipStream_t s, const void* res = nullptr, int ldres = 0,
                         void* dZout = nullptr, int lddz = 0, const void* extra = nullptr, int ldex = 0,
                         float* dbeta_copy = nullptr);      // dbeta_copy: a second destination of d(beta) (SNUNet: conv1's bias gradient)
// pairwise depthwise 3x3 convolution of SiamUnet_cross_conc's skip blocks (kernels_xconc.hip; SiamUnet_crossconc.py:14-18,24-33):
// G[n, p, c] = b[c] + sum_t w[c][0][t] * A1[n, p + t, c] + w[c][1][t] * A2[n, p + t, c]; A1 / A2 (and dA1 / dA2) `goff` elements apart
void launch_pairdw_fwd(int dt, const void* A, int lda, int64_t goff, void* G, int ldg, const float* w, const float* b, int B, int H,
                       int W, int C, hipStream_t s);
void launch_pairdw_bwd_data(int dt, const void* dG, int lddg, void* dA, int ldda, int64_t goff, const float* w, int B, int H, int W,
                            int C, hipStream_t s);
int64_t pairdw_partial_floats(int B, int H, int W, int C);
void launch_pairdw_bwd_filter(int dt, const void* A, int lda, int64_t goff, const void* dG, int lddg, float* dw, float* partial, int B,
                              int H, int W, int C, hipStream_t s);
// dst[.., 0:C] (ld ldd) = or += src[.., 0:C] (ld lds): dense concatenation by copy, and its gradient scatter
void launch_slice(int dt, void* dst, int ldd, const void* src, int lds, int64_t pixels, int C, int accumulate, hipStream_t s);
// ECAM head of SNUNet (SNUNet.py:46-59,144-149); scratch layouts documented at the kernels (kernels_ew.hip)
void launch_ecam_forward(int dt, const void* X, int ld, void* Z, int ldz, int N, int64_t HW, int C4, const float* w1a,
                         const float* w2a, const float* w1b, const float* w2b, float* pool, int64_t* argm, float* att,
                         float* hid, float* part, hipStream_t s);
int64_t ecam_part_floats(int N, int C4);       // scratch of the chunked ECAM reductions (`part`)
void launch_ecam_backward(int dt, const void* X, int ld, const void* dZ, int lddz, void* dX, int lddx, int N, int64_t HW, int C4,
                          const float* w1a, const float* w2a, const float* w1b, const float* w2b, float* gw1a, float* gw2a,
                          float* gw1b, float* gw2b, const float* pool, const int64_t* argm, const float* att, const float* hid,
                          float* sums, float* dpool, float* part, hipStream_t s);
// dA (+)= route(dP) to the first maximum of each 2x2 window of A
void launch_pool_bwd(int dt, const void* A, int lda, int64_t a_goff, const void* dP, int ldp, void* dA, int ldda,
                     int64_t da_goff, int groups, int npg, int H, int W, int C, int accumulate, hipStream_t s);
// encoder skip layer: dA = pool gradient + skip-fusion gradient and the BN-backward partial sums of it, in one pass
// (mode 0: |a1-a2| skips, 1: a2-a1); the sums go into bacc as launch_bn_bwd_reduce's
bool skip_pair_supported(int B, int H, int W, int C);   // pairs, map size, channels of the skip layer
void launch_skip_bwd_pair(int dt, int mode, const void* Y, int ldy, const void* dD, int ldd, const void* dP, int ldp, void* dA,
                          int ldda, int64_t da_goff, const float* stat, const float* mask, int B, int H, int W, int C,
                          long long* partial, hipStream_t s);
void launch_skip_bwd(int dt, int mode, const void* A, int lda, int64_t a_goff, const void* Y, int ldy, const void* dD, int ldd,
                     const void* dP, int ldp, void* dA, int ldda, int64_t da_goff, const float* stat, const float* mask, int B,
                     int H, int W, int C, long long* bacc, hipStream_t s);
// db[c] = sum over pixels of dY[.., c]  (db zeroed by caller; atomics)
// acc_i (bf16 mode): int64 accumulator [BN_REP][1][2][C] at scale BN_BS (zeroed by the caller, converted by k_bias_finish); nullptr: float atomics into db
void launch_bias_grad(int dt, const void* dY, int ld, int64_t pixels, int C, float* db, hipStream_t s, long long* acc_i = nullptr);
// masks from a counter hash: mask[i] = (u(seed, i) >= p) / (1-p)
void launch_dropout_gen(float* mask, int64_t n, uint64_t seed, float p, hipStream_t s);
void launch_fill(float* p, int64_t n, float v, hipStream_t s);

// ---- ResNet-50 UNet (SegCD) specific kernels (kernels_ew.hip)
void launch_stem_fwd(int dt, const void* X, const float* w, void* Y, int N, int H, int W, int cin, int Co, hipStream_t s);
// part: scratch of stem_wgrad_part_floats() floats (per-chunk partial filters, summed in chunk order by a finish launch)
int64_t stem_wgrad_part_floats(int N, int H, int W, int cin, int Co);
void launch_stem_wgrad(int dt, const void* X, const void* dY, float* dW, int N, int H, int W, int cin, int Co, hipStream_t s, float* part);
// idx: [N, H/2, W/2, C] bytes, the winning window position 0..8 of every pooled element (nullable in the forward: inference)
void launch_maxpool3(int dt, const void* A, int lda, void* P, int ldp, int N, int H, int W, int C, hipStream_t s, unsigned char* idx = nullptr);
void launch_maxpool3_bwd(int dt, const unsigned char* idx, const void* dP, int ldp, void* dA, int ldda, int N, int H, int W, int C,
                         hipStream_t s);
void launch_upsample2(int dt, const void* X, int ldx, void* D, int ldd, int N, int h, int w, int C, hipStream_t s);
void launch_upsample2_bwd(int dt, const void* dD, int ldd, void* dX, int ldx, int N, int h, int w, int C, hipStream_t s);
void launch_segcd_combine(const float* raw, float* out, int64_t n, hipStream_t s);
void launch_segcd_combine_bwd(const float* raw, const float* g, float* draw, int64_t n, hipStream_t s);

// losses / metric (kernels_loss.hip)
int64_t loss_scratch_bytes();
void launch_loss_ce(const float* logits, const int64_t* target, int B, int Cn, int64_t HW, int ignore, float* loss,
                    float* dlogits, void* scratch, hipStream_t s);
void launch_loss_bce_dice(const float* logits, const float* target, int64_t n, int from_logits, float* loss,
                          float* dlogits, void* scratch, hipStream_t s);
void launch_loss_contrastive(const float* pred, const int64_t* cd_label, const int64_t* pse_label, int64_t n_half, float* loss,
                             float* dpred, void* scratch, hipStream_t s);
void launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float omb1, float beta2, float omb2, float eps,
                 float wd, int decoupled, float step_size, float inv_bc2_sqrt, hipStream_t s);
void launch_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, int decoupled, hipStream_t s);
void launch_pseudo_pair(const uint8_t* A, const uint8_t* donor, const uint8_t* mask, const uint8_t* change, const float* alpha,
                        const int32_t* erase, uint64_t seed, int B, int H, int W, const float* mean, const float* std_,
                        float* x1, float* x2, int64_t* c_label, int64_t* s_label_a, int64_t* s_label_b, hipStream_t s);
void launch_augment(const float* x, const float* params, int N, int H, int W, const float* mean, const float* std_, float* out,
                    void* scratch, hipStream_t s);
int64_t augment_scratch_bytes(int N, int H, int W);
void launch_confusion(const float* logits, const int64_t* target, int B, int Cn, int64_t HW, int64_t* cm,
                      hipStream_t s);

// ---- ChangeFormer (transformer) kernels, kernels_tf.hip.  Tokens are NHWC pixels: a [n, N = h*w, C] sequence IS the
//      [n, h, w, C] map.  Every Dropout / DropPath site draws its mask from a counter hash of (site seed, element index in the
//      layout named at the launcher), so no mask is ever stored: the backward recomputes it.
struct DropSite {
    uint32_t seed = 0;     // stcd_cf_site_seed(step seed, site index)
    uint32_t thr = 0;      // floor(p * 2^24); 0: identity (evaluation mode or p == 0)
    float scale = 1.f;     // 1 / (1 - p)
};
__host__ __device__ inline uint32_t cf_hash(uint32_t idx, uint32_t seed) {
    uint32_t h = idx * 0x9E3779B1u + seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ inline float cf_keep(uint32_t idx, const DropSite& d) { return (cf_hash(idx, d.seed) >> 8) >= d.thr ? d.scale : 0.f; }
uint32_t cf_site_seed(uint64_t seed, int site);
DropSite cf_make_site(uint64_t seed, int site, float p, bool training);

// im2col with the reference's K order (ci, ky, kx): col[(n, oy, ox)][ci*k*k + ky*k + kx] = X[n, oy*stride - pad + ky, ox*stride - pad + kx, ci]
// (zero outside the image; columns [C*k*k, ldc) are zero-filled).  Patch embeddings (k7 s4 p3, k7 s2 p3) and the spatial-reduction
// convs (k = stride = sr, pad 0: a pure permutation) become plain GEMMs over it.
void launch_im2col(int dt, const void* X, int ldx, void* col, int ldc, int n, int H, int W, int C, int k, int stride, int pad, int Ho,
                   int Wo, hipStream_t s);
// dX[n, y, x, ci] (=, or += when accumulate) sum over the patches that contain the pixel of dcol
void launch_col2im(int dt, const void* dcol, int ldc, void* dX, int lddx, int n, int H, int W, int C, int k, int stride, int pad, int Ho,
                   int Wo, int accumulate, hipStream_t s);
// LayerNorm over the C channels of each of M rows; stats fp32 [M][2] = mean, rstd (kept for the backward)
void launch_layernorm(int dt, const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta, float* stats, int64_t M,
                      int C, float eps, hipStream_t s);
int64_t layernorm_bwd_scratch_floats(int64_t M, int C);
// dx = [add] + LayerNorm-backward(dy [+ dy2]); dgamma / dbeta (fp32 [C]) overwritten (two-phase, fixed order: reproducible)
void launch_layernorm_bwd(int dt, const void* dy, int lddy, const void* dy2, int lddy2, const void* x, int ldx, const float* stats,
                          const float* gamma, const void* add, int ldadd, void* dx, int lddx, float* dgamma, float* dbeta,
                          float* scratch, int64_t M, int C, hipStream_t s);
// out[c] = sum over M rows of X[.., c] (fp32, two-phase fixed order); scratch >= colsum_scratch_floats(M, C)
int64_t colsum_scratch_floats(int64_t M, int C);
void launch_colsum(int dt, const void* X, int ld, int64_t M, int C, float* out, float* scratch, hipStream_t s);
// all column sums (bias gradients) of a backward stage as ONE grouped launch + one finish launch over a device job table
struct ColsumJob {
    int64_t x_off;            // byte offset of the [M, C] tensor (pixel stride ld) in the workspace
    int64_t M;
    int64_t out_off;          // float offset of the C outputs in the flat gradient buffer
    int64_t part_off;         // float offset of this job's partial sums [nblocks][C] in the partial buffer
    int ld, C;
    int start_block, nblocks; // blocks of the reduction launch
    int fin_start, pad_;      // first block of the finish launch (ceil(C / 8) blocks per job)
};
int colsum_job_blocks(int64_t M);
void launch_colsum_group(int dt, const ColsumJob* jobs_dev, int njobs, int total_blocks, int fin_blocks, const char* ws, float* partial,
                         float* grads, hipStream_t s);
// BatchNorm batch statistics in double precision (few values per channel: see kernels_tf.hip): fills stat [4][C] = mean, invstd,
// scale, shift (what launch_bn_act reads with facc == gamma == nullptr, and the BN-backward kernels) and updates the running statistics
int64_t bn_precise_scratch_floats(int64_t M, int C);
void launch_bn_stats_precise(int dt, const void* Z, int ld, int64_t M, int C, const float* gamma, const float* beta, float* rmean,
                             float* rvar, float* stat, float* scratch, float momentum, float eps, hipStream_t s);
// softmax attention with spatial-reduction keys (ChangeFormer.py:334-358): q [n, N, heads*d] (channel = head*d + j), kv [n, Nkv, 2*heads*d]
// (k at channel head*d + j, v at heads*d + head*d + j), out [n, N, heads*d]; lse fp32 [n, heads, N] = log-sum-exp of the scaled scores;
// attention dropout on the probabilities, element index ((img*heads + head)*N + i)*Nkv + j.  impl 0: plain-FMA kernels (fp32 math,
// any dtype), 1: MFMA (bf16, d in {64, 80}... see kernels_attn.hip)
void launch_attn_fwd(int dt, const void* q, int ldq, const void* kv, int ldkv, void* out, int ldo, float* lse, int n, int N, int Nkv,
                     int heads, int d, float scale, DropSite drop, hipStream_t s);
int64_t attn_bwd_scratch_floats(int n, int N, int Nkv, int heads, int d);
// matrix-core path (kernels_attn.hip): bf16, head dimension a multiple of 16 (<= 128), Nkv <= 256; STCD_NO_MFMA_ATTENTION=1 disables it
bool attn_mfma_ok(int dt, int Nkv, int d);
int64_t attn_mfma_bwd_scratch_floats(int n, int N, int Nkv, int heads, int d);
int launch_attn_fwd_mfma(const void* q, int ldq, const void* kv, int ldkv, void* out, int ldo, float* lse, int n, int N, int Nkv, int heads,
                         int d, float scale, DropSite drop, hipStream_t s);
int launch_attn_bwd_mfma(const void* q, int ldq, const void* kv, int ldkv, const void* out, int ldo, const void* dout, int lddo,
                         const float* lse, void* dq, int lddq, void* dkv, int lddkv, float* scratch, int n, int N, int Nkv, int heads, int d,
                         float scale, DropSite drop, hipStream_t s);
// dq [n, N, heads*d], dkv [n, Nkv, 2*heads*d]; scratch: fp32 (row terms D + the dK/dV partial slabs)
void launch_attn_bwd(int dt, const void* q, int ldq, const void* kv, int ldkv, const void* out, int ldo, const void* dout, int lddo,
                     const float* lse, void* dq, int lddq, void* dkv, int lddkv, float* scratch, int n, int N, int Nkv, int heads, int d,
                     float scale, DropSite drop, hipStream_t s);
// Mix-FFN middle (ChangeFormer.py:287-295, 517-523): u = depthwise3x3(h) + b ; a = dropout(GELU(u)).  w: reference layout [Ch][1][3][3]
void launch_dwgelu_fwd(int dt, const void* h, void* u, void* a, const float* w, const float* b, int n, int H, int W, int Ch,
                       DropSite drop, hipStream_t s);
int64_t dwgelu_bwd_scratch_floats(int n, int H, int W, int Ch);
// g = da * mask * GELU'(u) (written over da), dh = depthwise3x3^T(g), dw [Ch][9], db [Ch] overwritten
void launch_dwgelu_bwd(int dt, const void* h, const void* u, void* da, void* dh, const float* w, float* dw, float* db, float* scratch,
                       int n, int H, int W, int Ch, DropSite drop, hipStream_t s);
// out = x + droppath(img) * dropout(idx) * y over [n, rows_per_img, C] (element index = linear NHWC index); out may alias x
void launch_resid_drop(int dt, const void* x, const void* y, void* out, int n, int64_t rows_per_img, int C, DropSite drop, DropSite path,
                       hipStream_t s);
// dy = droppath(img) * dropout(idx) * dout
void launch_resid_drop_bwd(int dt, const void* dout, void* dy, int n, int64_t rows_per_img, int C, DropSite drop, DropSite path,
                           hipStream_t s);
// bilinear resize, align_corners = False (F.interpolate, ChangeFormer.py:1585,1591): src [n,h,w,C] -> dst [n,H,W,C] (= or +=)
void launch_bilinear(int dt, const void* src, int lds, void* dst, int ldd, int n, int h, int w, int H, int W, int C, int accumulate,
                     hipStream_t s);
void launch_bilinear_bwd(int dt, const void* ddst, int ldd, void* dsrc, int lds, int n, int h, int w, int H, int W, int C, int accumulate,
                         hipStream_t s);
// elementwise family over [rows, C] tensors with pixel strides
// z = PReLU(y; alpha[0])
void launch_prelu(int dt, const void* y, int ldy, void* z, int ldz, const float* alpha, int64_t rows, int C, hipStream_t s);
// dy = dz * (y > 0 ? 1 : alpha) ; dalpha[0] = sum dz * y * [y <= 0] (overwritten; scratch >= 2048 doubles)
void launch_prelu_bwd(int dt, const void* dz, int lddz, const void* y, int ldy, void* dy, int lddy, const float* alpha, float* dalpha,
                      float* scratch, int64_t rows, int C, hipStream_t s);
// out = x * dropout(linear index over [rows, C]) ; out may alias x (forward and backward are the same map)
void launch_dropout_ew(int dt, const void* x, int ldx, void* out, int ldo, int64_t rows, int C, DropSite drop, hipStream_t s);
// out = relu(x) ; dx = dy * [a > 0] (a = the ReLU's OUTPUT)
void launch_relu(int dt, const void* x, int ldx, void* out, int ldo, int64_t rows, int C, hipStream_t s);
void launch_relu_bwd(int dt, const void* dy, int lddy, const void* a, int lda, void* dx, int lddx, int64_t rows, int C, hipStream_t s);
// out = alpha * x + beta * y (y nullable; out may alias x or y)
void launch_axpby(int dt, float alpha, const void* x, int ldx, float beta, const void* y, int ldy, void* out, int ldo, int64_t rows, int C,
                  hipStream_t s);
// fp32 NCHW [n, C, H, W] maps of the auxiliary prediction heads (ChangeFormer.py:1151-1157: ReLU - BatchNorm2d(C) - Conv3x3(C -> C), C <= 8)
void launch_aux_head(const float* y, float* out, const float* bn_w, const float* bn_b, float* running_mean, float* running_var,
                     const float* w, const float* b, float* stat, int n, int C, int H, int W, int training, hipStream_t s);
// backward of the same head given g = d(loss)/d(out): conv3's filter / bias gradients, BatchNorm's, and dy = d(loss)/d(y) (fp32 NCHW);
// stat as the forward left it (scale, shift at [2c], mean, invstd at [16 + 2c]); dz / dy / sums: scratch of y's size / 16 floats
void launch_aux_head_bwd(const float* y, const float* stat, const float* g, const float* w3, float* dz, float* dy, float* sums, float* dw3,
                         float* db3, float* dgamma, float* dbeta, int n, int C, int H, int W, hipStream_t s);

}  // namespace stcd
