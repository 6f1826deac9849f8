// kernels_tf.hip -- the ChangeFormer kernels that are not convolutions / GEMMs: im2col / col2im with the
// reference's K order, LayerNorm, plain-FMA softmax attention (the fp32 parity path and the fallback of kernels_attn.hip), the
// Mix-FFN middle (depthwise 3x3 + GELU + dropout), residual + DropPath, bilinear resize, PReLU / ReLU / dropout / axpby, the
// auxiliary prediction heads.  Reference semantics: /root/reference/models/ChangeFormer.py (lines cited per kernel).
// All kernels are templated on the storage type (float: parity mode, bf16: production) and compute in fp32.
#include <algorithm>

#include "common.h"

namespace stcd {

static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

uint32_t cf_site_seed(uint64_t seed, int site) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(site + 1);      // splitmix64, as k_dropout_gen
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}
DropSite cf_make_site(uint64_t seed, int site, float p, bool training) {
    DropSite d;
    if (!training || p <= 0.f) return d;
    d.seed = cf_site_seed(seed, site);
    d.thr = (uint32_t)(p * 16777216.0f);
    d.scale = 1.0f / (1.0f - p);
    return d;
}

template <typename T>
__device__ __forceinline__ T from_float(float v) { return (T)v; }

// ------------------------------------------------------------------------------------------------ partial-sum finish
// out[j] = sum_b partial[b * stride + j], j < n0 -> out0[j], else out1[j - n0]; fixed summation tree: reproducible.
// 32 lanes share one output (strided partial sums, then a butterfly), 8 outputs per block.
__global__ void __launch_bounds__(256)
k_partial_finish(const float* __restrict__ partial, int nblk, int64_t stride, int n0, int n, float* __restrict__ out0,
                 float* __restrict__ out1) {
    const int j = blockIdx.x * 8 + (threadIdx.x >> 5), bl = threadIdx.x & 31;
    float s = 0.f;
    if (j < n)
        for (int b = bl; b < nblk; b += 32) s += partial[(int64_t)b * stride + j];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (j < n && bl == 0) { if (j < n0) out0[j] = s; else out1[j - n0] = s; }
}
static inline void launch_partial_finish(const float* partial, int nblk, int64_t stride, int n0, int n, float* out0, float* out1, hipStream_t s) {
    k_partial_finish<<<cdiv(n, 8), 256, 0, s>>>(partial, nblk, stride, n0, n, out0, out1);
}

// ------------------------------------------------------------------------------------------------ im2col / col2im
// OverlapPatchEmbed.proj (ChangeFormer.py:207-208: Conv2d(k, stride, padding k // 2)) and Attention.sr (:315: Conv2d(k = stride = sr))
// as GEMMs over patches.  K order (ci, ky, kx) = the reference weight's own [Co][Ci][k][k] flattening, so the filter and its
// gradient need no permutation.  One block per output position: the patch is read pixel by pixel (contiguous channels) into an
// LDS tile [k*k][C + 2] (odd word pitch: the transposed read is bank-conflict free) and leaves channel-major.
template <typename T>
__global__ void k_im2col(const T* __restrict__ X, int ldx, T* __restrict__ col, int ldc, int H, int W, int C, int k, int stride, int pad,
                         int Ho, int Wo) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);
    const int pix = blockIdx.x;
    const int ox = pix % Wo, t_ = pix / Wo, oy = t_ % Ho, n = t_ / Ho;
    const int kk = k * k, P = C + 2;
    for (int e = threadIdx.x; e < kk * C; e += blockDim.x) {
        const int t = e / C, c = e - t * C, ky = t / k, kx = t - ky * k;
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        T v = from_float<T>(0.f);
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = X[((int64_t)(n * H + iy) * W + ix) * ldx + c];
        tile[t * P + c] = v;
    }
    __syncthreads();
    T* dst = col + (int64_t)pix * ldc;
    for (int e = threadIdx.x; e < ldc; e += blockDim.x) {
        T v = from_float<T>(0.f);
        if (e < kk * C) { const int c = e / kk, t = e - c * kk; v = tile[t * P + c]; }
        dst[e] = v;
    }
}
void launch_im2col(int dt, const void* X, int ldx, void* col, int ldc, int n, int H, int W, int C, int k, int stride, int pad, int Ho,
                   int Wo, hipStream_t s) {
    const size_t lds = (size_t)k * k * (C + 2) * dsize(dt);
    const int threads = 256;
    const unsigned grid = (unsigned)((int64_t)n * Ho * Wo);
    if (dt == BF16) k_im2col<bf16><<<grid, threads, lds, s>>>((const bf16*)X, ldx, (bf16*)col, ldc, H, W, C, k, stride, pad, Ho, Wo);
    else k_im2col<float><<<grid, threads, lds, s>>>((const float*)X, ldx, (float*)col, ldc, H, W, C, k, stride, pad, Ho, Wo);
}

template <typename T>
__global__ void k_col2im(const T* __restrict__ dcol, int ldc, T* __restrict__ dX, int lddx, int H, int W, int C, int k, int stride, int pad,
                         int Ho, int Wo, int accumulate) {
    const int pix = blockIdx.x;
    const int ix = pix % W, t_ = pix / W, iy = t_ % H, n = t_ / H;
    const int kk = k * k;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int ty = iy + pad - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int tx = ix + pad - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                acc += (float)dcol[((int64_t)(n * Ho + oy) * Wo + ox) * ldc + c * kk + ky * k + kx];
            }
        }
        T* d = dX + (int64_t)pix * lddx + c;
        if (accumulate) acc += (float)*d;
        *d = from_float<T>(acc);
    }
}
// Overlapping patches (stride < k: the patch embeddings).  The gather above reads one 2-byte element per lane and patch; here a block
// owns an 8 x 8 input tile x 64 channels, walks the output positions whose patches touch the tile, stages each position's
// (contiguous) 64-channel slice of the dcol row in LDS and adds its taps into an fp32 tile accumulator in LDS -- every global
// access is a full line.  Two taps of one position never hit the same pixel, so the adds of one position need no atomics.
template <typename T>
__global__ void __launch_bounds__(256)
k_col2im_tile(const T* __restrict__ dcol, int ldc, T* __restrict__ dX, int lddx, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo,
              int tiles_x, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int TS = 8, CB = 64;
    float* acc = reinterpret_cast<float*>(smem_raw);                 // [64 pixels][64 channels]
    T* rowb = reinterpret_cast<T*>(acc + TS * TS * CB);              // [64 channels][k * k]
    const int kk = k * k, tid = threadIdx.x, cl = tid & 63, qd = tid >> 6;
    const int ty0 = (blockIdx.x / tiles_x) * TS, tx0 = (blockIdx.x % tiles_x) * TS, n = blockIdx.y, c0 = blockIdx.z * CB;
    const int cw = C - c0 < CB ? C - c0 : CB;                        // channels of this chunk
    for (int i = tid; i < TS * TS * CB; i += 256) acc[i] = 0.f;
    // output positions whose patch [o * stride - pad, o * stride - pad + k) meets the tile rows / columns
    int oy_lo = (ty0 + pad - k + 1 + stride - 1) / stride, oy_hi = (ty0 + TS - 1 + pad) / stride;
    int ox_lo = (tx0 + pad - k + 1 + stride - 1) / stride, ox_hi = (tx0 + TS - 1 + pad) / stride;
    if (ty0 + pad - k + 1 < 0) oy_lo = 0;
    if (tx0 + pad - k + 1 < 0) ox_lo = 0;
    oy_hi = oy_hi < Ho - 1 ? oy_hi : Ho - 1;
    ox_hi = ox_hi < Wo - 1 ? ox_hi : Wo - 1;
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            __syncthreads();
            const T* src = dcol + ((int64_t)(n * Ho + oy) * Wo + ox) * ldc + (int64_t)c0 * kk;
            for (int i = tid; i < cw * kk; i += 256) rowb[i] = src[i];
            __syncthreads();
            if (cl < cw) {
                const int by = oy * stride - pad - ty0, bx = ox * stride - pad - tx0;
                for (int t = qd; t < kk; t += 4) {
                    const int ky = t / k, kx = t - ky * k, py = by + ky, px = bx + kx;
                    if ((unsigned)py < (unsigned)TS && (unsigned)px < (unsigned)TS) acc[(py * TS + px) * CB + cl] += (float)rowb[cl * kk + t];
                }
            }
        }
    __syncthreads();
    for (int i = tid; i < TS * TS * CB; i += 256) {
        const int pix = i >> 6, c = i & 63, y = ty0 + (pix >> 3), x = tx0 + (pix & 7);
        if (c < cw && y < H && x < W) {
            T* d = dX + ((int64_t)(n * H + y) * W + x) * lddx + c0 + c;
            float v = acc[i];
            if (accumulate) v += (float)*d;
            *d = from_float<T>(v);
        }
    }
}
// Non-overlapping patches (k == stride, no padding: the spatial-reduction convs): a pure permutation, the mirror image of k_im2col --
// the dcol row is read as it lies, transposed through the same LDS tile and leaves pixel by pixel with contiguous channels.
template <typename T>
__global__ void k_col2im_patch(const T* __restrict__ dcol, int ldc, T* __restrict__ dX, int lddx, int H, int W, int C, int k, int Ho, int Wo,
                               int accumulate) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);
    const int pix = blockIdx.x;
    const int ox = pix % Wo, t_ = pix / Wo, oy = t_ % Ho, n = t_ / Ho;
    const int kk = k * k, P = C + 2;
    const T* src = dcol + (int64_t)pix * ldc;
    for (int e = threadIdx.x; e < kk * C; e += blockDim.x) {
        const int c = e / kk, t = e - c * kk;
        tile[t * P + c] = src[e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kk * C; e += blockDim.x) {
        const int t = e / C, c = e - t * C, ky = t / k, kx = t - ky * k;
        const int iy = oy * k + ky, ix = ox * k + kx;
        if (iy < H && ix < W) {
            T* d = dX + ((int64_t)(n * H + iy) * W + ix) * lddx + c;
            float v = (float)tile[t * P + c];
            if (accumulate) v += (float)*d;
            *d = from_float<T>(v);
        }
    }
}
void launch_col2im(int dt, const void* dcol, int ldc, void* dX, int lddx, int n, int H, int W, int C, int k, int stride, int pad, int Ho,
                   int Wo, int accumulate, hipStream_t s) {
    if (stride == k && pad == 0 && H == Ho * k && W == Wo * k && (size_t)k * k * (C + 2) * dsize(dt) <= 64 * 1024) {
        const size_t lds = (size_t)k * k * (C + 2) * dsize(dt);
        const unsigned grid = (unsigned)((int64_t)n * Ho * Wo);
        if (dt == BF16) k_col2im_patch<bf16><<<grid, 256, lds, s>>>((const bf16*)dcol, ldc, (bf16*)dX, lddx, H, W, C, k, Ho, Wo, accumulate);
        else k_col2im_patch<float><<<grid, 256, lds, s>>>((const float*)dcol, ldc, (float*)dX, lddx, H, W, C, k, Ho, Wo, accumulate);
        return;
    }
    static const bool no_tile = [] { const char* e = getenv("STCD_NO_COL2IM_TILE"); return e && e[0] == '1'; }();
    if (stride < k && !no_tile) {
        const int tx = (W + 7) / 8, ty = (H + 7) / 8;
        const size_t lds = (size_t)64 * 64 * 4 + (size_t)64 * k * k * dsize(dt);
        dim3 grid(tx * ty, n, (C + 63) / 64);
        if (dt == BF16) k_col2im_tile<bf16><<<grid, 256, lds, s>>>((const bf16*)dcol, ldc, (bf16*)dX, lddx, H, W, C, k, stride, pad, Ho, Wo, tx, accumulate);
        else k_col2im_tile<float><<<grid, 256, lds, s>>>((const float*)dcol, ldc, (float*)dX, lddx, H, W, C, k, stride, pad, Ho, Wo, tx, accumulate);
        return;
    }
    const int threads = std::min(256, ((C + 63) / 64) * 64);
    const unsigned grid = (unsigned)((int64_t)n * H * W);
    if (dt == BF16) k_col2im<bf16><<<grid, threads, 0, s>>>((const bf16*)dcol, ldc, (bf16*)dX, lddx, H, W, C, k, stride, pad, Ho, Wo, accumulate);
    else k_col2im<float><<<grid, threads, 0, s>>>((const float*)dcol, ldc, (float*)dX, lddx, H, W, C, k, stride, pad, Ho, Wo, accumulate);
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// nn.LayerNorm(C, eps) over the channels of every token (ChangeFormer.py:209,317,478,486,1363: eps 1e-5 for the patch-embedding /
// spatial-reduction norms, 1e-6 for the block / stage norms).  LPR lanes share one row (8 for C = 64 ... 64 for C >= 320), so a wave
// holds 64 / LPR rows at once (a 64-channel token is 128 bytes: one row per wave left 56 lanes idle and the kernel latency-bound);
// the row lives in registers, two-pass variance.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T, int LPR, int NP>
__global__ void __launch_bounds__(256)
k_ln_fwd(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, const float* __restrict__ gamma, const float* __restrict__ beta,
         float* __restrict__ stats, int64_t M, int C, float eps) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, np = C >> 3, rl = lane / LPR, pl = lane % LPR;
    const float invC = 1.f / (float)C;
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wid) * RPW; row0 < M; row0 += (int64_t)gridDim.x * 4 * RPW) {
        const int64_t row = row0 + rl;
        const bool rok = row < M;
        float v[NP][8];
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int piece = pl + LPR * p;
            if (rok && piece < np) {
                load8(x + row * ldx + piece * 8, v[p]);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[p][j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[p][j] = 0.f;
            }
        }
        const float mean = group_sum<LPR>(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (pl + LPR * p < np) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[p][j] - mean; q += d * d; }
            }
        const float rstd = 1.f / sqrtf(group_sum<LPR>(q) * invC + eps);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int piece = pl + LPR * p;
            if (rok && piece < np) {
                float g[8], b[8], o[8];
                load8(gamma + piece * 8, g);
                load8(beta + piece * 8, b);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (v[p][j] - mean) * rstd * g[j] + b[j];
                store8(y + row * ldy + piece * 8, o);
            }
        }
        if (rok && pl == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    }
}
static inline int ln_lpr(int C) { const int np = C / 8; return np <= 8 ? 8 : np <= 16 ? 16 : np <= 32 ? 32 : 64; }
void launch_layernorm(int dt, const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta, float* stats, int64_t M,
                      int C, float eps, hipStream_t s) {
    const int lpr = ln_lpr(C), rpb = 4 * (64 / lpr);
    const unsigned grid = (unsigned)std::min<int64_t>((M + rpb - 1) / rpb, 4096);
    if (grid == 0) return;
#define LN_F(T_, L_, NP_) k_ln_fwd<T_, L_, NP_><<<grid, 256, 0, s>>>((const T_*)x, ldx, (T_*)y, ldy, gamma, beta, stats, M, C, eps)
#define LN_FT(L_, NP_) do { if (dt == BF16) LN_F(bf16, L_, NP_); else LN_F(float, L_, NP_); } while (0)
    if (lpr == 8) LN_FT(8, 1);
    else if (lpr == 16) LN_FT(16, 1);
    else if (lpr == 32) LN_FT(32, 1);
    else if (C <= 512) LN_FT(64, 1);
    else LN_FT(64, 4);
#undef LN_FT
#undef LN_F
}

static inline int ln_bwd_blocks(int64_t M, int C) { const int rpb = 4 * (64 / ln_lpr(C)); return (int)std::max<int64_t>(1, std::min<int64_t>((M + rpb - 1) / rpb, 1024)); }
int64_t layernorm_bwd_scratch_floats(int64_t M, int C) { return (int64_t)ln_bwd_blocks(M, C) * 2 * C + 16; }

template <typename T, int LPR, int NP>
__global__ void __launch_bounds__(256)
k_ln_bwd(const T* __restrict__ dy, int lddy, const T* __restrict__ dy2, int lddy2, const T* __restrict__ x, int ldx,
         const float* __restrict__ stats, const float* __restrict__ gamma, const T* __restrict__ add, int ldadd, T* __restrict__ dx, int lddx,
         float* __restrict__ partial, int64_t M, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int RPW = 64 / LPR;
    float* red = reinterpret_cast<float*>(smem_raw);          // [4 waves * RPW row groups][2][C]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, np = C >> 3, rl = lane / LPR, pl = lane % LPR;
    const float invC = 1.f / (float)C;
    float dg[NP][8], db[NP][8], gm[NP][8];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { dg[p][j] = 0.f; db[p][j] = 0.f; gm[p][j] = 0.f; }
        if (pl + LPR * p < np) load8(gamma + (pl + LPR * p) * 8, gm[p]);
    }
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wid) * RPW; row0 < M; row0 += (int64_t)gridDim.x * 4 * RPW) {
        const int64_t row = row0 + rl;
        const bool rok = row < M;
        const float mean = rok ? stats[row * 2] : 0.f, rstd = rok ? stats[row * 2 + 1] : 0.f;
        float g[NP][8], xh[NP][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int piece = pl + LPR * p;
#pragma unroll
            for (int j = 0; j < 8; ++j) { g[p][j] = 0.f; xh[p][j] = 0.f; }
            if (rok && piece < np) {
                float d[8], xv[8];
                load8(dy + row * lddy + piece * 8, d);
                if (dy2) {
                    float d2[8];
                    load8(dy2 + row * lddy2 + piece * 8, d2);
#pragma unroll
                    for (int j = 0; j < 8; ++j) d[j] += d2[j];
                }
                load8(x + row * ldx + piece * 8, xv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xh[p][j] = (xv[j] - mean) * rstd;
                    g[p][j] = d[j] * gm[p][j];
                    s1 += g[p][j];
                    s2 += g[p][j] * xh[p][j];
                    dg[p][j] += d[j] * xh[p][j];
                    db[p][j] += d[j];
                }
            }
        }
        s1 = group_sum<LPR>(s1) * invC;
        s2 = group_sum<LPR>(s2) * invC;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int piece = pl + LPR * p;
            if (rok && piece < np) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = rstd * (g[p][j] - s1 - xh[p][j] * s2);
                if (add) {
                    float a8[8];
                    load8(add + row * ldadd + piece * 8, a8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += a8[j];
                }
                store8(dx + row * lddx + piece * 8, o);
            }
        }
    }
    const int slot = wid * RPW + rl;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int piece = pl + LPR * p;
        if (piece < np) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[(slot * 2 + 0) * C + piece * 8 + j] = dg[p][j];
                red[(slot * 2 + 1) * C + piece * 8 + j] = db[p][j];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float sm = 0.f;
        for (int k = 0; k < 4 * RPW; ++k) sm += red[k * 2 * C + i];
        partial[(int64_t)blockIdx.x * 2 * C + i] = sm;
    }
}
void launch_layernorm_bwd(int dt, const void* dy, int lddy, const void* dy2, int lddy2, const void* x, int ldx, const float* stats,
                          const float* gamma, const void* add, int ldadd, void* dx, int lddx, float* dgamma, float* dbeta,
                          float* scratch, int64_t M, int C, hipStream_t s) {
    const int grid = ln_bwd_blocks(M, C), lpr = ln_lpr(C);
    const size_t lds = (size_t)4 * (64 / lpr) * 2 * C * 4;
#define LN_B(T_, L_, NP_) k_ln_bwd<T_, L_, NP_><<<grid, 256, lds, s>>>((const T_*)dy, lddy, (const T_*)dy2, lddy2, (const T_*)x, ldx, stats, gamma, \
                                                                   (const T_*)add, ldadd, (T_*)dx, lddx, scratch, M, C)
#define LN_BT(L_, NP_) do { if (dt == BF16) LN_B(bf16, L_, NP_); else LN_B(float, L_, NP_); } while (0)
    if (lpr == 8) LN_BT(8, 1);
    else if (lpr == 16) LN_BT(16, 1);
    else if (lpr == 32) LN_BT(32, 1);
    else if (C <= 512) LN_BT(64, 1);
    else LN_BT(64, 4);
#undef LN_BT
#undef LN_B
    launch_partial_finish(scratch, grid, (int64_t)2 * C, C, 2 * C, dgamma, dbeta, s);
}

// ------------------------------------------------------------------------------------------------ column sums (bias gradients)
static inline int colsum_blocks(int64_t M) { return (int)std::max<int64_t>(1, std::min<int64_t>((M + 127) / 128, 2048)); }
int64_t colsum_scratch_floats(int64_t M, int C) { return (int64_t)colsum_blocks(M) * C + 16; }

template <typename T>
__global__ void __launch_bounds__(256)
k_colsum(const T* __restrict__ X, int ld, int64_t M, int C, float* __restrict__ partial) {
    __shared__ float red[256 * 8];
    const int np = C >> 3;
    const int PW = np < 256 ? np : 256, RL = 256 / PW;
    const int pl = threadIdx.x % PW, rl = threadIdx.x / PW;
    const int64_t rows_per = (M + gridDim.x - 1) / gridDim.x, r0 = (int64_t)blockIdx.x * rows_per, r1 = r0 + rows_per < M ? r0 + rows_per : M;
    for (int pc = 0; pc < np; pc += PW) {
        const int piece = pc + pl;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (rl < RL && piece < np)
            for (int64_t r = r0 + rl; r < r1; r += RL) {
                float v[8];
                load8(X + r * ld + piece * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[j];
        __syncthreads();
        for (int i = threadIdx.x; i < PW * 8; i += 256) {
            const int p2 = i >> 3, j = i & 7;
            float s = 0.f;
            for (int r = 0; r < RL; ++r) s += red[(r * PW + p2) * 8 + j];
            if (pc + p2 < np) partial[(int64_t)blockIdx.x * C + (pc + p2) * 8 + j] = s;
        }
        __syncthreads();
    }
}
void launch_colsum(int dt, const void* X, int ld, int64_t M, int C, float* out, float* scratch, hipStream_t s) {
    const int grid = colsum_blocks(M);
    if (dt == BF16) k_colsum<bf16><<<grid, 256, 0, s>>>((const bf16*)X, ld, M, C, scratch);
    else k_colsum<float><<<grid, 256, 0, s>>>((const float*)X, ld, M, C, scratch);
    launch_partial_finish(scratch, grid, (int64_t)C, C, C, out, out, s);
}

// grouped form: ALL bias gradients of a backward stage in two launches (a ChangeFormer step has ~100 of them, most of a few
// microseconds: launch-bound one by one).  Block -> job by binary search over the jobs' first blocks (uniform: scalar loads).
template <typename T>
__global__ void __launch_bounds__(256)
k_colsum_group(const ColsumJob* __restrict__ jobs, int njobs, const char* __restrict__ ws, float* __restrict__ partial) {
    __shared__ float red[256 * 8];
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].start_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ColsumJob& jb = jobs[lo];
    const int bidx = blockIdx.x - jb.start_block, C = jb.C, ld = jb.ld;
    const int64_t M = jb.M;
    const T* X = reinterpret_cast<const T*>(ws + jb.x_off);
    float* part = partial + jb.part_off;
    const int np = C >> 3;
    const int PW = np < 256 ? np : 256, RL = 256 / PW;
    const int pl = threadIdx.x % PW, rl = threadIdx.x / PW;
    const int64_t rows_per = (M + jb.nblocks - 1) / jb.nblocks, r0 = (int64_t)bidx * rows_per, r1 = r0 + rows_per < M ? r0 + rows_per : M;
    for (int pc = 0; pc < np; pc += PW) {
        const int piece = pc + pl;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (rl < RL && piece < np)
            for (int64_t r = r0 + rl; r < r1; r += RL) {
                float v[8];
                load8(X + r * ld + piece * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[j];
        __syncthreads();
        for (int i = threadIdx.x; i < PW * 8; i += 256) {
            const int p2 = i >> 3, j = i & 7;
            float sm = 0.f;
            for (int r = 0; r < RL; ++r) sm += red[(r * PW + p2) * 8 + j];
            if (pc + p2 < np) part[(int64_t)bidx * C + (pc + p2) * 8 + j] = sm;
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256)
k_colsum_group_finish(const ColsumJob* __restrict__ jobs, int njobs, const float* __restrict__ partial, float* __restrict__ grads) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].fin_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ColsumJob& jb = jobs[lo];
    const int j = ((int)blockIdx.x - jb.fin_start) * 8 + (threadIdx.x >> 5), bl = threadIdx.x & 31;
    float sm = 0.f;
    if (j < jb.C)
        for (int b = bl; b < jb.nblocks; b += 32) sm += partial[jb.part_off + (int64_t)b * jb.C + j];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    if (j < jb.C && bl == 0) grads[jb.out_off + j] = sm;
}
int colsum_job_blocks(int64_t M) { return colsum_blocks(M); }
void launch_colsum_group(int dt, const ColsumJob* jobs_dev, int njobs, int total_blocks, int fin_blocks, const char* ws, float* partial,
                         float* grads, hipStream_t s) {
    if (njobs <= 0) return;
    if (dt == BF16) k_colsum_group<bf16><<<total_blocks, 256, 0, s>>>(jobs_dev, njobs, ws, partial);
    else k_colsum_group<float><<<total_blocks, 256, 0, s>>>(jobs_dev, njobs, ws, partial);
    k_colsum_group_finish<<<fin_blocks, 256, 0, s>>>(jobs_dev, njobs, partial, grads);
}

// ------------------------------------------------------------------------------------------------ BatchNorm statistics, double precision
// The decoder's BatchNorm layers see FEW values per channel on small inputs (B x 1 x 1 at the deepest scale of a 32x32 tile): the
// fixed-point accumulators of the convolutional families (common.h: sum x^2 at 2^-18) resolve such a variance to a few per cent
// only.  Here the moments are summed in double (block partials, fixed-order finish), the variance is E[x^2] - mean^2 in double, and
// the kernel publishes what k_bn_act / k_bn_bwd_* read: stat = [mean, invstd, scale, shift][C]; running statistics as torch
// (momentum 0.1, unbiased variance).
template <typename T>
__global__ void __launch_bounds__(256)
k_chan_moments(const T* __restrict__ X, int ld, int64_t M, int C, double* __restrict__ partial) {
    __shared__ double red[256 * 8];
    const int np = C >> 3;
    const int PW = np < 256 ? np : 256, RL = 256 / PW;
    const int pl = threadIdx.x % PW, rl = threadIdx.x / PW;
    const int64_t rows_per = (M + gridDim.x - 1) / gridDim.x, r0 = (int64_t)blockIdx.x * rows_per, r1 = r0 + rows_per < M ? r0 + rows_per : M;
    for (int pc = 0; pc < np; pc += PW) {
        const int piece = pc + pl;
        double a1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (rl < RL && piece < np)
            for (int64_t r = r0 + rl; r < r1; r += RL) {
                float v[8];
                load8(X + r * ld + piece * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) { a1[j] += (double)v[j]; a2[j] += (double)v[j] * (double)v[j]; }
            }
        for (int which = 0; which < 2; ++which) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = which ? a2[j] : a1[j];
            __syncthreads();
            for (int i = threadIdx.x; i < PW * 8; i += 256) {
                const int p2 = i >> 3, j = i & 7;
                double sm = 0.0;
                for (int r = 0; r < RL; ++r) sm += red[(r * PW + p2) * 8 + j];
                if (pc + p2 < np) partial[((int64_t)blockIdx.x * 2 + which) * C + (pc + p2) * 8 + j] = sm;
            }
            __syncthreads();
        }
    }
}
__global__ void k_bn_finish(const double* __restrict__ partial, int nblk, int64_t M, int C, const float* __restrict__ gamma,
                            const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ stat,
                            float momentum, float eps) {
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), bl = threadIdx.x & 31;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int b = bl; b < nblk; b += 32) { s1 += partial[((int64_t)b * 2) * C + c]; s2 += partial[((int64_t)b * 2 + 1) * C + c]; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (c >= C || bl != 0) return;
    const double mean = s1 / (double)M;
    double var = s2 / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)(gamma[c] * invstd), sh = (float)(beta[c] - mean * gamma[c] * invstd);
    stat[c] = (float)mean; stat[C + c] = (float)invstd; stat[2 * C + c] = sc; stat[3 * C + c] = sh;
    const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
    rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
    rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
}
int64_t bn_precise_scratch_floats(int64_t M, int C) { return (int64_t)colsum_blocks(M) * 2 * C * 2 + 16; }
void launch_bn_stats_precise(int dt, const void* Z, int ld, int64_t M, int C, const float* gamma, const float* beta, float* rmean,
                             float* rvar, float* stat, float* scratch, float momentum, float eps, hipStream_t s) {
    const int grid = colsum_blocks(M);
    double* partial = reinterpret_cast<double*>(scratch);
    if (dt == BF16) k_chan_moments<bf16><<<grid, 256, 0, s>>>((const bf16*)Z, ld, M, C, partial);
    else k_chan_moments<float><<<grid, 256, 0, s>>>((const float*)Z, ld, M, C, partial);
    k_bn_finish<<<cdiv(C, 8), 256, 0, s>>>(partial, grid, M, C, gamma, beta, rmean, rvar, stat, momentum, eps);
}

// ------------------------------------------------------------------------------------------------ attention, plain-FMA kernels
// Attention.forward (ChangeFormer.py:334-358): attn = softmax(q k^T * scale) ; attn_drop ; x = attn v.  One wave = 64 queries of one
// (image, head); keys / values stream through LDS in chunks of 32 (every lane reads the same key: LDS broadcast); online softmax.
// The row's log-sum-exp is kept for the backward, which recomputes the probabilities (nothing of size N x Nkv is ever stored).
template <typename T, int DMAX>
__global__ void __launch_bounds__(64)
k_attn_fwd_ref(const T* __restrict__ q, int ldq, const T* __restrict__ kv, int ldkv, T* __restrict__ out, int ldo, float* __restrict__ lse, int N,
               int Nkv, int heads, int d, float scale, DropSite drop) {
    __shared__ float Ks[32][DMAX + 1], Vs[32][DMAX + 1];
    __shared__ float Sc[32][64];
    const int lane = threadIdx.x, h = blockIdx.y, img = blockIdx.z, C = heads * d;
    const int i = blockIdx.x * 64 + lane, ic = i < N ? i : N - 1;
    float qr[DMAX], acc[DMAX];
    {
        const T* qp = q + ((int64_t)img * N + ic) * ldq + h * d;
#pragma unroll
        for (int dd = 0; dd < DMAX; ++dd) { qr[dd] = dd < d ? (float)qp[dd] * scale : 0.f; acc[dd] = 0.f; }
    }
    float m = -INFINITY, l = 0.f;
    const uint32_t base = (uint32_t)(((int64_t)(img * heads + h) * N + ic) * Nkv);
    for (int j0 = 0; j0 < Nkv; j0 += 32) {
        __syncthreads();
        for (int e = lane; e < 32 * DMAX; e += 64) {
            const int j = e / DMAX, dd = e - j * DMAX;
            float kvv = 0.f, vv = 0.f;
            if (j0 + j < Nkv && dd < d) {
                const T* p = kv + ((int64_t)img * Nkv + j0 + j) * ldkv + h * d + dd;
                kvv = (float)p[0]; vv = (float)p[C];
            }
            Ks[j][dd] = kvv; Vs[j][dd] = vv;
        }
        __syncthreads();
        float cmax = -INFINITY;
        for (int j = 0; j < 32; ++j) {
            float sdot = 0.f;
#pragma unroll
            for (int dd = 0; dd < DMAX; ++dd) sdot += qr[dd] * Ks[j][dd];
            if (j0 + j >= Nkv) sdot = -INFINITY;
            Sc[j][lane] = sdot;
            cmax = fmaxf(cmax, sdot);
        }
        const float mnew = fmaxf(m, cmax), corr = __expf(m - mnew);
        l *= corr;
#pragma unroll
        for (int dd = 0; dd < DMAX; ++dd) acc[dd] *= corr;
        for (int j = 0; j < 32; ++j) {
            const float sv = Sc[j][lane];
            const float p = j0 + j < Nkv ? __expf(sv - mnew) : 0.f;
            l += p;
            const float pm = p * cf_keep(base + (uint32_t)(j0 + j), drop);
#pragma unroll
            for (int dd = 0; dd < DMAX; ++dd) acc[dd] += pm * Vs[j][dd];
        }
        m = mnew;
    }
    if (i < N) {
        const float inv = 1.f / l;
        T* op = out + ((int64_t)img * N + i) * ldo + h * d;
#pragma unroll
        for (int dd = 0; dd < DMAX; ++dd)
            if (dd < d) op[dd] = from_float<T>(acc[dd] * inv);
        lse[(int64_t)(img * heads + h) * N + i] = m + __logf(l);
    }
}
void launch_attn_fwd(int dt, const void* q, int ldq, const void* kv, int ldkv, void* out, int ldo, float* lse, int n, int N, int Nkv,
                     int heads, int d, float scale, DropSite drop, hipStream_t s) {
    if (attn_mfma_ok(dt, Nkv, d) && launch_attn_fwd_mfma(q, ldq, kv, ldkv, out, ldo, lse, n, N, Nkv, heads, d, scale, drop, s) == 0) return;
    dim3 grid(cdiv(N, 64), heads, n);
#define AT_F(T_, D_) k_attn_fwd_ref<T_, D_><<<grid, 64, 0, s>>>((const T_*)q, ldq, (const T_*)kv, ldkv, (T_*)out, ldo, lse, N, Nkv, heads, d, scale, drop)
    if (d <= 64) { if (dt == BF16) AT_F(bf16, 64); else AT_F(float, 64); }
    else { if (dt == BF16) AT_F(bf16, 128); else AT_F(float, 128); }
#undef AT_F
}

// backward, part 1: D_i = dO_i . O_i (kept for part 2), dQ_i = scale * sum_j dS_ij k_j with dS_ij = P_ij (mask_ij dO_i . v_j - D_i)
template <typename T, int DMAX>
__global__ void __launch_bounds__(64)
k_attn_bwd_dq_ref(const T* __restrict__ q, int ldq, const T* __restrict__ kv, int ldkv, const T* __restrict__ out, int ldo,
                  const T* __restrict__ dout, int lddo, const float* __restrict__ lse, T* __restrict__ dq, int lddq, float* __restrict__ Drow, int N,
                  int Nkv, int heads, int d, float scale, DropSite drop) {
    __shared__ float Ks[32][DMAX + 1], Vs[32][DMAX + 1];
    const int lane = threadIdx.x, h = blockIdx.y, img = blockIdx.z, C = heads * d;
    const int i = blockIdx.x * 64 + lane, ic = i < N ? i : N - 1;
    float qr[DMAX], dor[DMAX], dqr[DMAX];
    float Di = 0.f;
    {
        const T* qp = q + ((int64_t)img * N + ic) * ldq + h * d;
        const T* op = out + ((int64_t)img * N + ic) * ldo + h * d;
        const T* dp = dout + ((int64_t)img * N + ic) * lddo + h * d;
#pragma unroll
        for (int dd = 0; dd < DMAX; ++dd) {
            qr[dd] = dd < d ? (float)qp[dd] * scale : 0.f;
            dor[dd] = dd < d ? (float)dp[dd] : 0.f;
            Di += dd < d ? dor[dd] * (float)op[dd] : 0.f;
            dqr[dd] = 0.f;
        }
    }
    const float L = lse[(int64_t)(img * heads + h) * N + ic];
    const uint32_t base = (uint32_t)(((int64_t)(img * heads + h) * N + ic) * Nkv);
    for (int j0 = 0; j0 < Nkv; j0 += 32) {
        __syncthreads();
        for (int e = lane; e < 32 * DMAX; e += 64) {
            const int j = e / DMAX, dd = e - j * DMAX;
            float kvv = 0.f, vv = 0.f;
            if (j0 + j < Nkv && dd < d) {
                const T* p = kv + ((int64_t)img * Nkv + j0 + j) * ldkv + h * d + dd;
                kvv = (float)p[0]; vv = (float)p[C];
            }
            Ks[j][dd] = kvv; Vs[j][dd] = vv;
        }
        __syncthreads();
        const int jn = Nkv - j0 < 32 ? Nkv - j0 : 32;
        for (int j = 0; j < jn; ++j) {
            float sdot = 0.f, dpv = 0.f;
#pragma unroll
            for (int dd = 0; dd < DMAX; ++dd) { sdot += qr[dd] * Ks[j][dd]; dpv += dor[dd] * Vs[j][dd]; }
            const float p = __expf(sdot - L);
            const float ds = p * (dpv * cf_keep(base + (uint32_t)(j0 + j), drop) - Di);
#pragma unroll
            for (int dd = 0; dd < DMAX; ++dd) dqr[dd] += ds * Ks[j][dd];
        }
    }
    if (i < N) {
        T* o = dq + ((int64_t)img * N + i) * lddq + h * d;
#pragma unroll
        for (int dd = 0; dd < DMAX; ++dd)
            if (dd < d) o[dd] = from_float<T>(dqr[dd] * scale);
        Drow[(int64_t)(img * heads + h) * N + i] = Di;
    }
}
// part 2: one lane = one key of (image, head); the block walks a slice of the queries (staged through LDS, broadcast reads) and
// writes its partial dK / dV slab: dK_j = scale * sum_i dS_ij q_i, dV_j = sum_i P_ij mask_ij dO_i
template <typename T, int DMAX>
__global__ void __launch_bounds__(64)
k_attn_bwd_dkv_ref(const T* __restrict__ q, int ldq, const T* __restrict__ kv, int ldkv, const T* __restrict__ dout, int lddo,
                   const float* __restrict__ lse, const float* __restrict__ Drow, float* __restrict__ partial, int n, int N, int Nkv, int heads,
                   int d, float scale, DropSite drop, int qper) {
    __shared__ float Qs[16][DMAX + 1], Os[16][DMAX + 1];
    __shared__ float Ls[16], Ds[16];
    const int lane = threadIdx.x, C = heads * d;
    const int img = blockIdx.z / heads, h = blockIdx.z - img * heads;
    const int j = blockIdx.y * 64 + lane, jc = j < Nkv ? j : Nkv - 1;
    float kr[DMAX], vr[DMAX], dk[DMAX], dv[DMAX];
    {
        const T* p = kv + ((int64_t)img * Nkv + jc) * ldkv + h * d;
#pragma unroll
        for (int dd = 0; dd < DMAX; ++dd) {
            kr[dd] = dd < d ? (float)p[dd] : 0.f;
            vr[dd] = dd < d ? (float)p[C + dd] : 0.f;
            dk[dd] = 0.f; dv[dd] = 0.f;
        }
    }
    const int i_begin = blockIdx.x * qper, i_end = i_begin + qper < N ? i_begin + qper : N;
    for (int i0 = i_begin; i0 < i_end; i0 += 16) {
        __syncthreads();
        for (int e = lane; e < 16 * DMAX; e += 64) {
            const int ii = e / DMAX, dd = e - ii * DMAX;
            float qv = 0.f, ov = 0.f;
            if (i0 + ii < i_end && dd < d) {
                qv = (float)q[((int64_t)img * N + i0 + ii) * ldq + h * d + dd];
                ov = (float)dout[((int64_t)img * N + i0 + ii) * lddo + h * d + dd];
            }
            Qs[ii][dd] = qv; Os[ii][dd] = ov;
        }
        if (lane < 16) {
            const bool ok = i0 + lane < i_end;
            Ls[lane] = ok ? lse[(int64_t)(img * heads + h) * N + i0 + lane] : 0.f;
            Ds[lane] = ok ? Drow[(int64_t)(img * heads + h) * N + i0 + lane] : 0.f;
        }
        __syncthreads();
        const int in_ = i_end - i0 < 16 ? i_end - i0 : 16;
        for (int ii = 0; ii < in_; ++ii) {
            float sdot = 0.f, dpv = 0.f;
#pragma unroll
            for (int dd = 0; dd < DMAX; ++dd) { sdot += Qs[ii][dd] * kr[dd]; dpv += Os[ii][dd] * vr[dd]; }
            const float p = __expf(sdot * scale - Ls[ii]);
            const float keep = cf_keep((uint32_t)(((int64_t)(img * heads + h) * N + i0 + ii) * Nkv + jc), drop);
            const float ds = p * (dpv * keep - Ds[ii]) * scale, pm = p * keep;
#pragma unroll
            for (int dd = 0; dd < DMAX; ++dd) { dk[dd] += ds * Qs[ii][dd]; dv[dd] += pm * Os[ii][dd]; }
        }
    }
    if (j < Nkv) {
        float* o = partial + (((int64_t)blockIdx.x * n + img) * Nkv + j) * 2 * C + h * d;
#pragma unroll
        for (int dd = 0; dd < DMAX; ++dd)
            if (dd < d) { o[dd] = dk[dd]; o[C + dd] = dv[dd]; }
    }
}
template <typename T>
__global__ void k_sum_slabs(const float* __restrict__ partial, int nslab, int64_t count, int row_elems, T* __restrict__ out, int ldo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    for (int b = 0; b < nslab; ++b) s += partial[(int64_t)b * count + i];
    out[(i / row_elems) * ldo + (i % row_elems)] = from_float<T>(s);
}
static inline int attn_qsplit(int N) { return std::max(1, std::min(32, (N + 255) / 256)); }
int64_t attn_bwd_scratch_floats(int n, int N, int Nkv, int heads, int d) {
    const int64_t ref = (int64_t)n * heads * N + (int64_t)attn_qsplit(N) * n * Nkv * 2 * heads * d + 64;
    return std::max(ref, attn_mfma_bwd_scratch_floats(n, N, Nkv, heads, d));
}
void launch_attn_bwd(int dt, const void* q, int ldq, const void* kv, int ldkv, const void* out, int ldo, const void* dout, int lddo,
                     const float* lse, void* dq, int lddq, void* dkv, int lddkv, float* scratch, int n, int N, int Nkv, int heads, int d,
                     float scale, DropSite drop, hipStream_t s) {
    if (attn_mfma_ok(dt, Nkv, d) &&
        launch_attn_bwd_mfma(q, ldq, kv, ldkv, out, ldo, dout, lddo, lse, dq, lddq, dkv, lddkv, scratch, n, N, Nkv, heads, d, scale, drop, s) == 0)
        return;
    float* Drow = scratch;
    float* partial = scratch + (((int64_t)n * heads * N + 15) & ~(int64_t)15);
    const int QS = attn_qsplit(N), qper = ((N + QS - 1) / QS + 15) & ~15;
    dim3 g1(cdiv(N, 64), heads, n), g2(QS, cdiv(Nkv, 64), n * heads);
    const int64_t count = (int64_t)n * Nkv * 2 * heads * d;
#define AT_B(T_, D_)                                                                                                                     \
    do {                                                                                                                                 \
        k_attn_bwd_dq_ref<T_, D_><<<g1, 64, 0, s>>>((const T_*)q, ldq, (const T_*)kv, ldkv, (const T_*)out, ldo, (const T_*)dout, lddo, lse, \
                                                    (T_*)dq, lddq, Drow, N, Nkv, heads, d, scale, drop);                                   \
        k_attn_bwd_dkv_ref<T_, D_><<<g2, 64, 0, s>>>((const T_*)q, ldq, (const T_*)kv, ldkv, (const T_*)dout, lddo, lse, Drow, partial, n, N, \
                                                     Nkv, heads, d, scale, drop, qper);                                                    \
        k_sum_slabs<T_><<<cdiv(count, 256), 256, 0, s>>>(partial, QS, count, 2 * heads * d, (T_*)dkv, lddkv);                              \
    } while (0)
    if (d <= 64) { if (dt == BF16) AT_B(bf16, 64); else AT_B(float, 64); }
    else { if (dt == BF16) AT_B(bf16, 128); else AT_B(float, 128); }
#undef AT_B
}

// ------------------------------------------------------------------------------------------------ Mix-FFN middle
// Mlp.forward (ChangeFormer.py:287-295) between fc1 and fc2: DWConv (:512-523, Conv2d(Ch, Ch, 3, 1, 1, groups = Ch)) -> GELU (exact,
// erf) -> Dropout.  u (the pre-activation) is kept for the backward.
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}
template <typename T>
__global__ void __launch_bounds__(256)
k_dwgelu_fwd(const T* __restrict__ h, T* __restrict__ u, T* __restrict__ a, const float* __restrict__ w, const float* __restrict__ b, int H, int W,
             int Ch, int64_t total, DropSite drop) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int np = Ch >> 3, piece = (int)(idx % np);
    const int64_t pix = idx / np;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), c0 = piece * 8;
    float wv[72], acc[8];
    {
        const float4* wp = reinterpret_cast<const float4*>(w + (int64_t)c0 * 9);
#pragma unroll
        for (int k = 0; k < 18; ++k) { const float4 t = wp[k]; wv[4 * k] = t.x; wv[4 * k + 1] = t.y; wv[4 * k + 2] = t.z; wv[4 * k + 3] = t.w; }
        load8(b + c0, acc);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        if ((unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W) {
            float v[8];
            load8(h + (pix + dy * W + dx) * Ch + c0, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j] * wv[j * 9 + t];
        }
    }
    float o[8];
    const uint32_t e0 = (uint32_t)(pix * Ch + c0);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = gelu_f(round_as<T>(acc[j])) * cf_keep(e0 + j, drop);
    store8(u + pix * Ch + c0, acc);
    store8(a + pix * Ch + c0, o);
}
// The same, a thread = 8 channels x a strip of DW_P pixels of one row: the 72 filter values are loaded once per strip and every
// input piece serves up to three outputs (18 + 18 load instructions per 4 pixels instead of 4 x 27); branch-free (a neighbour outside
// the map re-reads a clamped address and is zeroed by a select), so all loads of a strip are in flight together.  Same tap order
// per output (dy major, dx minor) as the pixel kernel.  MEASURED: DESIGN.md section 3b.
constexpr int DW_P = 4;
template <typename T, bool BWD>
__global__ void __launch_bounds__(256)
k_dw_strip(const T* __restrict__ src, T* __restrict__ u, T* __restrict__ a, const float* __restrict__ w, const float* __restrict__ b, int H, int W,
           int WS, int Ch, int64_t total, DropSite drop) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int np = Ch >> 3, piece = (int)(idx % np);
    const int64_t strip = idx / np;
    const int xs = (int)(strip % WS);
    const int64_t row = strip / WS;                       // n * H + y
    const int y = (int)(row % H), x0 = xs * DW_P, c0 = piece * 8;
    float wv[72], acc[DW_P][8];
    {
        const float4* wp = reinterpret_cast<const float4*>(w + (int64_t)c0 * 9);
#pragma unroll
        for (int k = 0; k < 18; ++k) { const float4 t = wp[k]; wv[4 * k] = t.x; wv[4 * k + 1] = t.y; wv[4 * k + 2] = t.z; wv[4 * k + 3] = t.w; }
        float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (!BWD) load8(b + c0, b8);
#pragma unroll
        for (int p = 0; p < DW_P; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[p][j] = b8[j];
    }
    // forward: out(x) += in(y + dy, x + dx) w[dy][dx];  backward (data): dh(x) += g(y - dy, x - dx) w[dy][dx]
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
        const int dy = t3 - 1, yy = BWD ? y - dy : y + dy;
        const bool yok = (unsigned)yy < (unsigned)H;
        const int64_t rbase = (row + (yok ? yy - y : 0)) * W;
        uint4 raw[DW_P + 2];
#pragma unroll
        for (int c = 0; c < DW_P + 2; ++c) {             // columns x0 - 1 .. x0 + DW_P
            const int xx = x0 - 1 + c;
            const int xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            if constexpr (sizeof(T) == 2) raw[c] = *reinterpret_cast<const uint4*>(src + (rbase + xc) * Ch + c0);
        }
#pragma unroll
        for (int c = 0; c < DW_P + 2; ++c) {
            const int xx = x0 - 1 + c;
            const bool ok = yok && (unsigned)xx < (unsigned)W;
            float v[8];
            if constexpr (sizeof(T) == 2) {
                const uint32_t wd[4] = {raw[c].x, raw[c].y, raw[c].z, raw[c].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(wd[i] << 16); v[2 * i + 1] = __uint_as_float(wd[i] & 0xffff0000u); }
            } else {
                const int xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                load8(src + (rbase + xc) * Ch + c0, v);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? v[j] : 0.f;
            // this column is tap dx = (c - 1) - p of output p (forward), dx = p - (c - 1) (backward): in increasing dx per output
#pragma unroll
            for (int p = 0; p < DW_P; ++p) {
                const int dx = BWD ? p - (c - 1) : (c - 1) - p;
                if (dx >= -1 && dx <= 1) {
                    const int t = (dy + 1) * 3 + dx + 1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[p][j] += v[j] * wv[j * 9 + t];
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < DW_P; ++p) {
        if (x0 + p >= W) break;
        const int64_t pix = row * W + x0 + p;
        if (BWD) {
            store8(u + pix * Ch + c0, acc[p]);
        } else {
            float o[8];
            const uint32_t e0 = (uint32_t)(pix * Ch + c0);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = gelu_f(round_as<T>(acc[p][j])) * cf_keep(e0 + j, drop);
            store8(u + pix * Ch + c0, acc[p]);
            store8(a + pix * Ch + c0, o);
        }
    }
}
static bool dw_strip_on() {
    static const bool on = [] { const char* e = getenv("STCD_NO_DW_STRIP"); return !(e && e[0] == '1'); }();
    return on;
}
void launch_dwgelu_fwd(int dt, const void* h, void* u, void* a, const float* w, const float* b, int n, int H, int W, int Ch,
                       DropSite drop, hipStream_t s) {
    const int64_t total = (int64_t)n * H * W * (Ch / 8);
    if (dw_strip_on() && W >= DW_P) {
        const int WS = (W + DW_P - 1) / DW_P;
        const int64_t tot = (int64_t)n * H * WS * (Ch / 8);
        if (dt == BF16) k_dw_strip<bf16, false><<<cdiv(tot, 256), 256, 0, s>>>((const bf16*)h, (bf16*)u, (bf16*)a, w, b, H, W, WS, Ch, tot, drop);
        else k_dw_strip<float, false><<<cdiv(tot, 256), 256, 0, s>>>((const float*)h, (float*)u, (float*)a, w, b, H, W, WS, Ch, tot, drop);
        return;
    }
    if (dt == BF16) k_dwgelu_fwd<bf16><<<cdiv(total, 256), 256, 0, s>>>((const bf16*)h, (bf16*)u, (bf16*)a, w, b, H, W, Ch, total, drop);
    else k_dwgelu_fwd<float><<<cdiv(total, 256), 256, 0, s>>>((const float*)h, (float*)u, (float*)a, w, b, H, W, Ch, total, drop);
}

template <typename T>
__global__ void __launch_bounds__(256)
k_dwgelu_bwd_gate(const T* __restrict__ u, T* __restrict__ da, int64_t total8, DropSite drop) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total8) return;
    float uv[8], g[8];
    load8(u + idx * 8, uv);
    load8(da + idx * 8, g);
    const uint32_t e0 = (uint32_t)(idx * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = g[j] * cf_keep(e0 + j, drop) * gelu_grad_f(uv[j]);
    store8(da + idx * 8, g);
}
template <typename T>
__global__ void __launch_bounds__(256)
k_dw_bwd_data(const T* __restrict__ g, T* __restrict__ dh, const float* __restrict__ w, int H, int W, int Ch, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int np = Ch >> 3, piece = (int)(idx % np);
    const int64_t pix = idx / np;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), c0 = piece * 8;
    float wv[72], acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float4* wp = reinterpret_cast<const float4*>(w + (int64_t)c0 * 9);
#pragma unroll
    for (int k = 0; k < 18; ++k) { const float4 t = wp[k]; wv[4 * k] = t.x; wv[4 * k + 1] = t.y; wv[4 * k + 2] = t.z; wv[4 * k + 3] = t.w; }
#pragma unroll
    for (int t = 0; t < 9; ++t) {      // u[p] = sum_t h[p + off_t] w[t]  =>  dh[p'] = sum_t g[p' - off_t] w[t]
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        if ((unsigned)(y - dy) < (unsigned)H && (unsigned)(x - dx) < (unsigned)W) {
            float v[8];
            load8(g + (pix - dy * W - dx) * Ch + c0, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j] * wv[j * 9 + t];
        }
    }
    store8(dh + pix * Ch + c0, acc);
}
// filter / bias gradients: block = 8 pieces (64 channels: one 128-B line per pixel in bf16) x 32 pixel lanes over a pixel slice
constexpr int DW_PPB = 512;
template <typename T>
__global__ void __launch_bounds__(256)
k_dw_bwd_filter(const T* __restrict__ g, const T* __restrict__ h, float* __restrict__ partial, int H, int W, int Ch, int64_t npix) {
    __shared__ float red[256 * 8];
    const int pl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = (blockIdx.x * 8 + pl) * 8;
    const bool cok = c0 < Ch;
    const int64_t p0 = (int64_t)blockIdx.y * DW_PPB, p1 = p0 + DW_PPB < npix ? p0 + DW_PPB : npix;
    float acc[10][8];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    if (cok)
        for (int64_t pix = p0 + rl; pix < p1; pix += 32) {
            const int x = (int)(pix % W), y = (int)((pix / W) % H);
            float gv[8];
            load8(g + pix * Ch + c0, gv);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[9][j] += gv[j];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                if ((unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W) {
                    float hv[8];
                    load8(h + (pix + dy * W + dx) * Ch + c0, hv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[t][j] += gv[j] * hv[j];
                }
            }
        }
#pragma unroll
    for (int t = 0; t < 10; ++t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[t][j];
        __syncthreads();
        if (threadIdx.x < 64) {
            const int p2 = threadIdx.x >> 3, j = threadIdx.x & 7;
            float s = 0.f;
            for (int r = 0; r < 32; ++r) s += red[(r * 8 + p2) * 8 + j];
            const int c = (blockIdx.x * 8 + p2) * 8 + j;
            if (c < Ch) partial[((int64_t)blockIdx.y * Ch + c) * 10 + t] = s;
        }
        __syncthreads();
    }
}
// The same over strips of DW_P pixels of one row (as k_dw_strip: every h piece serves up to three taps of the strip's pixels, no
// branches around the loads): 6 + 3 x 6 load instructions per 4 pixels instead of 4 x 10.  A block's slice = DW_PPB / DW_P strips.
template <typename T>
__global__ void __launch_bounds__(256)
k_dw_bwd_filter_strip(const T* __restrict__ g, const T* __restrict__ h, float* __restrict__ partial, int H, int W, int WS, int Ch, int64_t nstrips) {
    __shared__ float red[256 * 8];
    const int pl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c0 = (blockIdx.x * 8 + pl) * 8;
    const bool cok = c0 < Ch;
    constexpr int SPB = DW_PPB / DW_P;
    const int64_t s0 = (int64_t)blockIdx.y * SPB, s1 = s0 + SPB < nstrips ? s0 + SPB : nstrips;
    float acc[10][8];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    if (cok)
        for (int64_t st = s0 + rl; st < s1; st += 32) {
            const int xs = (int)(st % WS);
            const int64_t row = st / WS;
            const int y = (int)(row % H), x0 = xs * DW_P;
            float gv[DW_P][8];
#pragma unroll
            for (int p = 0; p < DW_P; ++p) {
                const bool ok = x0 + p < W;
                load8(g + (row * W + (ok ? x0 + p : x0)) * Ch + c0, gv[p]);
#pragma unroll
                for (int j = 0; j < 8; ++j) { gv[p][j] = ok ? gv[p][j] : 0.f; acc[9][j] += gv[p][j]; }
            }
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
                const int dy = t3 - 1;
                const bool yok = (unsigned)(y + dy) < (unsigned)H;
                const int64_t rbase = (row + (yok ? dy : 0)) * W;
                float hv[DW_P + 2][8];
#pragma unroll
                for (int c = 0; c < DW_P + 2; ++c) {
                    const int xx = x0 - 1 + c;
                    const int xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                    load8(h + (rbase + xc) * Ch + c0, hv[c]);
                }
#pragma unroll
                for (int c = 0; c < DW_P + 2; ++c) {
                    const int xx = x0 - 1 + c;
                    const bool ok = yok && (unsigned)xx < (unsigned)W;
#pragma unroll
                    for (int p = 0; p < DW_P; ++p) {
                        const int dx = (c - 1) - p;          // dw[dy][dx] += g(x0 + p) h(y + dy, x0 + p + dx)
                        if (dx >= -1 && dx <= 1) {
                            const int t = (dy + 1) * 3 + dx + 1;
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[t][j] += ok ? gv[p][j] * hv[c][j] : 0.f;
                        }
                    }
                }
            }
        }
#pragma unroll
    for (int t = 0; t < 10; ++t) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[t][j];
        __syncthreads();
        if (threadIdx.x < 64) {
            const int p2 = threadIdx.x >> 3, j = threadIdx.x & 7;
            float s = 0.f;
            for (int r = 0; r < 32; ++r) s += red[(r * 8 + p2) * 8 + j];
            const int c = (blockIdx.x * 8 + p2) * 8 + j;
            if (c < Ch) partial[((int64_t)blockIdx.y * Ch + c) * 10 + t] = s;
        }
        __syncthreads();
    }
}
// one WAVE per output (c, t): its lanes stride over the nblk partial rows (fixed assignment), then a butterfly sum -- a thread per
// output walked up to 256 rows one dependent load at a time (62 us at stage 1; 13 launches, 0.27 ms per ChangeFormer step)
__global__ void __launch_bounds__(256)
k_dw_filter_finish(const float* __restrict__ partial, int nblk, int Ch, float* __restrict__ dw, float* __restrict__ db) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= Ch * 10) return;
    const int c = i / 10, t = i - c * 10;
    float s = 0.f;
    for (int b = lane; b < nblk; b += 64) s += partial[((int64_t)b * Ch + c) * 10 + t];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) { if (t < 9) dw[c * 9 + t] = s; else db[c] = s; }
}
int64_t dwgelu_bwd_scratch_floats(int n, int H, int W, int Ch) {
    const int64_t npix = (int64_t)n * H * W, nstrips = (int64_t)n * H * ((W + DW_P - 1) / DW_P);
    const int64_t slices = std::max((npix + DW_PPB - 1) / DW_PPB, (nstrips + DW_PPB / DW_P - 1) / (DW_PPB / DW_P));
    return slices * Ch * 10 + 16;
}
void launch_dwgelu_bwd(int dt, const void* h, const void* u, void* da, void* dh, const float* w, float* dw, float* db, float* scratch,
                       int n, int H, int W, int Ch, DropSite drop, hipStream_t s) {
    const int64_t npix = (int64_t)n * H * W, total = npix * (Ch / 8);
    const int pb = (int)((npix + DW_PPB - 1) / DW_PPB);
    dim3 gf(cdiv(Ch, 64), pb);
#define DW_B(T_)                                                                                                              \
    do {                                                                                                                      \
        k_dwgelu_bwd_gate<T_><<<cdiv(total, 256), 256, 0, s>>>((const T_*)u, (T_*)da, total, drop);                            \
        if (dw_strip_on() && W >= DW_P) {                                                                                     \
            const int WS_ = (W + DW_P - 1) / DW_P;                                                                                \
            const int64_t tot_ = (int64_t)n * H * WS_ * (Ch / 8);                                                                 \
            k_dw_strip<T_, true><<<cdiv(tot_, 256), 256, 0, s>>>((const T_*)da, (T_*)dh, nullptr, w, nullptr, H, W, WS_, Ch, tot_, drop); \
        } else                                                                                                                    \
            k_dw_bwd_data<T_><<<cdiv(total, 256), 256, 0, s>>>((const T_*)da, (T_*)dh, w, H, W, Ch, total);                        \
        if (dw_strip_on() && W >= DW_P) {                                                                                     \
            const int WS_ = (W + DW_P - 1) / DW_P;                                                                                \
            dim3 gfs(cdiv(Ch, 64), (unsigned)pbs);                                                                                \
            k_dw_bwd_filter_strip<T_><<<gfs, 256, 0, s>>>((const T_*)da, (const T_*)h, scratch, H, W, WS_, Ch, nstrips);           \
        } else                                                                                                                    \
            k_dw_bwd_filter<T_><<<gf, 256, 0, s>>>((const T_*)da, (const T_*)h, scratch, H, W, Ch, npix);                          \
    } while (0)
    const int64_t nstrips = (int64_t)n * H * ((W + DW_P - 1) / DW_P);
    const int pbs = (int)((nstrips + DW_PPB / DW_P - 1) / (DW_PPB / DW_P));
    const bool strip = dw_strip_on() && W >= DW_P;
    if (dt == BF16) DW_B(bf16); else DW_B(float);
#undef DW_B
    k_dw_filter_finish<<<cdiv(Ch * 10, 4), 256, 0, s>>>(scratch, strip ? pbs : pb, Ch, dw, db);
}

// ------------------------------------------------------------------------------------------------ residual + dropout + DropPath
// Block.forward (ChangeFormer.py:505-509): x = x + drop_path(f(norm(x))) where f ends in a Dropout (proj_drop :356, Mlp.drop :294).
// DropPath (timm, published definition): one Bernoulli(keep) draw per sample, divided by keep.
template <typename T, bool BWD>
__global__ void __launch_bounds__(256)
k_resid_drop(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ out, int64_t per_img8, int64_t total8, DropSite drop, DropSite path) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total8) return;
    const int img = (int)(idx / per_img8);
    const float ps = cf_keep((uint32_t)img, path);
    float yv[8], o[8];
    load8(y + idx * 8, yv);
    const uint32_t e0 = (uint32_t)(idx * 8);
    if (BWD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = yv[j] * ps * cf_keep(e0 + j, drop);
    } else {
        float xv[8];
        load8(x + idx * 8, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = xv[j] + yv[j] * ps * cf_keep(e0 + j, drop);
    }
    store8(out + idx * 8, o);
}
void launch_resid_drop(int dt, const void* x, const void* y, void* out, int n, int64_t rows_per_img, int C, DropSite drop, DropSite path,
                       hipStream_t s) {
    const int64_t per = rows_per_img * C / 8, total = per * n;
    if (dt == BF16) k_resid_drop<bf16, false><<<cdiv(total, 256), 256, 0, s>>>((const bf16*)x, (const bf16*)y, (bf16*)out, per, total, drop, path);
    else k_resid_drop<float, false><<<cdiv(total, 256), 256, 0, s>>>((const float*)x, (const float*)y, (float*)out, per, total, drop, path);
}
void launch_resid_drop_bwd(int dt, const void* dout, void* dy, int n, int64_t rows_per_img, int C, DropSite drop, DropSite path,
                           hipStream_t s) {
    const int64_t per = rows_per_img * C / 8, total = per * n;
    if (dt == BF16) k_resid_drop<bf16, true><<<cdiv(total, 256), 256, 0, s>>>(nullptr, (const bf16*)dout, (bf16*)dy, per, total, drop, path);
    else k_resid_drop<float, true><<<cdiv(total, 256), 256, 0, s>>>(nullptr, (const float*)dout, (float*)dy, per, total, drop, path);
}

// ------------------------------------------------------------------------------------------------ bilinear resize
// F.interpolate(mode="bilinear", align_corners=False) (ChangeFormer.py:1585,1591,1599,1607; `resize` :238-257): source index
// max(scale * (dst + 0.5) - 0.5, 0), scale = in / out; the second tap is clamped to the last row / column.
__device__ __forceinline__ void bil_src(int o, float scale, int in, int* i0, int* i1, float* l1) {
    float sf = scale * ((float)o + 0.5f) - 0.5f;
    sf = sf < 0.f ? 0.f : sf;
    int a = (int)sf;
    a = a < in - 1 ? a : in - 1;
    *i0 = a;
    *i1 = a + (a < in - 1 ? 1 : 0);
    *l1 = sf - (float)a;
}
template <typename T>
__global__ void __launch_bounds__(256)
k_bilinear(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd, int h, int w, int H, int W, int C, int64_t total, int accumulate) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int np = C >> 3, piece = (int)(idx % np);
    const int64_t pix = idx / np;
    const int ox = (int)(pix % W), oy = (int)((pix / W) % H), n = (int)(pix / ((int64_t)W * H));
    int y0, y1, x0, x1; float ly, lx;
    bil_src(oy, (float)h / (float)H, h, &y0, &y1, &ly);
    bil_src(ox, (float)w / (float)W, w, &x0, &x1, &lx);
    const T* b = src + (int64_t)n * h * w * lds + piece * 8;
    float v00[8], v01[8], v10[8], v11[8], o[8];
    load8(b + ((int64_t)y0 * w + x0) * lds, v00);
    load8(b + ((int64_t)y0 * w + x1) * lds, v01);
    load8(b + ((int64_t)y1 * w + x0) * lds, v10);
    load8(b + ((int64_t)y1 * w + x1) * lds, v11);
    const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = hy * (hx * v00[j] + lx * v01[j]) + ly * (hx * v10[j] + lx * v11[j]);
    T* d = dst + pix * ldd + piece * 8;
    if (accumulate) {
        float e[8];
        load8(d, e);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += e[j];
    }
    store8(d, o);
}
void launch_bilinear(int dt, const void* src, int lds, void* dst, int ldd, int n, int h, int w, int H, int W, int C, int accumulate,
                     hipStream_t s) {
    const int64_t total = (int64_t)n * H * W * (C / 8);
    if (dt == BF16) k_bilinear<bf16><<<cdiv(total, 256), 256, 0, s>>>((const bf16*)src, lds, (bf16*)dst, ldd, h, w, H, W, C, total, accumulate);
    else k_bilinear<float><<<cdiv(total, 256), 256, 0, s>>>((const float*)src, lds, (float*)dst, ldd, h, w, H, W, C, total, accumulate);
}
// gradient in gather form: a source pixel collects from every destination pixel whose two taps include it (fixed order: reproducible)
template <typename T>
__global__ void __launch_bounds__(256)
k_bilinear_bwd(const T* __restrict__ ddst, int ldd, T* __restrict__ dsrc, int lds, int h, int w, int H, int W, int C, int64_t total, int accumulate) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int np = C >> 3, piece = (int)(idx % np);
    const int64_t pix = idx / np;
    const int ix = (int)(pix % w), iy = (int)((pix / w) % h), n = (int)(pix / ((int64_t)w * h));
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    // destination rows that can touch source row iy: source index in (iy - 1, iy + 1)  <=>  o in ((iy - 0.5) / s - 0.5, (iy + 1.5) / s - 0.5)
    int oy_lo = (int)floorf(((float)iy - 0.5f) / sy - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / sy - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)ix - 0.5f) / sx - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / sx - 0.5f) + 1;
    oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
    oy_hi = oy_hi > H - 1 ? H - 1 : oy_hi; ox_hi = ox_hi > W - 1 ? W - 1 : ox_hi;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const T* b = ddst + (int64_t)n * H * W * ldd + piece * 8;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1; float ly;
        bil_src(oy, sy, h, &y0, &y1, &ly);
        const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            int x0, x1; float lx;
            bil_src(ox, sx, w, &x0, &x1, &lx);
            const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
            if (wx == 0.f) continue;
            float v[8];
            load8(b + ((int64_t)oy * W + ox) * ldd, v);
            const float wgt = wy * wx;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += wgt * v[j];
        }
    }
    T* d = dsrc + pix * lds + piece * 8;
    if (accumulate) {
        float e[8];
        load8(d, e);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += e[j];
    }
    store8(d, acc);
}
void launch_bilinear_bwd(int dt, const void* ddst, int ldd, void* dsrc, int lds, int n, int h, int w, int H, int W, int C, int accumulate,
                         hipStream_t s) {
    const int64_t total = (int64_t)n * h * w * (C / 8);
    if (dt == BF16) k_bilinear_bwd<bf16><<<cdiv(total, 256), 256, 0, s>>>((const bf16*)ddst, ldd, (bf16*)dsrc, lds, h, w, H, W, C, total, accumulate);
    else k_bilinear_bwd<float><<<cdiv(total, 256), 256, 0, s>>>((const float*)ddst, ldd, (float*)dsrc, lds, h, w, H, W, C, total, accumulate);
}

// ------------------------------------------------------------------------------------------------ elementwise family
// mode 0: PReLU(y; alpha)            (conv_diff, ChangeFormer.py:1141,1145: nn.PReLU(), one shared slope)
// mode 1: x * dropout(index)         (conv_diff's nn.Dropout(p = 0.6), :1143,1147)
// mode 2: relu(x)                    (ResidualBlock, ChangeFormerBaseNetworks.py:117)
// mode 3: dy * [a > 0]               (its gradient, from the ReLU's output)
// mode 4: alpha * x + beta * y       (ResidualBlock: conv2(.) * 0.1 + residual, :118-119; gradient sums)
// mode 5: dz * (y > 0 ? 1 : alpha)   (PReLU gradient w.r.t. its input)
template <typename T, int MODE>
__global__ void __launch_bounds__(256)
k_ew(const T* __restrict__ x, int ldx, const T* __restrict__ y, int ldy, T* __restrict__ out, int ldo, int C, int64_t total, float alpha, float beta,
     const float* __restrict__ aptr, DropSite drop) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int np = C >> 3, piece = (int)(idx % np);
    const int64_t row = idx / np;
    float xv[8], yv[8], o[8];
    load8(x + row * ldx + piece * 8, xv);
    if (MODE == 3 || MODE == 5 || (MODE == 4 && y)) load8(y + row * ldy + piece * 8, yv);
    const float al = (MODE == 0 || MODE == 5) ? aptr[0] : alpha;
    const uint32_t e0 = (uint32_t)(row * C + piece * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (MODE == 0) o[j] = xv[j] > 0.f ? xv[j] : al * xv[j];
        else if (MODE == 1) o[j] = xv[j] * cf_keep(e0 + j, drop);
        else if (MODE == 2) o[j] = xv[j] > 0.f ? xv[j] : 0.f;
        else if (MODE == 3) o[j] = yv[j] > 0.f ? xv[j] : 0.f;
        else if (MODE == 4) o[j] = al * xv[j] + (y ? beta * yv[j] : 0.f);
        else o[j] = yv[j] > 0.f ? xv[j] : al * xv[j];
    }
    store8(out + row * ldo + piece * 8, o);
}
#define EW_LAUNCH(MODE_, X_, LDX_, Y_, LDY_, O_, LDO_, A_, B_, AP_, DS_)                                                                 \
    do {                                                                                                                               \
        const int64_t total = rows * (C / 8);                                                                                          \
        if (total == 0) break;                                                                                                         \
        if (dt == BF16) k_ew<bf16, MODE_><<<cdiv(total, 256), 256, 0, s>>>((const bf16*)(X_), LDX_, (const bf16*)(Y_), LDY_, (bf16*)(O_), LDO_, C, total, A_, B_, AP_, DS_); \
        else k_ew<float, MODE_><<<cdiv(total, 256), 256, 0, s>>>((const float*)(X_), LDX_, (const float*)(Y_), LDY_, (float*)(O_), LDO_, C, total, A_, B_, AP_, DS_); \
    } while (0)
void launch_prelu(int dt, const void* y, int ldy, void* z, int ldz, const float* alpha, int64_t rows, int C, hipStream_t s) {
    EW_LAUNCH(0, y, ldy, nullptr, 0, z, ldz, 0.f, 0.f, alpha, DropSite());
}
void launch_dropout_ew(int dt, const void* x, int ldx, void* out, int ldo, int64_t rows, int C, DropSite drop, hipStream_t s) {
    EW_LAUNCH(1, x, ldx, nullptr, 0, out, ldo, 0.f, 0.f, nullptr, drop);
}
void launch_relu(int dt, const void* x, int ldx, void* out, int ldo, int64_t rows, int C, hipStream_t s) {
    EW_LAUNCH(2, x, ldx, nullptr, 0, out, ldo, 0.f, 0.f, nullptr, DropSite());
}
void launch_relu_bwd(int dt, const void* dy, int lddy, const void* a, int lda, void* dx, int lddx, int64_t rows, int C, hipStream_t s) {
    EW_LAUNCH(3, dy, lddy, a, lda, dx, lddx, 0.f, 0.f, nullptr, DropSite());
}
void launch_axpby(int dt, float alpha, const void* x, int ldx, float beta, const void* y, int ldy, void* out, int ldo, int64_t rows, int C,
                  hipStream_t s) {
    EW_LAUNCH(4, x, ldx, y, ldy, out, ldo, alpha, beta, nullptr, DropSite());
}
// d(alpha) = sum dz * y * [y <= 0]: block partials, fixed-order finish
template <typename T>
__global__ void __launch_bounds__(256)
k_prelu_dalpha(const T* __restrict__ dz, int lddz, const T* __restrict__ y, int ldy, int C, int64_t total, double* __restrict__ partial) {
    __shared__ double red[256];
    double acc = 0.0;
    const int np = C >> 3;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int piece = (int)(idx % np);
        const int64_t row = idx / np;
        float g[8], v[8];
        load8(dz + row * lddz + piece * 8, g);
        load8(y + row * ldy + piece * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j] > 0.f ? 0.0 : (double)g[j] * (double)v[j];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(64) k_sum_doubles(const double* __restrict__ partial, int n, float* __restrict__ out) {
    double s = 0.0;
    for (int b = threadIdx.x; b < n; b += 64) s += partial[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[0] = (float)s;
}
void launch_prelu_bwd(int dt, const void* dz, int lddz, const void* y, int ldy, void* dy, int lddy, const float* alpha, float* dalpha,
                      float* scratch, int64_t rows, int C, hipStream_t s) {
    const int64_t tot = rows * (C / 8);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((tot + 255) / 256, 1024));
    double* part = reinterpret_cast<double*>(scratch);
    if (dt == BF16) k_prelu_dalpha<bf16><<<grid, 256, 0, s>>>((const bf16*)dz, lddz, (const bf16*)y, ldy, C, tot, part);
    else k_prelu_dalpha<float><<<grid, 256, 0, s>>>((const float*)dz, lddz, (const float*)y, ldy, C, tot, part);
    k_sum_doubles<<<1, 64, 0, s>>>(part, grid, dalpha);
    EW_LAUNCH(5, dz, lddz, y, ldy, dy, lddy, 0.f, 0.f, alpha, DropSite());      // after the reduction: dy may alias dz
}
#undef EW_LAUNCH

// ------------------------------------------------------------------------------------------------ auxiliary prediction heads
// make_prediction (ChangeFormer.py:1151-1157) after its first conv: ReLU -> BatchNorm2d(C) -> Conv2d(C, C, 3, padding 1) on fp32
// NCHW maps with C = output classes (<= 8).  Tiny maps ([B, 2, H/32 .. H/4, ..]): one block computes the batch statistics.
__global__ void __launch_bounds__(1024)
k_aux_stats(const float* __restrict__ y, const float* __restrict__ bn_w, const float* __restrict__ bn_b, float* __restrict__ rmean,
            float* __restrict__ rvar, float* __restrict__ stat, int n, int C, int64_t HW, int training) {
    __shared__ double r1[1024], r2[1024];
    for (int c = 0; c < C; ++c) {
        float mean, var;
        if (training) {
            double s1 = 0.0, s2 = 0.0;
            for (int64_t i = threadIdx.x; i < (int64_t)n * HW; i += 1024) {
                const int img = (int)(i / HW);
                const float v = fmaxf(y[((int64_t)img * C + c) * HW + (i - img * HW)], 0.f);
                s1 += v; s2 += (double)v * v;
            }
            r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
            __syncthreads();
            for (int o = 512; o > 0; o >>= 1) {
                if ((int)threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
                __syncthreads();
            }
            const double cnt = (double)n * HW, m = r1[0] / cnt, vv = fmax(r2[0] / cnt - m * m, 0.0);
            mean = (float)m; var = (float)vv;
            if (threadIdx.x == 0) {
                rmean[c] = 0.9f * rmean[c] + 0.1f * mean;
                rvar[c] = 0.9f * rvar[c] + 0.1f * (float)(vv * cnt / fmax(cnt - 1.0, 1.0));
            }
            __syncthreads();
        } else { mean = rmean[c]; var = rvar[c]; }
        if (threadIdx.x == 0) {
            const float inv = 1.f / sqrtf(var + 1e-5f);
            stat[2 * c] = bn_w[c] * inv;
            stat[2 * c + 1] = bn_b[c] - mean * bn_w[c] * inv;
            stat[16 + 2 * c] = mean;           // for the backward (launch_aux_head_bwd)
            stat[16 + 2 * c + 1] = inv;
        }
    }
}
__global__ void __launch_bounds__(256)
k_aux_conv(const float* __restrict__ y, const float* __restrict__ stat, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out,
           int C, int H, int W, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % W), yy = (int)((idx / W) % H), co = (int)((idx / ((int64_t)W * H)) % C), n = (int)(idx / ((int64_t)W * H * C));
    float acc = b[co];
    for (int ci = 0; ci < C; ++ci) {
        const float sc = stat[2 * ci], sh = stat[2 * ci + 1];
        const float* p = y + ((int64_t)n * C + ci) * H * W;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = yy + ky - 1, ix = x + kx - 1;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                    acc += (fmaxf(p[(int64_t)iy * W + ix], 0.f) * sc + sh) * w[((co * C + ci) * 3 + ky) * 3 + kx];
            }
    }
    out[idx] = acc;
}
void launch_aux_head(const float* y, float* out, const float* bn_w, const float* bn_b, float* running_mean, float* running_var,
                     const float* w, const float* b, float* stat, int n, int C, int H, int W, int training, hipStream_t s) {
    k_aux_stats<<<1, 1024, 0, s>>>(y, bn_w, bn_b, running_mean, running_var, stat, n, C, (int64_t)H * W, training);
    const int64_t total = (int64_t)n * C * H * W;
    k_aux_conv<<<cdiv(total, 256), 256, 0, s>>>(y, stat, w, b, out, C, H, W, total);
}

// ---- backward of the auxiliary head (multi_scale_train, models/trainer.py:300-309): everything on the tiny fp32 NCHW maps, in a
//      fixed summation order (one block per reduced value, double-precision tree): conv3's data / filter / bias gradients, the
//      BatchNorm backward with batch statistics, the ReLU gate.  dy = gradient of the first conv's output (the engine packs it and
//      runs that conv's weight / data gradient on the MFMA kernels).
// dz[n, ci, q] = sum_{co, k} g[n, co, q - (k - 1)] * w3[co][ci][k]
__global__ void __launch_bounds__(256)
k_aux_bwd_dz(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dz, int C, int H, int W, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % W), yy = (int)((idx / W) % H), ci = (int)((idx / ((int64_t)W * H)) % C), n = (int)(idx / ((int64_t)W * H * C));
    float acc = 0.f;
    for (int co = 0; co < C; ++co) {
        const float* p = g + ((int64_t)n * C + co) * H * W;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int oy = yy - (ky - 1), ox = x - (kx - 1);
                if ((unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W) acc += p[(int64_t)oy * W + ox] * w[((co * C + ci) * 3 + ky) * 3 + kx];
            }
    }
    dz[idx] = acc;
}
// block b < C*C*9: dw3[co][ci][ky][kx] = sum g[n, co, p] * z[n, ci, p + k - 1] (z = relu(y)*sc + sh inside the map, 0 outside);
// C*C*9 <= b < C*C*9 + C: db3[co] = sum g;  then per channel c: S1 = sum dz, S2 = sum dz * rhat (rhat = (relu(y) - mean) * inv) ->
// dbeta, dgamma and sums[2c], sums[2c + 1] for the elementwise pass
__global__ void __launch_bounds__(256)
k_aux_bwd_reduce(const float* __restrict__ y, const float* __restrict__ stat, const float* __restrict__ g, const float* __restrict__ dz,
                 float* __restrict__ dw3, float* __restrict__ db3, float* __restrict__ dgamma, float* __restrict__ dbeta,
                 float* __restrict__ sums, int n, int C, int H, int W) {
    __shared__ double r1[256], r2[256];
    const int b = blockIdx.x, nw = C * C * 9;
    const int64_t HW = (int64_t)H * W, tot = (int64_t)n * HW;
    double a1 = 0.0, a2 = 0.0;
    if (b < nw) {
        const int kx = b % 3, ky = (b / 3) % 3, ci = (b / 9) % C, co = b / (9 * C);
        const float sc = stat[2 * ci], sh = stat[2 * ci + 1];
        for (int64_t i = threadIdx.x; i < tot; i += 256) {
            const int img = (int)(i / HW);
            const int64_t p = i - (int64_t)img * HW;
            const int py = (int)(p / W) + ky - 1, px = (int)(p % W) + kx - 1;
            if ((unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W)
                a1 += (double)g[((int64_t)img * C + co) * HW + p] * (double)(fmaxf(y[((int64_t)img * C + ci) * HW + (int64_t)py * W + px], 0.f) * sc + sh);
        }
    } else if (b < nw + C) {
        const int co = b - nw;
        for (int64_t i = threadIdx.x; i < tot; i += 256) { const int img = (int)(i / HW); a1 += (double)g[((int64_t)img * C + co) * HW + (i - (int64_t)img * HW)]; }
    } else {
        const int c = b - nw - C;
        const float mean = stat[16 + 2 * c], inv = stat[16 + 2 * c + 1];
        for (int64_t i = threadIdx.x; i < tot; i += 256) {
            const int img = (int)(i / HW);
            const int64_t at = ((int64_t)img * C + c) * HW + (i - (int64_t)img * HW);
            const double d = dz[at];
            a1 += d; a2 += d * (double)((fmaxf(y[at], 0.f) - mean) * inv);
        }
    }
    r1[threadIdx.x] = a1; r2[threadIdx.x] = a2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (b < nw) dw3[b] = (float)r1[0];
        else if (b < nw + C) db3[b - nw] = (float)r1[0];
        else { const int c = b - nw - C; dbeta[c] = (float)r1[0]; dgamma[c] = (float)r2[0]; sums[2 * c] = (float)(r1[0] / (double)tot); sums[2 * c + 1] = (float)(r2[0] / (double)tot); }
    }
}
// dy = (y > 0) ? gamma * inv * (dz - mean(dz) - rhat * mean(dz * rhat)) : 0
__global__ void __launch_bounds__(256)
k_aux_bwd_dy(const float* __restrict__ y, const float* __restrict__ stat, const float* __restrict__ dz, const float* __restrict__ sums,
             float* __restrict__ dy, int C, int64_t HW, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)((idx / HW) % C);
    const float v = y[idx], mean = stat[16 + 2 * c], inv = stat[16 + 2 * c + 1];
    const float rhat = (fmaxf(v, 0.f) - mean) * inv;
    dy[idx] = v > 0.f ? stat[2 * c] * (dz[idx] - sums[2 * c] - rhat * sums[2 * c + 1]) : 0.f;
}
void launch_aux_head_bwd(const float* y, const float* stat, const float* g, const float* w3, float* dz, float* dy, float* sums, float* dw3,
                         float* db3, float* dgamma, float* dbeta, int n, int C, int H, int W, hipStream_t s) {
    const int64_t total = (int64_t)n * C * H * W;
    k_aux_bwd_dz<<<cdiv(total, 256), 256, 0, s>>>(g, w3, dz, C, H, W, total);
    k_aux_bwd_reduce<<<C * C * 9 + 2 * C, 256, 0, s>>>(y, stat, g, dz, dw3, db3, dgamma, dbeta, sums, n, C, H, W);
    k_aux_bwd_dy<<<cdiv(total, 256), 256, 0, s>>>(y, stat, dz, sums, dy, C, (int64_t)H * W, total);
}

}  // namespace stcd
