// kernels_xconc.hip -- the pairwise depthwise 3x3 convolution of SiamUnet_cross_conc's skip blocks (gfx950).
//
// cross_conc.forward (/root/reference/models/SiamUnet_crossconc.py:24-33) interleaves the two dates' C-channel skip maps into
// 2C channels (inputs[:, 0::2] = x1, inputs[:, 1::2] = x2) and runs nn.Conv2d(2C, C, 3, padding=1, groups=C) on them (:14-18):
// output channel c sees exactly x1[c] and x2[c],
//     G[n, c, p] = bias[c] + sum_t  w[c][0][t] * x1[n, c, p + t]  +  w[c][1][t] * x2[n, c, p + t]          (w: [C][2][3][3])
// i.e. two depthwise convolutions added up -- HBM-bound elementwise work: thread = (pixel, 8 channels), 16-B accesses on the NHWC
// maps as they sit in the engine (the two dates `goff` elements apart), nothing interleaved in memory.  Backward: the two data
// gradients in one pass, the filter gradient as per-block partials (fixed summation order) + a finish launch.
#include "common.h"

namespace stcd {

static inline int xc_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// the block's copy of the filter, transposed to [k = d*9 + t][c] so a thread reads its 8 channels of a tap as two float4
// (read per thread from global, the 144 scalar weight loads were the whole kernel: 100 us instead of ~25 per full-resolution level)
__device__ __forceinline__ void pairdw_stage_weights(float* wl, const float* __restrict__ w, int C) {
    for (int i = threadIdx.x; i < C * 18; i += blockDim.x) {
        const int c = i / 18, k = i - c * 18;
        wl[k * C + c] = w[i];
    }
    __syncthreads();
}

template <typename T>
__global__ void __launch_bounds__(256)
k_pairdw_fwd(const T* __restrict__ A, int lda, int64_t goff, T* __restrict__ G, int ldg, const float* __restrict__ w,
             const float* __restrict__ b, int H, int W, int C, int64_t total) {
    extern __shared__ float wl[];                        // [18][C]
    pairdw_stage_weights(wl, w, C);
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cb = C >> 3;
    const int c0 = (int)(idx % cb) * 8;
    const int64_t pix = idx / cb;                       // (n * H + y) * W + x
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = b ? b[c0 + j] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
            const int64_t q = pix + (int64_t)(ky - 1) * W + (kx - 1);
            float a1[8], a2[8];
            load8<T>(A + q * lda + c0, a1);
            load8<T>(A + goff + q * lda + c0, a2);
            const float* w1 = wl + (ky * 3 + kx) * C + c0;
            const float* w2 = w1 + 9 * C;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w1[j] * a1[j] + w2[j] * a2[j];
        }
    store8<T>(G + pix * ldg + c0, acc);
}

// dA_d[n, c, q] = sum_t w[c][d][t] * dG[n, c, q - t]   for both dates d (the tensors `goff` elements apart)
template <typename T>
__global__ void __launch_bounds__(256)
k_pairdw_bwd_data(const T* __restrict__ dG, int lddg, T* __restrict__ dA, int ldda, int64_t goff, const float* __restrict__ w, int H,
                  int W, int C, int64_t total) {
    extern __shared__ float wl[];                        // [18][C]
    pairdw_stage_weights(wl, w, C);
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cb = C >> 3;
    const int c0 = (int)(idx % cb) * 8;
    const int64_t pix = idx / cb;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    float d1[8], d2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) d1[j] = d2[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y - (ky - 1), xx = x - (kx - 1);          // the output position this input pixel fed through tap (ky, kx)
            if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
            float g[8];
            load8<T>(dG + (pix - (int64_t)(ky - 1) * W - (kx - 1)) * lddg + c0, g);
            const float* w1 = wl + (ky * 3 + kx) * C + c0;
            const float* w2 = w1 + 9 * C;
#pragma unroll
            for (int j = 0; j < 8; ++j) { d1[j] += w1[j] * g[j]; d2[j] += w2[j] * g[j]; }
        }
    store8<T>(dA + pix * ldda + c0, d1);
    store8<T>(dA + goff + pix * ldda + c0, d2);
}

// filter gradient: block (channel block cb, pixel chunk ch) -> partial[(ch * CB + cb)][18][8]; dw[c][d][t] = sum over pixels of
// dG[p][c] * A_d[p + t][c]
template <typename T>
__global__ void __launch_bounds__(256)
k_pairdw_bwd_filter(const T* __restrict__ A, int lda, int64_t goff, const T* __restrict__ dG, int lddg, float* __restrict__ partial,
                    int H, int W, int C, int64_t pixels, int nchunk) {
    __shared__ float red[4 * 144];
    const int cbi = blockIdx.x, chunk = blockIdx.y, c0 = cbi * 8;
    const int64_t per = (pixels + nchunk - 1) / nchunk, p0 = (int64_t)chunk * per, p1 = min(pixels, p0 + per);
    float acc[18][8];
#pragma unroll
    for (int k = 0; k < 18; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    for (int64_t pix = p0 + threadIdx.x; pix < p1; pix += 256) {
        const int x = (int)(pix % W), y = (int)((pix / W) % H);
        float g[8];
        load8<T>(dG + pix * lddg + c0, g);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = y + ky - 1, xx = x + kx - 1;
                if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
                const int64_t q = pix + (int64_t)(ky - 1) * W + (kx - 1);
                float a1[8], a2[8];
                load8<T>(A + q * lda + c0, a1);
                load8<T>(A + goff + q * lda + c0, a2);
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc[ky * 3 + kx][j] += g[j] * a1[j]; acc[9 + ky * 3 + kx][j] += g[j] * a2[j]; }
            }
    }
    // block sum of the 144 values: xor-shuffles inside each wave, the four waves meet through LDS (18 rounds of a 256-wide LDS tree
    // with 10 barriers each cost more than the accumulation)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 18; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[k][j];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) red[wv * 144 + k * 8 + j] = v;
        }
    __syncthreads();
    if (threadIdx.x < 144)
        partial[((int64_t)chunk * gridDim.x + cbi) * 144 + threadIdx.x] =
            (red[threadIdx.x] + red[144 + threadIdx.x]) + (red[288 + threadIdx.x] + red[432 + threadIdx.x]);
}
// dw[c][d][t] (reference layout [C][2][3][3]) = sum over the chunks, in order
__global__ void k_pairdw_filter_finish(const float* __restrict__ partial, float* __restrict__ dw, int C, int nchunk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;                  // (c, k = d*9 + t)
    if (i >= C * 18) return;
    const int c = i / 18, k = i - c * 18, cb = c >> 3, j = c & 7, CB = C >> 3;
    float s = 0.f;
    for (int ch = 0; ch < nchunk; ++ch) s += partial[((int64_t)ch * CB + cb) * 144 + k * 8 + j];
    dw[i] = s;
}

// pixel chunks of the filter-gradient grid: ~4096 pixels (16 trips of a 256-thread block) each, so the 2 ... 16 channel blocks of a
// level still spread over the chip and the block-sum epilogue stays small beside the accumulation
static int pairdw_chunks(int64_t pixels) { return (int)std::max<int64_t>(1, std::min<int64_t>(1024, pixels / 4096)); }
int64_t pairdw_partial_floats(int B, int H, int W, int C) { return (int64_t)pairdw_chunks((int64_t)B * H * W) * (C / 8) * 144; }

void launch_pairdw_fwd(int dt, const void* A, int lda, int64_t goff, void* G, int ldg, const float* w, const float* b, int B, int H,
                       int W, int C, hipStream_t s) {
    const int64_t total = (int64_t)B * H * W * (C / 8);
    if (dt == BF16) k_pairdw_fwd<bf16><<<xc_cdiv(total, 256), 256, (size_t)C * 18 * 4, s>>>((const bf16*)A, lda, goff, (bf16*)G, ldg, w, b, H, W, C, total);
    else k_pairdw_fwd<float><<<xc_cdiv(total, 256), 256, (size_t)C * 18 * 4, s>>>((const float*)A, lda, goff, (float*)G, ldg, w, b, H, W, C, total);
}
void launch_pairdw_bwd_data(int dt, const void* dG, int lddg, void* dA, int ldda, int64_t goff, const float* w, int B, int H, int W,
                            int C, hipStream_t s) {
    const int64_t total = (int64_t)B * H * W * (C / 8);
    if (dt == BF16) k_pairdw_bwd_data<bf16><<<xc_cdiv(total, 256), 256, (size_t)C * 18 * 4, s>>>((const bf16*)dG, lddg, (bf16*)dA, ldda, goff, w, H, W, C, total);
    else k_pairdw_bwd_data<float><<<xc_cdiv(total, 256), 256, (size_t)C * 18 * 4, s>>>((const float*)dG, lddg, (float*)dA, ldda, goff, w, H, W, C, total);
}
void launch_pairdw_bwd_filter(int dt, const void* A, int lda, int64_t goff, const void* dG, int lddg, float* dw, float* partial, int B,
                              int H, int W, int C, hipStream_t s) {
    const int64_t pixels = (int64_t)B * H * W;
    const int nchunk = pairdw_chunks(pixels);
    dim3 grid(C / 8, nchunk);
    if (dt == BF16) k_pairdw_bwd_filter<bf16><<<grid, 256, 0, s>>>((const bf16*)A, lda, goff, (const bf16*)dG, lddg, partial, H, W, C, pixels, nchunk);
    else k_pairdw_bwd_filter<float><<<grid, 256, 0, s>>>((const float*)A, lda, goff, (const float*)dG, lddg, partial, H, W, C, pixels, nchunk);
    k_pairdw_filter_finish<<<xc_cdiv(C * 18, 256), 256, 0, s>>>(partial, dw, C, nchunk);
}

}  // namespace stcd
