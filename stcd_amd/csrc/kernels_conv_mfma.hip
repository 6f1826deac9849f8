// kernels_conv_mfma.hip -- bf16 implicit-GEMM "tap-list" convolution on the CDNA4 matrix cores (gfx950).
//
// One kernel family serves every conv-shaped op of the path (same geometry struct as the reference FMA kernel in
// kernels_conv_ref.hip, which is its on-device cross-check):
//   Conv2d 3x3 fwd / dgrad, ConvTranspose2d 3x3 (s=1) fwd / dgrad        SiamUnet_diff.py:18-48, 54-90
//   ConvTranspose2d 3x3 s=2 as 4 sub-pixel phases, its dgrad as a stride-2 conv   SiamUnet_diff.py:52
//
// GEMM view per block:  D[co][pos] = sum_k W[co][k] * X[pos][k],  k = (tap, ci)
//   * block = 256 threads = 4 waves, output tile 8 rows x 16 positions; wave w owns rows 2w, 2w+1
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand (rows = co) and the ACTIVATIONS as the B operand
//     (cols = positions): the accumulator then holds 4 consecutive channels of one position per lane, so the NHWC
//     store is 8 contiguous bytes per lane and bias / BN-statistics are per-register constants
//   * activations: an input halo tile (all taps of the output tile) is staged ONCE per channel chunk in LDS as
//     [pixel][CiB] bf16 with the 16-B chunk index XOR-swizzled by the pixel column (ds_read_b128 conflicts)
//   * weights: pre-packed on the device in MFMA-fragment order ([k-step][n-tile][lane][8]) so a wave's read of one
//     fragment is 1 KiB contiguous; streamed tap by tap through a double-buffered LDS image, the next tap's
//     global loads in flight in registers while the current tap's MFMAs run (one barrier per tap)
//   * small-Ci layers (Ci in {8,16,48}: first layer, 16-channel stages) use a flattened k = tap*Ci + c walk with the
//     whole weight image resident in LDS -- they are HBM-bound, the matrix cores just ride along
#include "common.h"

namespace stcd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct ConvMfmaArgs {
    stcd_conv_geom g;
    const bf16* in;
    const bf16* wf;        // fragment-order weights
    const float* bias;     // nullable
    void* out;
    int out_nchw;          // 1: fp32 NCHW [n][co][ho][wo]
    int CiB, nchunks, KS;  // channel chunk, chunks per Ci, k-steps per tap (mode A) / per chunk (mode B)
    int modeB;
    int NTtot;             // ceil(co/16)
    int dymin, dxmin, HH, HWp, swmask;
    int tiles_x, tiles_y;
    int halo_bytes, wbuf_bytes;
    int tap[9];            // (dy << 16) | (dx & 0xffff): 32-bit words, so a uniform tap index is a SCALAR load (see ConvGemmArgs)
};

// host: the packed tap table of a geometry
static inline void fill_taps(const stcd_conv_geom& g, int* tap) {
    for (int t = 0; t < 9; ++t) tap[t] = t < g.ntaps ? (int)(((unsigned)(int)g.dy[t] << 16) | ((unsigned)(int)g.dx[t] & 0xffffu)) : 0;
}
#define TAP_DY(w_) ((w_) >> 16)
#define TAP_DX(w_) ((int)(short)((w_) & 0xffff))

template <int NT>
__device__ __forceinline__ void conv_mfma_body(const ConvMfmaArgs& a, const int bx, const int cob) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int q = lane >> 4, r = lane & 15;
    int bid = bx;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int n = bid / a.tiles_y;
    const int my0 = ty * 8, mx0 = tx * 16;
    const int is = a.g.in_stride;
    const int CiB = a.CiB, nch8 = CiB >> 3, swmask = a.swmask, HWp = a.HWp;
    bf16* halo = reinterpret_cast<bf16*>(smem);
    bf16* wl = reinterpret_cast<bf16*>(smem + a.halo_bytes);
    int* tapoff = reinterpret_cast<int*>(smem + a.halo_bytes + 2 * a.wbuf_bytes);   // [9] pixel offset, [9] dx (mode B)
    if (tid < a.g.ntaps) {
        int tw = 0;
#pragma unroll
        for (int tt = 0; tt < 9; ++tt) tw = tid == tt ? a.tap[tt] : tw;      // scalar loads + selects (no per-lane byte loads)
        tapoff[tid] = (TAP_DY(tw) - a.dymin) * HWp + (TAP_DX(tw) - a.dxmin);
        tapoff[9 + tid] = TAP_DX(tw) - a.dxmin;
    }

    f32x4 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int WP = (2 * NT + 3) / 4;   // 16-B weight pieces per thread per tap (KS <= 2)
    const int w_pieces = a.KS * NT * 64;   // per tap (mode A)
    const int ntaps = a.g.ntaps;

    for (int cc = 0; cc < a.nchunks; ++cc) {
        __syncthreads();   // everyone is done reading the previous chunk's halo / weights
        // ---- stage the input halo tile of this channel chunk
        const int npieces = a.HH * HWp * nch8;
        for (int i = tid; i < npieces; i += 256) {
            const int ch = i % nch8, pix = i / nch8;
            const int hx = pix % HWp, hy = pix / HWp;
            const int gy = my0 * is + a.dymin + hy, gx = mx0 * is + a.dxmin + hx;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (gy >= 0 && gy < a.g.hi && gx >= 0 && gx < a.g.wi)
                v = *reinterpret_cast<const uint4*>(a.in + (((int64_t)n * a.g.hi + gy) * a.g.wi + gx) * a.g.ldi + cc * CiB + ch * 8);
            *reinterpret_cast<uint4*>(halo + (pix * nch8 + (ch ^ ((hx >> 1) & swmask))) * 8) = v;
        }
        if (a.modeB) {
            // ---- whole weight image of the (single) chunk: KS x NT fragments
            const bf16* wsrc = a.wf;
            for (int i = tid; i < a.KS * NT * 64; i += 256) {
                const int ks = i / (NT * 64), rem = i - ks * NT * 64;
                *reinterpret_cast<uint4*>(wl + (int64_t)i * 8) =
                    *reinterpret_cast<const uint4*>(wsrc + ((int64_t)(ks * a.NTtot + cob * NT) * 64 + rem) * 8);
            }
            __syncthreads();
            for (int ks = 0; ks < a.KS; ++ks) {
                const int kl = ks * 32 + 8 * q;
                int t = kl / CiB, c = kl - t * CiB;
                if (t >= ntaps) { t = 0; c = 0; }    // zero weights there; keep the address valid
                const int toff = tapoff[t], tdx = tapoff[9 + t];
                bf16x8 af[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int hx = r * is + tdx;
                    const int pix = (wid * 2 + m) * is * HWp + r * is + toff;
                    af[m] = *reinterpret_cast<const bf16x8*>(halo + (pix * nch8 + ((c >> 3) ^ ((hx >> 1) & swmask))) * 8);
                }
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2) {
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(wl + ((ks * NT + t2) * 64 + lane) * 8);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        acc[m][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, af[m], acc[m][t2], 0, 0, 0);
                }
            }
        } else {
            // ---- mode A: stream the weights tap by tap through two LDS buffers.  Software pipeline with ONE barrier
            //      per tap: regs(t) -> LDS buf[t&1]; barrier; issue global loads of tap t+1 into regs (in flight during
            //      the MFMAs); MFMAs of tap t.  buf[(t+1)&1] is rewritten only after every wave passed the next barrier,
            //      i.e. after all reads of tap t-1.
            // (named registers, not an array: hipcc parks a conditionally-written uint4 array in scratch)
            uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, w2 = w0, w3 = w0;
#define W_ADDR(P_, SRC_) ((SRC_) + ((int64_t)((min(tid + (P_) * 256, w_pieces - 1) / (NT * 64)) * a.NTtot + cob * NT) * 64 + \
                                   (min(tid + (P_) * 256, w_pieces - 1) % (NT * 64))) * 8)
#define W_LOAD(SRC_)                                                                      \
    do {                                                                                  \
        w0 = *reinterpret_cast<const uint4*>(W_ADDR(0, SRC_));                            \
        if constexpr (WP > 1) w1 = *reinterpret_cast<const uint4*>(W_ADDR(1, SRC_));      \
        if constexpr (WP > 2) w2 = *reinterpret_cast<const uint4*>(W_ADDR(2, SRC_));      \
        if constexpr (WP > 3) w3 = *reinterpret_cast<const uint4*>(W_ADDR(3, SRC_));      \
    } while (0)
#define W_STORE(DST_)                                                                                         \
    do {                                                                                                      \
        if (tid < w_pieces) *reinterpret_cast<uint4*>((DST_) + tid * 8) = w0;                                 \
        if constexpr (WP > 1) { if (tid + 256 < w_pieces) *reinterpret_cast<uint4*>((DST_) + (tid + 256) * 8) = w1; } \
        if constexpr (WP > 2) { if (tid + 512 < w_pieces) *reinterpret_cast<uint4*>((DST_) + (tid + 512) * 8) = w2; } \
        if constexpr (WP > 3) { if (tid + 768 < w_pieces) *reinterpret_cast<uint4*>((DST_) + (tid + 768) * 8) = w3; } \
    } while (0)
            {
                const bf16* wsrc = a.wf + ((int64_t)(cc * ntaps) * a.KS) * a.NTtot * 512;
                W_LOAD(wsrc);
            }
            for (int t = 0; t < ntaps; ++t) {
                bf16* wb = wl + (t & 1) * (a.wbuf_bytes >> 1);
                W_STORE(wb);
                __syncthreads();
                if (t + 1 < ntaps) {
                    const bf16* wsrc = a.wf + ((int64_t)(cc * ntaps + t + 1) * a.KS) * a.NTtot * 512;
                    W_LOAD(wsrc);
                }
                const int tw = a.tap[t];
                const int tdy = TAP_DY(tw) - a.dymin, tdx = TAP_DX(tw) - a.dxmin;
                for (int ks = 0; ks < a.KS; ++ks) {
                    bf16x8 af[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int hx = r * is + tdx;
                        const int pix = ((wid * 2 + m) * is + tdy) * HWp + hx;
                        af[m] = *reinterpret_cast<const bf16x8*>(halo + (pix * nch8 + ((ks * 4 + q) ^ ((hx >> 1) & swmask))) * 8);
                    }
#pragma unroll
                    for (int t2 = 0; t2 < NT; ++t2) {
                        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(wb + ((ks * NT + t2) * 64 + lane) * 8);
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            acc[m][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, af[m], acc[m][t2], 0, 0, 0);
                    }
                }
            }
#undef W_ADDR
#undef W_LOAD
#undef W_STORE
            __syncthreads();   // the last tap's buffer is free before the next chunk restages
        }
    }

    // ---- epilogue: lane (q, r) holds, for position r of row (wid*2+m), channels nt*16 + 4q + j
    const int mx = mx0 + r;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int my = my0 + wid * 2 + m;
        if (my >= a.g.hm || mx >= a.g.wm) continue;
        const int oy = my * a.g.out_stride + a.g.oy0, ox = mx * a.g.out_stride + a.g.ox0;
#pragma unroll
        for (int t2 = 0; t2 < NT; ++t2) {
            const int cb = (cob * NT + t2) * 16 + 4 * q;
            if (cb >= a.g.co) continue;
            float bvv[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) {
                if (cb + 3 < a.g.co) {
                    const float4 b4 = *reinterpret_cast<const float4*>(a.bias + cb);   // flat params are 16-B aligned
                    bvv[0] = b4.x; bvv[1] = b4.y; bvv[2] = b4.z; bvv[3] = b4.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (cb + j < a.g.co) bvv[j] = a.bias[cb + j];
                }
            }
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[m][t2][j] + bvv[j];
            if (a.out_nchw) {
                float* o = reinterpret_cast<float*>(a.out);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cb + j < a.g.co) o[(((int64_t)n * a.g.co + cb + j) * a.g.ho + oy) * a.g.wo + ox] = v[j];
            } else {
                bf16* o = reinterpret_cast<bf16*>(a.out) + (((int64_t)n * a.g.ho + oy) * a.g.wo + ox) * a.g.ldo + cb;
                if (cb + 3 < a.g.co) {
                    uint2 pk;
                    pk.x = pack_bf16x2(v[0], v[1]);
                    pk.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(o) = pk;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (cb + j < a.g.co) o[j] = (bf16)v[j];
                }
            }
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(256, 2)     // <= 256 registers: hipcc then selects the VGPR form of the MFMAs (see k_conv_small)
k_conv_mfma(const ConvMfmaArgs a) {
    conv_mfma_body<NT>(a, blockIdx.x, blockIdx.y);
}

// the 4 sub-pixel phases of a stride-2 transposed convolution in ONE launch (same input, disjoint outputs, same
// grid shape): blockIdx.z picks the phase's argument block from the kernarg segment
struct ConvMfmaArgs4 { ConvMfmaArgs a[4]; };
template <int NT>
__global__ void __launch_bounds__(256, 2)     // <= 256 registers: hipcc then selects the VGPR form of the MFMAs (see k_conv_small)
k_conv_mfma_x4(const ConvMfmaArgs4 j) {
    conv_mfma_body<NT>(j.a[blockIdx.z], blockIdx.x, blockIdx.y);
}
// ... or all four phases of a tile in ONE block, back to back (large maps: >= 1024 tiles).  The phases read the same input tile
// (one HBM fetch, three L2 hits instead of four fetches by blocks scattered over the XCDs) and write the interleaved halves /
// quarters of the same output lines within microseconds of each other on one XCD, so its L2 merges them into whole-line writes
// (a phase alone writes 32-B pieces at a 64-B stride on the 16-channel level).
template <int NT>
__global__ void __launch_bounds__(256, 2)     // <= 256 registers: hipcc then selects the VGPR form of the MFMAs (see k_conv_small)
k_conv_mfma_x4s(const ConvMfmaArgs4 j) {
#pragma unroll 1
    for (int z = 0; z < 4; ++z) {
        conv_mfma_body<NT>(j.a[z], blockIdx.x, blockIdx.y);
        __syncthreads();       // the next phase rewrites the tap table and the halo
    }
}

// ------------------------------------------------------------------ host side
static void taps_extent(const stcd_conv_geom& g, int* dymin, int* dymax, int* dxmin, int* dxmax) {
    *dymin = *dxmin = 127; *dymax = *dxmax = -127;
    for (int t = 0; t < g.ntaps; ++t) {
        *dymin = std::min<int>(*dymin, g.dy[t]); *dymax = std::max<int>(*dymax, g.dy[t]);
        *dxmin = std::min<int>(*dxmin, g.dx[t]); *dxmax = std::max<int>(*dxmax, g.dx[t]);
    }
}

ConvMfmaPlan conv_mfma_plan(const stcd_conv_geom& g) {
    ConvMfmaPlan p;
    const int Ci = g.ci;
    if (Ci % 64 == 0) p.CiB = 64;
    else if (Ci % 32 == 0) p.CiB = 32;
    else p.CiB = Ci;
    p.modeB = (p.CiB % 32) != 0;
    p.nchunks = Ci / p.CiB;
    p.KS = p.modeB ? (g.ntaps * Ci + 31) / 32 : p.CiB / 32;
    p.NTtot = (g.co + 15) / 16;
    p.NT = p.NTtot >= 8 ? 8 : p.NTtot >= 4 ? 4 : p.NTtot >= 2 ? 2 : 1;
    {   // small-spatial, wide layers: halve the co block so the grid covers the chip with >= 2 blocks per CU
        const int64_t tiles = (int64_t)g.n * ((g.hm + 7) / 8) * ((g.wm + 15) / 16);
        while (p.NT > 1 && tiles * ((p.NTtot + p.NT - 1) / p.NT) < 512) p.NT /= 2;
    }
    p.NTtot = ((p.NTtot + p.NT - 1) / p.NT) * p.NT;     // pad the n-tiles to whole blocks
    p.wf_elems = p.modeB ? (int64_t)p.KS * p.NTtot * 512 : (int64_t)p.nchunks * g.ntaps * p.KS * p.NTtot * 512;
    p.ok = (Ci % 8 == 0) && (p.modeB ? (Ci <= 64 && p.KS <= 24) : true) && g.ldi % 8 == 0;
    return p;
}

// fp32 [tap][kpad][wld] -> bf16 fragment order
__global__ void k_pack_frag(const float* __restrict__ w, int kpad, int wld, int ntaps, int Ci, int Co, int CiB, int nchunks,
                            int KS, int NTtot, int modeB, bf16* __restrict__ dst, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    int64_t f = i >> 9;                                // fragment index
    const int nt = (int)(f % NTtot); f /= NTtot;
    const int ks = (int)(f % KS); f /= KS;
    const int co = nt * 16 + (lane & 15);
    int t, ci;
    if (modeB) {
        const int kf = ks * 32 + 8 * (lane >> 4) + j;
        t = kf / Ci; ci = kf - t * Ci;
    } else {
        t = (int)(f % ntaps);
        const int cc = (int)(f / ntaps);
        ci = cc * CiB + ks * 32 + 8 * (lane >> 4) + j;
    }
    float v = 0.f;
    if (t < ntaps && ci < Ci && co < Co) v = w[((int64_t)t * kpad + ci) * wld + co];
    dst[i] = (bf16)v;
}

__device__ __forceinline__ float ref_weight(const PackSpec& ps, const float* src, int t, int k, int n) {
    if (n >= ps.N || k >= ps.K) return 0.f;
    const int64_t a_ = ps.kn_major ? ((int64_t)k * ps.N + n) : ((int64_t)n * ps.K + k);
    return src[(a_ * ps.ks + ps.ky[t]) * ps.ks + ps.kx[t]];
}

__global__ void __launch_bounds__(256)
k_pack_jobs(const PackJob* __restrict__ jobs, int njobs, int64_t total, const float* __restrict__ params, char* __restrict__ ws) {
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x, i = i0 + threadIdx.x;
    int lo = 0, hi = njobs - 1;            // last job with start <= i0: jobs start on 256-element boundaries, so the
    while (lo < hi) {                      // whole block shares it (uniform search: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].start <= i0) lo = mid; else hi = mid - 1;
    }
    const PackJob& jb = jobs[lo];
    const int64_t e = i - jb.start;
    if (e >= jb.count) return;
    const float* src = params + jb.src_off;
    if (jb.kind == 0) {
        const int n = (int)(e % jb.ps.wld);
        const int k = (int)((e / jb.ps.wld) % jb.ps.kpad);
        const int t = (int)(e / ((int64_t)jb.ps.wld * jb.ps.kpad));
        reinterpret_cast<float*>(ws + jb.dst_off)[e] = ref_weight(jb.ps, src, t, k, n);
    } else if (jb.kind == 2) {
        // 7x7 stride-2 stem as a 7-tap GEMM: tap = ky, "channel" k = kx*8 + c over a row of 8 pixels x 8 channels (kx == 7 and
        // c >= cin are zero); source layout [co][cin][7][7]
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
        int64_t f = e >> 9;
        const int nt = (int)(f % jb.NTtot); f /= jb.NTtot;
        const int ks = (int)(f % 2); f /= 2;
        const int ky = (int)f, co = nt * 16 + (lane & 15);
        const int k = ks * 32 + 8 * (lane >> 4) + j, kx = k >> 3, c = k & 7;
        float v = 0.f;
        if (ky < 7 && kx < 7 && c < jb.aux && co < jb.Co) v = src[(((int64_t)co * jb.aux + c) * 7 + ky) * 7 + kx];
        reinterpret_cast<bf16*>(ws + jb.dst_off)[e] = (bf16)v;
    } else if (!jb.modeB) {
        // fragment image [chunk][tap][ks][n-tile][lane][8]: one thread = one (ci, co) pair for ALL taps of the launch -- its taps
        // are adjacent in the reference tensor ([..][..][k][k]), so a wave reads whole lines (8 consecutive ci x k*k floats) and
        // writes one full 128-B segment per tap (thread-per-element read one float per line: ~0.8 TB/s on SegCD's 32.5 M weights)
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
        int64_t f = e >> 9;
        const int nt = (int)(f % jb.NTtot); f /= jb.NTtot;
        const int ks = (int)(f % jb.KS);
        const int cc = (int)(f / jb.KS);
        const int co = nt * 16 + (lane & 15), ci = cc * jb.CiB + ks * 32 + 8 * (lane >> 4) + j;
        const bool in = ci < jb.Ci && co < jb.Co;
        bf16* dst = reinterpret_cast<bf16*>(ws + jb.dst_off) + (((int64_t)cc * jb.ps.ntaps * jb.KS + ks) * jb.NTtot + nt) * 512 + (e & 511);
        const int64_t tstride = (int64_t)jb.KS * jb.NTtot * 512;
        for (int t = 0; t < jb.ps.ntaps; ++t) dst[t * tstride] = (bf16)(in ? ref_weight(jb.ps, src, t, ci, co) : 0.f);
    } else {
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
        int64_t f = e >> 9;
        const int nt = (int)(f % jb.NTtot); f /= jb.NTtot;
        const int ks = (int)(f % jb.KS); f /= jb.KS;
        const int co = nt * 16 + (lane & 15);
        int t, ci;
        if (jb.modeB) {
            const int kf = ks * 32 + 8 * (lane >> 4) + j;
            t = kf / jb.Ci; ci = kf - t * jb.Ci;
        } else {
            t = (int)(f % jb.ps.ntaps);
            ci = (int)(f / jb.ps.ntaps) * jb.CiB + ks * 32 + 8 * (lane >> 4) + j;
        }
        float v = 0.f;
        if (t < jb.ps.ntaps && ci < jb.Ci && co < jb.Co) v = ref_weight(jb.ps, src, t, ci, co);
        reinterpret_cast<bf16*>(ws + jb.dst_off)[e] = (bf16)v;
    }
}
void launch_pack_jobs(const PackJob* jobs_dev, int njobs, int64_t total, const float* params, char* ws, hipStream_t s) {
    if (njobs > 0 && total > 0) k_pack_jobs<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(jobs_dev, njobs, total, params, ws);
}

void launch_pack_frag(const stcd_conv_geom& g, const ConvMfmaPlan& p, const float* w, int kpad, int wld, void* dst,
                      hipStream_t s) {
    k_pack_frag<<<(unsigned)((p.wf_elems + 255) / 256), 256, 0, s>>>(w, kpad, wld, g.ntaps, g.ci, g.co, p.CiB, p.nchunks, p.KS,
                                                                      p.NTtot, p.modeB, (bf16*)dst, p.wf_elems);
}

static size_t conv_mfma_args(const stcd_conv_geom& g, const ConvMfmaPlan& p, const void* in, const void* wf, const float* bias,
                             void* out, bool out_nchw, ConvMfmaArgs& a) {
    a.g = g;
    a.in = (const bf16*)in; a.wf = (const bf16*)wf; a.bias = bias; a.out = out; a.out_nchw = out_nchw ? 1 : 0;
    a.CiB = p.CiB; a.nchunks = p.nchunks; a.KS = p.KS; a.modeB = p.modeB; a.NTtot = p.NTtot;
    int dymax, dxmax;
    taps_extent(g, &a.dymin, &dymax, &a.dxmin, &dxmax);
    a.HH = 7 * g.in_stride + (dymax - a.dymin) + 1;
    a.HWp = 15 * g.in_stride + (dxmax - a.dxmin) + 1;
    const int nch8 = p.CiB / 8;
    a.swmask = (nch8 & (nch8 - 1)) == 0 ? nch8 - 1 : 0;
    a.tiles_x = (g.wm + 15) / 16;
    a.tiles_y = (g.hm + 7) / 8;
    a.halo_bytes = (a.HH * a.HWp * p.CiB * 2 + 255) & ~255;
    a.wbuf_bytes = p.modeB ? ((p.KS * p.NT * 1024 + 1) / 2 + 255) & ~255 : (p.KS * p.NT * 1024 + 255) & ~255;
    fill_taps(g, a.tap);
    return (size_t)a.halo_bytes + 2 * (size_t)a.wbuf_bytes + 128;
}

int launch_conv_mfma(const stcd_conv_geom& g, const ConvMfmaPlan& p, const void* in, const void* wf, const float* bias,
                     void* out, bool out_nchw, hipStream_t s) {
    ConvMfmaArgs a;
    const size_t lds = conv_mfma_args(g, p, in, wf, bias, out, out_nchw, a);
    if (lds > 160 * 1024) return 1;
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * g.n), (unsigned)(p.NTtot / p.NT));
#define LAUNCH_NT(N_)                                                                                             \
    do {                                                                                                          \
        static bool attr_set = false;                                                                             \
        if (!attr_set) {                                                                                          \
            (void)hipFuncSetAttribute((const void*)k_conv_mfma<N_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                      \
        }                                                                                                         \
        k_conv_mfma<N_><<<grid, 256, lds, s>>>(a);                                                                \
    } while (0)
    switch (p.NT) {
        case 1: LAUNCH_NT(1); break;
        case 2: LAUNCH_NT(2); break;
        case 4: LAUNCH_NT(4); break;
        default: LAUNCH_NT(8); break;
    }
#undef LAUNCH_NT
    return 0;
}

// 0 = launched; 1 = the four phases do not share a launch shape (caller falls back to four launches)
int launch_conv_mfma_x4(const stcd_conv_geom g[4], const ConvMfmaPlan p[4], const void* in, const void* const wf[4],
                        const float* bias, void* out, hipStream_t s) {
    ConvMfmaArgs4 j;
    size_t lds = 0;
    for (int k = 0; k < 4; ++k) {
        if (!p[k].ok || p[k].NT != p[0].NT || p[k].NTtot != p[0].NTtot || g[k].n != g[0].n || g[k].hm != g[0].hm || g[k].wm != g[0].wm)
            return 1;
        lds = std::max(lds, conv_mfma_args(g[k], p[k], in, wf[k], bias, out, false, j.a[k]));
    }
    if (lds > 160 * 1024) return 1;
    const int64_t tiles = (int64_t)j.a[0].tiles_x * j.a[0].tiles_y * g[0].n * (p[0].NTtot / p[0].NT);
    // MEASURED SLOWER (round 4, SiamUnet_diff 16 x 256^2: the four launches 0.088 -> 0.133 ms per step; SNUNet 12.94 -> 13.20 ms): a
    // quarter of the blocks, each four times as long, fill the chip worse than the extra input fetches cost -- opt-in (STCD_X4_SEQ=1)
    static const int seq_env = [] { const char* e = getenv("STCD_X4_SEQ"); return e ? atoi(e) : 0; }();
    const bool seq = seq_env != 0 && tiles >= 1;
    dim3 grid((unsigned)(j.a[0].tiles_x * j.a[0].tiles_y * g[0].n), (unsigned)(p[0].NTtot / p[0].NT), seq ? 1 : 4);
#define LAUNCH_X4(N_)                                                                                             \
    do {                                                                                                          \
        static bool attr_set = false;                                                                             \
        if (!attr_set) {                                                                                          \
            (void)hipFuncSetAttribute((const void*)k_conv_mfma_x4<N_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            (void)hipFuncSetAttribute((const void*)k_conv_mfma_x4s<N_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                      \
        }                                                                                                         \
        if (seq) k_conv_mfma_x4s<N_><<<grid, 256, lds, s>>>(j);                                                   \
        else k_conv_mfma_x4<N_><<<grid, 256, lds, s>>>(j);                                                        \
    } while (0)
    switch (p[0].NT) {
        case 1: LAUNCH_X4(1); break;
        case 2: LAUNCH_X4(2); break;
        case 4: LAUNCH_X4(4); break;
        default: LAUNCH_X4(8); break;
    }
#undef LAUNCH_X4
    return 0;
}


// =====================================================================================================
// Weight gradient on the matrix cores:  dW[t][ci][co] = sum_pos X[pos*is + tap_t][ci] * dY[pos][co]
//
// GEMM per tap: M = ci, N = co, K = positions.  Both operands are stored position-major in NHWC, i.e. K is the slow
// axis of both -- exactly what ds_read_b64_tr_b16 is for: each 16-lane group reads a 4-position x 16-channel block and
// gets it channel-major, two reads make one 8-deep MFMA fragment.  The dY fragment is read once per k-step and reused
// by all 9 taps; the X fragment comes from the same halo tile shifted by the tap.
//   block  = 4 waves; WCI waves side by side over 16-channel ci tiles, the other 4/WCI waves split the k-steps
//            (position rows) of each tile; every wave keeps 9 x NTW accumulators for the whole block lifetime
//   grid.x = position blocks, each walks tiles t = bx, bx+gx, ... and finally writes ONE fp32 slab
//            [tap][ci][co] (plain coalesced stores, no global atomics: deterministic, and far cheaper than 1.3 TB/s
//            atomics); k_reduce_dw sums the slabs straight into the reference-layout gradient.

__device__ __forceinline__ bf16x8 tr_frag(const char* base0, const char* base1) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)base0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)base1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// LDS image of a [pixel][C] tile for ds_read_b64_tr_b16 (bank = (addr/4) % 64, conflicts inside a 32-lane half, whose
// two 16-lane groups read pixels x..x+3 and x+8..x+11 of one row): pixel x of a row sits at x*R + (x>>3)*PAD with
// R = 2*C bytes; PAD = 128 (C=16) / 32 (C=32) shifts the second group onto the banks the first one leaves free.
template <int C>
__device__ __forceinline__ int px_off(int x) { return x * (2 * C) + (x >> 3) * (C == 16 ? 128 : 32); }

// XF: X is a virtual activation (job.xf_*): a separate instantiation -- as a run-time branch the staging transform cost EVERY job of
// the kernel 20-26 more registers (<2,2> spilled 28 B / lane, <2,1> fell from 4 to 3 waves per SIMD: +40 % on the plain jobs' time)
template <int WCI, int NTW, bool T9, bool XF = false>
__device__ __forceinline__ void wgrad_body(const WgradJob& a, const char* base, const int bx, const int by, const int bz) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bf16* a_in = reinterpret_cast<const bf16*>(base + a.in_off);
    const bf16* a_dout = reinterpret_cast<const bf16*>(base + a.dout_off);
    float* a_slab = reinterpret_cast<float*>(const_cast<char*>(base) + a.slab_off);
    constexpr int WK = 4 / WCI, CIB = WCI * 16, COB = NTW * 16;
    constexpr int XCH = CIB / 8, YCH = COB / 8;
    // WCI == 4: a 64-channel X tile as TWO 32-channel sub-images (the padded layout is conflict-free up to 32 channels per
    // pixel); each wave then owns one 16-channel ci tile for ALL k-steps (WK == 1: no cross-wave exchange at the end)
    constexpr int SCX = CIB > 32 ? 32 : CIB, NSX = CIB / SCX;
    const int xsub = a.x_bytes / NSX;
    // NTW == 4 (64 output channels: the 64 x 64 tile of the wide layers): the dY tile likewise as two 32-channel sub-images
    constexpr int SCY = COB > 32 ? 32 : COB, NSY = COB / SCY;
    const int ysub = a.y_bytes / NSY;
    constexpr int YROW = 16 * 2 * SCY + 2 * (SCY == 16 ? 128 : 32);       // bytes of one 16-position dY row (of a sub-image)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wci = wid % WCI, wk = wid / WCI;
    const int grp = lane >> 4, li = lane & 15, qrow = li >> 2, pcol = li & 3;
    const int HWp = a.HWp;
    const int ci0 = by * CIB, co0 = bz * COB;
    const int buf_bytes = a.x_bytes + a.y_bytes;
    const int ntaps = T9 ? 9 : a.g.ntaps;

    f32x4 acc[9][NTW];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n_ = 0; n_ < NTW; ++n_) acc[t][n_] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging plan, computed once.  X halo: pieces i = tid + p*256 < nx (<= MAXX per thread); dY: exactly NTW
    //      pieces per thread (128 positions x YCH chunks = 256*NTW).  Every load below is unconditional (an
    //      out-of-range piece reads the tile origin and is zeroed by a select), so the loop body has no divergent
    //      branches and all loads of a tile are in flight together.
    const int nx = a.HH * HWp * XCH;
    constexpr int MAXX = WCI == 4 ? 6 : 3;
    int xg[MAXX], xl[MAXX], xa[MAXX], xb[MAXX];       // global element offset, LDS byte offset, bound coords
#pragma unroll
    for (int p = 0; p < MAXX; ++p) {
        const int i = min(tid + p * 256, nx - 1);
        const int ch = i % XCH, pix = i / XCH, hx = pix % HWp, hy = pix / HWp;
        xa[p] = hy + a.dymin; xb[p] = hx + a.dxmin;
        xg[p] = ((hy * a.g.wi + hx) * a.g.ldi + ci0 + ch * 8) * 2;  // bytes past the tile's halo corner
        xl[p] = (ch / (SCX / 8)) * xsub + hy * a.xrow_bytes + px_off<SCX>(hx) + (ch % (SCX / 8)) * 16;
        if (ci0 + ch * 8 >= a.g.ci) xa[p] = 1 << 28;                 // channels beyond Ci: zero
    }
    int yg[NTW], yl[NTW], ya[NTW], yb[NTW];
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
        const int j = tid + q * 256, ch = j % YCH, pix = j / YCH, py = pix >> 4, pxx = pix & 15;
        ya[q] = py; yb[q] = pxx;
        yg[q] = (((py * a.g.out_stride) * a.g.wo + pxx * a.g.out_stride) * a.g.ldo + co0 + ch * 8) * 2;   // bytes
        yl[q] = a.x_bytes + (ch / (SCY / 8)) * ysub + py * YROW + px_off<SCY>(pxx) + (ch % (SCY / 8)) * 16;
        if (co0 + ch * 8 >= a.co_valid) ya[q] = 1 << 28;
    }
    const int npx = (nx + 255) >> 8;                  // X pieces in use (uniform)

    // tile walk (incremental)
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int step = a.gx;
    const int dn = step / tiles_img, drem = step - dn * tiles_img, dty = drem / a.tiles_x, dtx = drem - dty * a.tiles_x;
    int tile = bx;
    int tn = tile / tiles_img, trem = tile - tn * tiles_img, tty = trem / a.tiles_x, ttx = trem - tty * a.tiles_x;

    // Raw buffer loads: the X descriptor starts `lead` bytes before the tensor so offsets relative to a tile's halo
    // corner are never negative; a piece outside the image / beyond the channels gets an offset past num_records and the
    // hardware returns zeros (no selects, no 64-bit address arithmetic per piece, nothing between load and use).
    const int64_t lead = ((int64_t)(-a.dymin) * a.g.wi + (-a.dxmin)) * a.g.ldi * 2;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a_in)) - lead, (short)0, (int)(a.in_bytes + (unsigned)lead), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a_dout)), (short)0, (int)a.dout_bytes, 0x00020000);
    uint4 prex[MAXX], prey[NTW];
    // X as a virtual activation (job.xf_C > 0; uniform per block): scale / shift table of the producer in LDS behind the two staging
    // buffers, the fetched tile's validity bits, image group and Dropout2d factors ride with the loads, xf_act8 at the LDS store
    constexpr bool xf_on = XF;
    float* const xf_tab = reinterpret_cast<float*>(smem + 2 * buf_bytes);
    const int xf_ch = min(ci0 + (tid % XCH) * 8, max(a.xf_C - 8, 0));        // (tid + p * 256) % XCH is the same for every piece p
    unsigned xf_ok = 0; int xf_g = 0;
    float4 xf_m0 = make_float4(1.f, 1.f, 1.f, 1.f), xf_m1 = xf_m0;
    if constexpr (xf_on) {
        const float* st = reinterpret_cast<const float*>(base + a.xf_stat_off);
        for (int i = tid; i < a.xf_groups * 2 * a.xf_C; i += 256) {
            const int c = i % a.xf_C, w = (i / a.xf_C) & 1, g_ = i / (2 * a.xf_C);
            xf_tab[i] = st[((int64_t)g_ * 4 + 2 + w) * a.xf_C + c];
        }
        __syncthreads();
    }
#define WG_FETCH(N_, Y_, X_)                                                                                          \
    do {                                                                                                              \
        const int my0_ = (Y_) * 8, mx0_ = (X_) * 16;                                                                   \
        if constexpr (xf_on) {                                                                                        \
            xf_ok = 0; xf_g = (N_) / a.xf_npg;                                                                        \
            if (a.xf_mask_off >= 0) {                                                                                 \
                const float4* mp_ = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base + a.xf_mask_off) + (int64_t)(N_) * a.xf_C + xf_ch); \
                xf_m0 = mp_[0]; xf_m1 = mp_[1];                                                                       \
            }                                                                                                         \
        }                                                                                                             \
        const unsigned xs_ = (unsigned)(((((int64_t)(N_) * a.g.hi + my0_) * a.g.wi + mx0_) * a.g.ldi) * 2);            \
        const unsigned ys_ = (unsigned)(((((int64_t)(N_) * a.g.ho + my0_ * a.g.out_stride + a.g.oy0) * a.g.wo +        \
                                          mx0_ * a.g.out_stride + a.g.ox0) * a.g.ldo) * 2);                            \
        _Pragma("unroll") for (int p = 0; p < MAXX; ++p) {                                                            \
            if (p < npx) {                                                                                            \
                const bool ok_ = (unsigned)(my0_ + xa[p]) < (unsigned)a.g.hi && (unsigned)(mx0_ + xb[p]) < (unsigned)a.wi_valid; \
                const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok_ ? (unsigned)xg[p] : 0x80000000u, xs_, 0); \
                prex[p] = make_uint4(v_[0], v_[1], v_[2], v_[3]);                                                     \
                if constexpr (xf_on) xf_ok |= ok_ ? (1u << p) : 0u;                                                   \
            }                                                                                                         \
        }                                                                                                             \
        _Pragma("unroll") for (int q = 0; q < NTW; ++q) {                                                             \
            const bool ok_ = (unsigned)(my0_ + ya[q]) < (unsigned)a.g.hm && mx0_ + yb[q] < a.g.wm;                     \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(yrs, ok_ ? (unsigned)yg[q] : 0x80000000u, ys_, 0); \
            prey[q] = make_uint4(v_[0], v_[1], v_[2], v_[3]);                                                         \
        }                                                                                                             \
    } while (0)
#define WG_STASH(BUF_)                                                                                                \
    do {                                                                                                              \
        if constexpr (xf_on) {                                                                                        \
            float sc_[8], sh_[8];                                                                                     \
            const float* tb_ = xf_tab + (xf_g * 2) * a.xf_C + xf_ch;                                                  \
            xf_fold8(tb_, tb_ + a.xf_C, xf_m0, xf_m1, sc_, sh_);                                                      \
            _Pragma("unroll") for (int p = 0; p < MAXX; ++p)                                                          \
                if (p < npx) prex[p] = xf_act8(prex[p], sc_, sh_, 0u - ((xf_ok >> p) & 1u));                          \
        }                                                                                                             \
        _Pragma("unroll") for (int p = 0; p < MAXX; ++p)                                                              \
            if (p < npx && tid + p * 256 < nx) *reinterpret_cast<uint4*>(smem + (BUF_) * buf_bytes + xl[p]) = prex[p]; \
        _Pragma("unroll") for (int q = 0; q < NTW; ++q)                                                               \
            *reinterpret_cast<uint4*>(smem + (BUF_) * buf_bytes + yl[q]) = prey[q];                                   \
    } while (0)

    // per-lane fragment geometry inside a k-step (32 positions = 2 tile rows): half h -> x = 8*(grp&1) + 4*h + qrow
    const int frow = grp >> 1;
    const int fx0 = 8 * (grp & 1) + qrow, fx1 = fx0 + 4;
    const int yb0 = frow * YROW + px_off<SCY>(fx0) + 8 * pcol, yb1 = frow * YROW + px_off<SCY>(fx1) + 8 * pcol;
    int xoff0[9], xoff1[9];      // X fragment byte offsets per tap (relative to the k-step's first halo row)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tdy = t < ntaps ? a.g.dy[t] - a.dymin : 0, tdx = t < ntaps ? a.g.dx[t] - a.dxmin : 0;
        const int csub = (wci * 16 / SCX) * xsub + (wci * 16 % SCX) * 2;      // this wave's ci tile: sub-image + byte offset inside a pixel
        xoff0[t] = (tdy + frow) * a.xrow_bytes + px_off<SCX>(fx0 + tdx) + csub + 8 * pcol;
        xoff1[t] = (tdy + frow) * a.xrow_bytes + px_off<SCX>(fx1 + tdx) + csub + 8 * pcol;
    }

    int buf = 0;
    if (tile < a.ntiles) { WG_FETCH(tn, tty, ttx); WG_STASH(0); }
    __syncthreads();
    for (; tile < a.ntiles; tile += step, buf ^= 1) {
        int nn = tn, nyy = tty, nxx = ttx;
        nxx += dtx; if (nxx >= a.tiles_x) { nxx -= a.tiles_x; ++nyy; }
        nyy += dty; if (nyy >= a.tiles_y) { nyy -= a.tiles_y; ++nn; }
        nn += dn;
        const bool have_next = tile + step < a.ntiles;
        if (have_next) WG_FETCH(nn, nyy, nxx);
        const char* xs = smem + buf * buf_bytes;
        const char* ys = xs + a.x_bytes;
#pragma unroll
        for (int ks = wk; ks < 4; ks += WK) {
            bf16x8 bfr[NTW];
#pragma unroll
            for (int n_ = 0; n_ < NTW; ++n_) {
                const int yo = ((n_ * 16) / SCY) * ysub + ((n_ * 16) % SCY) * 2;      // sub-image + byte offset of the n-tile inside a pixel
                bfr[n_] = tr_frag(ys + ks * 2 * YROW + yb0 + yo, ys + ks * 2 * YROW + yb1 + yo);
            }
            const char* xr = xs + ks * 2 * a.xrow_bytes;
            if constexpr (T9) {
                bf16x8 afr[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) afr[t] = tr_frag(xr + xoff0[t], xr + xoff1[t]);
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int n_ = 0; n_ < NTW; ++n_)
                        acc[t][n_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[t], bfr[n_], acc[t][n_], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (t < ntaps) {
                        const bf16x8 afr = tr_frag(xr + xoff0[t], xr + xoff1[t]);
#pragma unroll
                        for (int n_ = 0; n_ < NTW; ++n_)
                            acc[t][n_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[n_], acc[t][n_], 0, 0, 0);
                    }
                }
            }
        }
        if (have_next) WG_STASH(buf ^ 1);
        tn = nn; tty = nyy; ttx = nxx;
        __syncthreads();
    }
#undef WG_FETCH
#undef WG_STASH

    // ---- block result -> slab[bx][t][ci][co]; lane holds D[ci_local = 4*grp + j][co_local = li].
    //      The WK waves that split the k-steps meet through a lane-contiguous LDS exchange (no LDS float atomics: they
    //      serialise -- 72 ds_add_f32 per lane cost ~30 us per block); wave wk == 0 of each ci tile then owns the sum.
    float* red = reinterpret_cast<float*>(smem);
    constexpr int PERW = 9 * NTW * 4 * 64;            // floats one wave parks
    if (wk > 0) {
        float* dst = red + ((wk - 1) * WCI + wci) * PERW + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int n_ = 0; n_ < NTW; ++n_)
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[((t * NTW + n_) * 4 + j) * 64] = acc[t][n_][j];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll 1
        for (int w = 1; w < WK; ++w) {
            const float* src = red + ((w - 1) * WCI + wci) * PERW + lane;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int n_ = 0; n_ < NTW; ++n_)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][n_][j] += src[((t * NTW + n_) * 4 + j) * 64];
        }
        float* slab = a_slab + (int64_t)bx * ntaps * a.kpad * a.wld;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (t < ntaps)
#pragma unroll
                for (int n_ = 0; n_ < NTW; ++n_)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ci = ci0 + wci * 16 + 4 * grp + j, co = co0 + n_ * 16 + li;
                        if (ci < a.kpad && co < a.wld) slab[((int64_t)t * a.kpad + ci) * a.wld + co] = acc[t][n_][j];
                    }
    }
}

template <int WCI, int NTW, bool T9>
__global__ void __launch_bounds__(256, NTW == 4 ? 1 : 2)
k_wgrad_mfma(const WgradJob a) {
    wgrad_body<WCI, NTW, T9>(a, nullptr, blockIdx.x, blockIdx.y, blockIdx.z);
}

template <int WCI, int NTW, bool T9, bool XF = false>
// <4,2>: 72 accumulators + 6 X pieces in flight: 2 blocks per CU; <4,4>: 144 accumulators: one wave per SIMD
__global__ void __launch_bounds__(256, (WCI == 2 && NTW == 2 && !XF) ? 3 : NTW == 4 ? 1 : 2)
k_wgrad_group(const WgradJob* __restrict__ jobs, int njobs, const char* base) {
    int lo = 0, hi = njobs - 1;            // last job with start <= blockIdx.x (uniform: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const WgradJob& a = jobs[lo];
    const int lb = blockIdx.x - a.start;
    // XCD-aware order: the gy*gz blocks that walk the SAME tile sequence (K slice bx) over different (ci, co) groups re-read
    // the same X / dY tiles; workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share one), so those blocks
    // get ids 8 apart: they run side by side on one XCD and the re-reads hit its L2 instead of crossing the fabric.
    const int G = a.gy * a.gz, full = (a.gx >> 3) << 3;
    int bx, r;
    if (lb < full * G) { const int rem = lb % (8 * G); bx = (lb / (8 * G)) * 8 + (rem & 7); r = rem >> 3; }
    else { const int l2 = lb - full * G, tail = a.gx - full; bx = full + l2 % tail; r = l2 / tail; }
    wgrad_body<WCI, NTW, T9, XF>(a, base, bx, r % a.gy, r / a.gy);
}

// sum the slabs; out[t][k][n] (engine layout, same as the slab) or, with a PackSpec, straight into the reference layout
// PARTS threads cooperate on one output (256/PARTS outputs per block): many slabs x few outputs (16-channel layers)
// use PARTS = 64 so the chip still sees enough blocks and each thread's chain of dependent loads stays short.
template <int PARTS>
__global__ void __launch_bounds__(256)
k_reduce_dw(const float* __restrict__ slab, int gx, int64_t slab_stride, int ntaps, int K, int N, int kpad, int wld,
            PackSpec ps, int use_ps, float* __restrict__ out) {
    constexpr int OUTS = 256 / PARTS;
    __shared__ float red[256];
    const int o = blockIdx.x * OUTS + (threadIdx.x % OUTS), part = threadIdx.x / OUTS;
    const int total = ntaps * K * N;
    float acc = 0.f;
    int n = 0, k = 0, t = 0;
    if (o < total) {
        n = o % N; k = (o / N) % K; t = o / (N * K);
        const float* p = slab + ((int64_t)t * kpad + k) * wld + n;
        for (int b = part; b < gx; b += PARTS) acc += p[(int64_t)b * slab_stride];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (part == 0 && o < total) {
        acc = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) acc += red[q * OUTS + threadIdx.x];
        if (use_ps) {
            const int64_t a_ = ps.kn_major ? ((int64_t)k * ps.N + n) : ((int64_t)n * ps.K + k);
            out[(a_ * ps.ks + ps.ky[t]) * ps.ks + ps.kx[t]] = acc;
        } else {
            out[((int64_t)t * kpad + k) * wld + n] = acc;
        }
    }
}

// Block = (tile of 32 output channels x TK input channels x all taps of the launch): the slabs are read as rows of 32
// consecutive floats (the slab's fast axis is the output channel), summed, passed through LDS and written in the order of the
// REFERENCE-layout gradient tensor ([Cout][Cin][k][k]: the TK x k x k values of one output channel are contiguous; transposed
// convs [Cin][Cout][k][k] likewise with the roles swapped).  The thread-per-output form wrote one float per 64-B line (0.45 ms
// per SegCD step for 32.5 M weights).
// DETERMINISTIC (round 4): every output is the sum of its slabs in ONE fixed order whatever the block schedule.  Small filters:
// `parts` threads OF ONE BLOCK each sum a contiguous range of slabs (4 independent chains, fixed association) and thread 0 of the
// output adds the parts in index order through LDS; tiled filters: one block walks all slabs of its tile.  Rounds 1-3 cut
// many-slab jobs into parts of other blocks that met through float atomicAdd: two runs of one step differed by ~1e-6.
__global__ void __launch_bounds__(256)
k_reduce_jobs(const ReduceJob* __restrict__ jobs, int njobs, int64_t total, const char* __restrict__ ws, float* __restrict__ grads) {
    __shared__ float tile[9 * 16 * 33];            // [tap][k][33] (ntaps > 1: TK = 16) or [32][33]; small filters: [parts][outs] partials
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x;
    int lo = 0, hi = njobs - 1;            // block-uniform job (starts are 256-aligned)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].start <= i0) lo = mid; else hi = mid - 1;
    }
    const ReduceJob& jb = jobs[lo];
    if (!jb.tiled) {       // small filters (L2-resident gradient tensor): (output, part) threads spread a few thousand sums best
        const int parts = jb.parts, OUTS = 256 / parts;                     // parts: power of two <= 64 (planner)
        const int lb = (int)((i0 - jb.start) >> 8);
        const int ol = threadIdx.x % OUTS, part = threadIdx.x / OUTS;
        const int outs = jb.ntaps * jb.K * jb.N;
        const int o = lb * OUTS + ol;
        const int chunk = (jb.gx + parts - 1) / parts;
        const int b0 = part * chunk, b1 = min(jb.gx, b0 + chunk);
        int n = 0, k = 0, t = 0;
        float acc = 0.f;
        if (o < outs) {
            n = o % jb.N; k = (o / jb.N) % jb.K; t = o / (jb.N * jb.K);
            const float* p = reinterpret_cast<const float*>(ws + jb.slab_off) + ((int64_t)t * jb.kpad + k) * jb.wld + n;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int b = b0;
            for (; b + 3 < b1; b += 4) {
                a0 += p[(int64_t)b * jb.slab_stride];
                a1 += p[(int64_t)(b + 1) * jb.slab_stride];
                a2 += p[(int64_t)(b + 2) * jb.slab_stride];
                a3 += p[(int64_t)(b + 3) * jb.slab_stride];
            }
            for (; b < b1; ++b) a0 += p[(int64_t)b * jb.slab_stride];
            acc = (a0 + a1) + (a2 + a3);
        }
        if (parts > 1) {
            tile[threadIdx.x] = acc;
            __syncthreads();
            if (part == 0) {
                acc = tile[ol];
                for (int q = 1; q < parts; ++q) acc += tile[q * OUTS + ol];      // fixed order
            }
        }
        if (part == 0 && o < outs) {
            const int64_t a_ = jb.kn_major ? ((int64_t)k * jb.N + n) : ((int64_t)n * jb.K + k);
            grads[jb.out_off + (a_ * jb.ks + jb.ky[t]) * jb.ks + jb.kx[t]] = acc;
        }
        return;
    }
    const int TK = jb.ntaps == 1 ? 32 : 16;
    const int tiles_n = (jb.N + 31) >> 5, tiles_k = (jb.K + TK - 1) / TK, ntiles = tiles_n * tiles_k;
    const int lb = (int)((i0 - jb.start) >> 8);
    if (lb >= ntiles) return;
    const int tk = lb / tiles_n, tn = lb - tk * tiles_n;
    const int TE = jb.ntaps * TK * 32;
    const float* slab = reinterpret_cast<const float*>(ws + jb.slab_off);
    for (int idx = threadIdx.x; idx < TE; idx += 256) {
        const int nl = idx & 31, kl = (idx >> 5) % TK, t = idx / (32 * TK);
        const int n = tn * 32 + nl, k = tk * TK + kl;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (n < jb.N && k < jb.K) {
            const float* p = slab + ((int64_t)t * jb.kpad + k) * jb.wld + n;
            int b = 0;
            for (; b + 3 < jb.gx; b += 4) {
                a0 += p[(int64_t)b * jb.slab_stride];
                a1 += p[(int64_t)(b + 1) * jb.slab_stride];
                a2 += p[(int64_t)(b + 2) * jb.slab_stride];
                a3 += p[(int64_t)(b + 3) * jb.slab_stride];
            }
            for (; b < jb.gx; ++b) a0 += p[(int64_t)b * jb.slab_stride];
        }
        tile[(t * TK + kl) * 33 + nl] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TE; idx += 256) {
        const int t = idx % jb.ntaps;
        int nl, kl;
        if (jb.kn_major) { nl = (idx / jb.ntaps) & 31; kl = idx / (jb.ntaps * 32); }      // [k][n][taps]
        else { kl = (idx / jb.ntaps) % TK; nl = idx / (jb.ntaps * TK); }                   // [n][k][taps]
        const int n = tn * 32 + nl, k = tk * TK + kl;
        if (n >= jb.N || k >= jb.K) continue;
        const int64_t a_ = jb.kn_major ? ((int64_t)k * jb.N + n) : ((int64_t)n * jb.K + k);
        grads[jb.out_off + (a_ * jb.ks + jb.ky[t]) * jb.ks + jb.kx[t]] = tile[(t * TK + kl) * 33 + nl];
    }
}
void launch_reduce_jobs(const ReduceJob* jobs_dev, int njobs, int64_t total, const char* ws, float* grads, hipStream_t s) {
    if (njobs > 0 && total > 0) k_reduce_jobs<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(jobs_dev, njobs, total, ws, grads);
}

WgradMfmaPlan wgrad_mfma_plan(const stcd_conv_geom& g, int kpad, int wld, bool allow_wide, int force_co64) {
    WgradMfmaPlan p;
    int dymin, dymax, dxmin, dxmax;
    taps_extent(g, &dymin, &dymax, &dxmin, &dxmax);
    const int HH = 7 * g.in_stride + (dymax - dymin) + 1, HWp = 15 * g.in_stride + (dxmax - dxmin) + 1;
    // block tile: 16 / 32 / 64 input channels x 16 / 32 output channels.  <= 32 x 32: 72 accumulator registers, >= 3 waves per
    // SIMD, the k-steps of a tile split over the waves; 64 x 32 (wide layers): every wave owns one 16-channel ci tile for all
    // k-steps, dY is re-read once per 64 input channels instead of 32, twice the MFMAs per staged tile.
    static const int wide_env = [] { const char* e = getenv("STCD_WGRAD_CI64"); return e ? atoi(e) : -1; }();     // -1: the caller decides
    const bool wide_ok = wide_env < 0 ? allow_wide : wide_env != 0;
    const int tiles_x = (g.wm + 15) / 16, tiles_y = (g.hm + 7) / 8;
    const int64_t ntiles = (int64_t)g.n * tiles_x * tiles_y;
    // STCD_WGRAD_CO64=1 (default off): 64 x 64-channel tile for layers with >= 64 channels on both sides -- 26 transposed LDS reads per
    // 36 MFMAs instead of 22 per 18, X / dY re-read once per 64 instead of 32 channels.  Built, parity-tested (tests/test_ops_gpu.py
    // runs it through impl 4) and MEASURED SLOWER: its 144 accumulators leave one wave per SIMD and one block per CU (102 KB of LDS),
    // so nothing hides the staging round trips -- ChangeFormer's 3x3 group 6.05 ms vs 5.40 (32 x 32 tiles), SNUNet's 2.61 vs 1.86
    // (64 x 32).  Kept as an opt-in variant.
    static const int co64_env = [] { const char* e = getenv("STCD_WGRAD_CO64"); return e ? atoi(e) : 0; }();
    const bool co64 = (force_co64 >= 0 ? force_co64 != 0 : co64_env != 0) && wide_ok && g.ci >= 64 && g.co >= 64 && HH * HWp * 8 <= 6 * 256;
    p.NTW = co64 ? 4 : g.co >= 32 ? 2 : 1;
    // (allow_wide is the ENGINE's per-family choice: all qualifying layers of a stage must move together -- two grouped launches
    //  overlap worse than one.  Measured: SNUNet (hundreds of concatenated input channels on 128^2 / 256^2 maps) 2.50 -> 2.12 ms;
    //  SegCD +0.11 ms, SiamUnet_diff +0.07 ms: on their small deep maps halving the (ci, co) groups costs more parallelism and
    //  slab traffic than the wider tile saves.)
    p.WCI = (co64 || (wide_ok && g.ci >= 64 && p.NTW == 2 && HH * HWp * 8 <= 6 * 256)) ? 4 : g.ci >= 32 ? 2 : 1;
    p.gy = (g.ci + p.WCI * 16 - 1) / (p.WCI * 16);
    p.gz = (g.co + p.NTW * 16 - 1) / (p.NTW * 16);
    const int64_t slab_bytes = (int64_t)g.ntaps * kpad * wld * 4;
    static const int target_blocks = [] { const char* e = getenv("STCD_WGRAD_BLOCKS"); return e ? atoi(e) : 1536; }();
    int64_t gx = std::max<int64_t>(1, target_blocks / (p.gy * p.gz));
    gx = std::min<int64_t>(gx, std::max<int64_t>(1, ((int64_t)24 << 20) / slab_bytes));
    gx = std::min<int64_t>(gx, ntiles);
    p.gx = (int)gx;
    p.slab_floats = (int64_t)p.gx * g.ntaps * kpad * wld;
    const int xpieces = HH * HWp * (p.WCI * 2);
    p.ok = g.ci % 8 == 0 && g.ldi % 8 == 0 && g.ldo % 8 == 0 && g.in_stride == 1 && xpieces <= (p.WCI == 4 ? 6 : 3) * 256 && g.hm <= g.hi && g.wm <= g.wi &&
           ((int64_t)g.n * g.hi + 4) * g.wi * g.ldi * 2 < ((int64_t)1 << 31) && (int64_t)g.n * g.ho * g.wo * g.ldo * 2 < ((int64_t)1 << 31);
    return p;
}

WgradJob wgrad_make_job(const stcd_conv_geom& g, const WgradMfmaPlan& p, int64_t in_off, int64_t dout_off, int64_t slab_off,
                        int kpad, int wld) {
    WgradJob a;
    memset(&a, 0, sizeof(a));
    a.g = g;
    a.in_off = in_off; a.dout_off = dout_off; a.slab_off = slab_off; a.kpad = kpad; a.wld = wld;
    int dymax, dxmax;
    taps_extent(g, &a.dymin, &dymax, &a.dxmin, &dxmax);
    a.HH = 7 * g.in_stride + (dymax - a.dymin) + 1;
    a.HWp = 15 * g.in_stride + (dxmax - a.dxmin) + 1;
    a.tiles_x = (g.wm + 15) / 16;
    a.tiles_y = (g.hm + 7) / 8;
    a.ntiles = g.n * a.tiles_x * a.tiles_y;
    const int CIB = p.WCI * 16, COB = p.NTW * 16;
    const int SCX = CIB > 32 ? 32 : CIB, NSX = CIB / SCX;          // X tile as NSX sub-images of SCX channels (wgrad_body)
    const int SCY = COB > 32 ? 32 : COB, NSY = COB / SCY;          // dY tile likewise (the 64 x 64 tile)
    const int xpad = SCX == 16 ? 128 : 32, ypad = SCY == 16 ? 128 : 32;
    a.xrow_bytes = (a.HWp * 2 * SCX + ((a.HWp + 7) / 8) * xpad + 15) & ~15;
    a.x_bytes = NSX * ((a.HH * a.xrow_bytes + 255) & ~255);
    a.y_bytes = NSY * ((8 * (16 * 2 * SCY + 2 * ypad) + 255) & ~255);
    a.co_valid = (g.co + 7) & ~7;
    a.in_bytes = (unsigned)((int64_t)g.n * g.hi * g.wi * g.ldi * 2);
    a.dout_bytes = (unsigned)((int64_t)g.n * g.ho * g.wo * g.ldo * 2);
    a.gx = p.gx; a.gy = p.gy; a.gz = p.gz; a.start = 0;
    a.wi_valid = p.wi_valid > 0 ? p.wi_valid : g.wi; a.pad_ = 0;
    const int WK = 4 / p.WCI;
    a.lds_bytes = (int)std::max<size_t>(2 * (size_t)(a.x_bytes + a.y_bytes), (size_t)(WK - 1) * p.WCI * 9 * p.NTW * 4 * 64 * 4);
    a.xf_stat_off = -1; a.xf_mask_off = -1; a.xf_C = 0; a.xf_groups = 1; a.xf_npg = 1; a.pad2_ = 0;     // plain X (wgrad_job_set_xf)
    return a;
}

// X of this job is the producer's raw conv output (virtual activation): its published table, masks and BatchNorm groups
void wgrad_job_set_xf(WgradJob& a, int64_t stat_off, int64_t mask_off, int C, int groups, int npg) {
    a.xf_stat_off = stat_off; a.xf_mask_off = mask_off; a.xf_C = C; a.xf_groups = groups; a.xf_npg = npg;
    a.lds_bytes = std::max(a.lds_bytes, 2 * (a.x_bytes + a.y_bytes) + groups * 2 * C * 4);
}

#define WG_DISPATCH(W_, N_, T_, WHAT)                                       \
    do {                                                                    \
        if ((W_) == 4 && (N_) == 4) { if (T_) { WHAT(4, 4, true); } else { WHAT(4, 4, false); } }   \
        else if ((W_) == 4) { if (T_) { WHAT(4, 2, true); } else { WHAT(4, 2, false); } }           \
        else if ((W_) == 1 && (N_) == 1) { if (T_) { WHAT(1, 1, true); } else { WHAT(1, 1, false); } }   \
        else if ((W_) == 1) { if (T_) { WHAT(1, 2, true); } else { WHAT(1, 2, false); } }           \
        else if ((N_) == 1) { if (T_) { WHAT(2, 1, true); } else { WHAT(2, 1, false); } }           \
        else { if (T_) { WHAT(2, 2, true); } else { WHAT(2, 2, false); } }                          \
    } while (0)

int wgrad_variant_slots(int WCI, int NTW, bool t9, int lds_bytes) {
    int dev = 0, cus = 256, per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 512; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
#define WG_OCC(W_, N_, T_) \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wgrad_group<W_, N_, T_>, 256, (size_t)lds_bytes) != hipSuccess) per_cu = 2
    WG_DISPATCH(WCI, NTW, t9, WG_OCC);
#undef WG_OCC
    (void)hipGetLastError();
    return std::max(1, per_cu) * cus;
}

int launch_wgrad_group(int WCI, int NTW, bool t9, const WgradJob* jobs_dev, int njobs, int total_blocks, int lds_bytes,
                       const char* base, hipStream_t s, bool xf) {
    if (njobs <= 0 || total_blocks <= 0) return 0;
    if (lds_bytes > 128 * 1024) return 1;
    if (lds_bytes > 64 * 1024) {      // the 64-channel tiles: two 34-KB X buffers (+ two 17-KB dY buffers at 64 x 64)
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)k_wgrad_group<4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_wgrad_group<4, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_wgrad_group<4, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_wgrad_group<4, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
    }
    if (xf) {      // virtual-activation jobs (opt-in plan): the tile shapes FC-Siam's layers use
#define WG_GROUP_XF(W_, N_, T_) k_wgrad_group<W_, N_, T_, true><<<(unsigned)total_blocks, 256, (size_t)lds_bytes, s>>>(jobs_dev, njobs, base)
        if (WCI == 4 || NTW == 4) return 1;
        WG_DISPATCH(WCI, NTW, t9, WG_GROUP_XF);
#undef WG_GROUP_XF
        return 0;
    }
#define WG_GROUP(W_, N_, T_) k_wgrad_group<W_, N_, T_><<<(unsigned)total_blocks, 256, (size_t)lds_bytes, s>>>(jobs_dev, njobs, base)
    WG_DISPATCH(WCI, NTW, t9, WG_GROUP);
#undef WG_GROUP
    return 0;
}

int launch_wgrad_mfma(const stcd_conv_geom& g, const WgradMfmaPlan& p, const void* in, const void* dout, float* slab,
                      int kpad, int wld, hipStream_t s) {
    if (!p.ok) return 1;
    const WgradJob a = wgrad_make_job(g, p, (int64_t)(intptr_t)in, (int64_t)(intptr_t)dout, (int64_t)(intptr_t)slab, kpad, wld);
    if (a.lds_bytes > 128 * 1024) return 1;
    if (a.lds_bytes > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)k_wgrad_mfma<4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_wgrad_mfma<4, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_wgrad_mfma<4, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)k_wgrad_mfma<4, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
    }
    dim3 grid((unsigned)p.gx, (unsigned)p.gy, (unsigned)p.gz);
    const bool t9 = g.ntaps == 9;
#define WG_ONE(W_, N_, T_) k_wgrad_mfma<W_, N_, T_><<<grid, 256, (size_t)a.lds_bytes, s>>>(a)
    WG_DISPATCH(p.WCI, p.NTW, t9, WG_ONE);
#undef WG_ONE
    return 0;
}

void launch_reduce_dw(const float* slab, int gx, const stcd_conv_geom& g, int K, int N, int kpad, int wld,
                      const PackSpec* ps, float* out, hipStream_t s) {
    const int total = g.ntaps * K * N;
    PackSpec dummy{};
    if (gx >= 128 && total <= 16384)
        k_reduce_dw<64><<<(total + 3) / 4, 256, 0, s>>>(slab, gx, (int64_t)g.ntaps * kpad * wld, g.ntaps, K, N, kpad, wld,
                                                          ps ? *ps : dummy, ps ? 1 : 0, out);
    else
        k_reduce_dw<16><<<(total + 15) / 16, 256, 0, s>>>(slab, gx, (int64_t)g.ntaps * kpad * wld, g.ntaps, K, N, kpad, wld,
                                                            ps ? *ps : dummy, ps ? 1 : 0, out);
}

// =====================================================================================================
// Small-channel convolution (ntaps*Ci <= 288: the 256^2 / 128^2 layers, i.e. the HBM-bound bulk of the network).
//   * the whole filter lives in REGISTERS as MFMA A-fragments (KS x NT x 4 VGPRs), loaded once per wave
//   * a block is persistent over tiles (t = b, b + blocks, ...): the next tile's halo is fetched global -> registers
//     while the current tile's MFMAs run, then dropped into the other LDS buffer -- one barrier per tile, no
//     per-tile filter traffic, no per-tile index divisions (per-lane tap offsets are precomputed)
//   * optional fused BatchNorm statistics: per-channel sum / sum-of-squares of the (rounded) outputs accumulate in
//     registers across the block's tiles and leave as ONE partial row per block: the separate full-tensor
//     statistics pass disappears.  Blocks are split evenly over the BN groups (T1 / T2 halves of the batch).
struct ConvSmallArgs {
    stcd_conv_geom g;
    const bf16* in;
    const bf16* wf;        // fragment-order weights [KS][NTtot][64][8] (mode-B image)
    const float* bias;
    void* out;
    int out_nchw;
    int KS, NTtot, Ci;
    int dymin, dxmin, HH, HWp;
    int tiles_x, tiles_y, ntiles, groups;
    int halo_bytes;
    long long* stat_acc;   // nullable: BatchNorm forward accumulators int64 [BN_REP][groups][2][cpad] (common.h)
    int cpad;
    int tap[9];            // packed taps (fill_taps)
    XfSrc xf;              // XF kernels: `in` is the producer's raw conv output (see k_conv_res)
    // BWD kernels (a data gradient that also sums the BatchNorm backward of the layer it writes dA for, see BwdSum)
    const bf16* bw_Y; int bw_ldy; const float* bw_stat; const float* bw_mask;
    float s1_scale, s2_scale;
};

// __launch_bounds__(256, 2), i.e. at most 256 registers per lane: with the default bound (512 = 256 VGPRs + 256 AGPRs) hipcc selects the
// AGPR form of every MFMA, and wherever an accumulator crosses a basic block (the run-time `ks < KS` tests here, the run-time tap /
// k-step loops of k_conv_mfma) it lives in VGPRs and is copied to and from the AGPRs around each MFMA: 8 v_accvgpr_write + 8
// v_accvgpr_read + an s_nop for the MFMA result per k-step -- 80 of the ~210 VALU instructions of a tile here, 12-15 per MFMA in
// k_conv_mfma, and a fully serialised MFMA chain.  With <= 256 registers the compiler takes the VGPR form and the copies vanish.
// EXACT: the layer has exactly KSMAX k-steps -- the k-loop unrolls without the run-time tests (3 k-steps: 8 -> 16 channels, 5:
// 16 -> 16, 9: 32 -> 16).
// FAST == 2: additionally the map is a whole number of 8 x 16 tiles, every n-tile has its 16 channels and the output is NHWC bf16: the
// epilogue loses its five run-time branches per row (border, layout, channel tail).
// BWD (with FAST == 2, NT == 1): the launch is a data gradient whose output dA belongs to a conv -> BN -> ReLU -> Dropout2d layer; its
// epilogue also loads that layer's Y at the tile's positions and accumulates sum(dz) and sum(dz * xhat) of the ROUNDED dA it stores --
// k_bn_reduce<T, 1>'s second pass over dA and Y (and its launch) disappears.
template <int NT, int KSMAX, bool XF = false, int FAST = 0, bool BWD = false>
__global__ void __launch_bounds__(256, 2)
k_conv_small(const ConvSmallArgs a) {
    constexpr bool EXACT = FAST >= 1, FULL = FAST == 2;
    static_assert(!BWD || (FULL && NT == 1 && !XF), "BWD: full-tile single-n-tile data gradients only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int is = a.g.in_stride, HWp = a.HWp, Ci = a.Ci, nch8 = Ci >> 3;
    const int KS = a.KS;
    bf16* const halo0 = reinterpret_cast<bf16*>(smem);     // two halo buffers, halo_elems apart (plain offset
    const int halo_elems = a.halo_bytes >> 1;              // arithmetic keeps the LDS address space: no flat ops)

    // ---- filter fragments -> registers; per-lane LDS offset of each k-step's 8-channel piece
    bf16x8 wreg[KSMAX][NT];
    int aoff[KSMAX];
    // branch-free: every k-step slot loads (the index clamped to the last real step), so the KSMAX x NT loads leave as one
    // batch; with `if (ks < KS) load` each one sat in its own block behind an s_waitcnt vmcnt(0).  Unused slots are never
    // multiplied (the MFMA section tests ks < KS), their registers just hold a copy of the last step.
#pragma unroll
    for (int ks = 0; ks < KSMAX; ++ks) {
        const int kk = EXACT ? ks : min(ks, KS - 1);
#pragma unroll
        for (int t2 = 0; t2 < NT; ++t2)
            wreg[ks][t2] = *reinterpret_cast<const bf16x8*>(a.wf + ((int64_t)(kk * a.NTtot + t2) * 64 + lane) * 8);
        const int kl = kk * 32 + 8 * q;
        int t = kl / Ci, c = kl - t * Ci;
        if (t >= a.g.ntaps) { t = 0; c = 0; }
        int tw = 0;
#pragma unroll
        for (int tt = 0; tt < 9; ++tt) tw = t == tt ? a.tap[tt] : tw;    // per-lane tap: scalar loads + selects
        aoff[ks] = ((TAP_DY(tw) - a.dymin) * HWp + (TAP_DX(tw) - a.dxmin)) * Ci + c;
    }

    const int bpg = gridDim.x / a.groups, grp = blockIdx.x / bpg, bl = blockIdx.x - grp * bpg;
    const int tpg = a.ntiles / a.groups;             // tiles per group (images are split evenly over groups)
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
        for (int j = 0; j < 4; ++j) s1[t2][j] = s2[t2][j] = 0.f;

    // ---- halo staging: piece i of the tile = pixel (i / nch8), 16-B chunk (i % nch8); <= 3 pieces per thread.
    //      Everything that does not depend on the tile is computed ONCE here (the loop below has no divisions):
    //      per-piece element offset relative to the tile's origin pixel, and its (hy,hx) for the border test.
    const int npieces = a.HH * HWp * nch8;
    constexpr int MAXP = 3;
    int poff[MAXP], phy[MAXP], phx[MAXP];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
        const int i = tid + p * 256;
        const int ch = i % nch8, pix = i / nch8;
        phx[p] = pix % HWp + a.dxmin;
        phy[p] = (i < npieces) ? pix / HWp + a.dymin : (1 << 28);     // out-of-range piece: always fails the row test
        poff[p] = (phy[p] * a.g.wi + phx[p]) * a.g.ldi + ch * 8;
    }
    // bias of this lane's channels
    float bv[NT][4];
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cb = t2 * 16 + 4 * q + j;
            bv[t2][j] = (a.bias && cb < a.g.co) ? a.bias[cb] : 0.f;
        }
    // BWD: this lane's four channels of the destination layer's published statistics (the block walks tiles of ONE group)
    float bw_mean[4], bw_inv[4], bw_sc[4], bw_sh[4];
    if constexpr (BWD) {
        const float* st = a.bw_stat + (int64_t)grp * 4 * a.g.co + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; ++j) { bw_mean[j] = st[j]; bw_inv[j] = st[a.g.co + j]; bw_sc[j] = st[2 * a.g.co + j]; bw_sh[j] = st[3 * a.g.co + j]; }
    }
    // tile walk: tile -> (n, ty, tx) advanced incrementally by bpg
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int dn = bpg / tiles_img, drem = bpg - dn * tiles_img, dty = drem / a.tiles_x, dtx = drem - dty * a.tiles_x;
    int tile = grp * tpg + bl;
    const int tile_end = (grp + 1) * tpg;
    int tn = tile / tiles_img, trem = tile - tn * tiles_img, tty = trem / a.tiles_x, ttx = trem - tty * a.tiles_x;
    auto advance = [&](int& n_, int& y_, int& x_) {
        x_ += dtx; if (x_ >= a.tiles_x) { x_ -= a.tiles_x; ++y_; }
        y_ += dty; if (y_ >= a.tiles_y) { y_ -= a.tiles_y; ++n_; }
        n_ += dn;
    };
    uint4 pre[MAXP];
    unsigned xf_ok = 0;                                                  // XF: bit p = piece p of the fetched tile lies inside the image
    float4 xf_m0 = make_float4(1.f, 1.f, 1.f, 1.f), xf_m1 = xf_m0;      // XF: Dropout2d factors of the fetched tile's (image, 8 channels)
    float xf_sc[8], xf_sh[8];                                            // XF: this thread's 8 channels (tid % nch8 is the same for every piece)
    float xf_fs[8], xf_fh[8];                                            // ... folded with the fetched tile's Dropout2d factors
    const int xch = (tid % nch8) * 8;
    auto fetch = [&](int n_, int y_, int x_) {
        const int gy0 = y_ * 8 * is, gx0 = x_ * 16 * is;
        const bf16* org = a.in + (((int64_t)n_ * a.g.hi + gy0) * a.g.wi + gx0) * a.g.ldi;
        if constexpr (XF) xf_ok = 0;
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            const unsigned gy = (unsigned)(gy0 + phy[p]), gx = (unsigned)(gx0 + phx[p]);
            pre[p] = make_uint4(0, 0, 0, 0);
            if (gy < (unsigned)a.g.hi && gx < (unsigned)a.g.wi) {
                pre[p] = *reinterpret_cast<const uint4*>(org + poff[p]);
                if constexpr (XF) xf_ok |= 1u << p;
            }
        }
        if constexpr (XF) {
            if (a.xf.mask) {
                const float4* mp = reinterpret_cast<const float4*>(a.xf.mask + (int64_t)n_ * a.xf.C + xch);
                xf_m0 = mp[0]; xf_m1 = mp[1];
            }
        }
    };
    auto stash = [&](int buf_) {
        if constexpr (XF) {
            xf_fold8(xf_sc, xf_sh, xf_m0, xf_m1, xf_fs, xf_fh);
#pragma unroll
            for (int p = 0; p < MAXP; ++p) {
                const int i = tid + p * 256;
                if (i < npieces) *reinterpret_cast<uint4*>(halo0 + buf_ * halo_elems + i * 8) = a.xf.on == 2 ? pre[p] : xf_act8(pre[p], xf_fs, xf_fh, 0u - ((xf_ok >> p) & 1u));
            }
        } else {
#pragma unroll
            for (int p = 0; p < MAXP; ++p) {
                const int i = tid + p * 256;
                if (i < npieces) *reinterpret_cast<uint4*>(halo0 + buf_ * halo_elems + i * 8) = pre[p];
            }
        }
    };

    int buf = 0;
    if (tile < tile_end) fetch(tn, tty, ttx);
    if constexpr (XF) {      // the producer's scale / shift (block 0 publishes them and updates the running statistics); a block walks
        float* tab = reinterpret_cast<float*>(smem + 2 * a.halo_bytes);      // tiles of ONE group: its 8 channels stay in registers
        bn_fwd_table(tab, a.xf.facc, a.xf.gamma, a.xf.beta, a.xf.rmean, a.xf.rvar, a.xf.stat, a.xf.C, a.xf.groups, a.xf.ppg,
                     a.xf.momentum, a.xf.eps, a.xf.publish != 0 && blockIdx.x == 0);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) { xf_sc[j] = tab[(grp * 2 + 0) * a.xf.C + xch + j]; xf_sh[j] = tab[(grp * 2 + 1) * a.xf.C + xch + j]; }
    }
    if (tile < tile_end) stash(0);
    __syncthreads();
    const int base0 = ((wid * 2) * is * HWp + r * is) * Ci, rowstep = is * HWp * Ci;
    for (; tile < tile_end; tile += bpg, buf ^= 1) {
        int nn = tn, ny = tty, nx = ttx;
        advance(nn, ny, nx);
        const bool have_next = tile + bpg < tile_end;
        if (have_next) fetch(nn, ny, nx);
        uint2 bw_y[2];                     // BWD: Y of this tile's two rows (this lane's 4 channels), in flight under the MFMAs
        float4 bw_mk = make_float4(1.f, 1.f, 1.f, 1.f);
        if constexpr (BWD) {
            const bf16* yrow = a.bw_Y + (((int64_t)tn * a.g.ho + tty * 8 + wid * 2) * a.g.wo + ttx * 16 + r) * a.bw_ldy + 4 * q;
            bw_y[0] = *reinterpret_cast<const uint2*>(yrow);
            bw_y[1] = *reinterpret_cast<const uint2*>(yrow + (int64_t)a.g.wo * a.bw_ldy);
            if (a.bw_mask) bw_mk = *reinterpret_cast<const float4*>(a.bw_mask + (int64_t)tn * a.g.co + 4 * q);
        }
        // ---- MFMAs of this tile
        f32x4 acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t2 = 0; t2 < NT; ++t2) acc[m][t2] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16* hb = halo0 + buf * halo_elems;
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            if (EXACT || ks < KS) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(hb + base0 + aoff[ks]);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(hb + base0 + rowstep + aoff[ks]);
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2) {
                    acc[0][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks][t2], a0, acc[0][t2], 0, 0, 0);
                    acc[1][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks][t2], a1, acc[1][t2], 0, 0, 0);
                }
            }
        }
        // the prefetched halo goes to the other buffer BEFORE this tile's stores are issued: the wait for the loads
        // must not also wait for fresh stores (vmcnt counts both)
        if (have_next) stash(buf ^ 1);
        // ---- epilogue of this tile
        const int mx = ttx * 16 + r;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int my = tty * 8 + wid * 2 + m;
            if constexpr (FULL) {
                const int oy = my * a.g.out_stride + a.g.oy0, ox = mx * a.g.out_stride + a.g.ox0;
                bf16* orow = reinterpret_cast<bf16*>(a.out) + (((int64_t)tn * a.g.ho + oy) * a.g.wo + ox) * a.g.ldo + 4 * q;
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2) {
                    uint2 pk;
                    pk.x = pack_bf16x2(acc[m][t2][0] + bv[t2][0], acc[m][t2][1] + bv[t2][1]);
                    pk.y = pack_bf16x2(acc[m][t2][2] + bv[t2][2], acc[m][t2][3] + bv[t2][3]);
                    *reinterpret_cast<uint2*>(orow + t2 * 16) = pk;
                    const float rv[4] = {__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u),
                                         __uint_as_float(pk.y << 16), __uint_as_float(pk.y & 0xffff0000u)};     // the rounded outputs
                    if constexpr (BWD) {
                        const float y[4] = {__uint_as_float(bw_y[m].x << 16), __uint_as_float(bw_y[m].x & 0xffff0000u),
                                            __uint_as_float(bw_y[m].y << 16), __uint_as_float(bw_y[m].y & 0xffff0000u)};
                        const float mk[4] = {bw_mk.x, bw_mk.y, bw_mk.z, bw_mk.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {       // the arithmetic of k_bn_reduce<T, 1>
                            const float z = y[j] * bw_sc[j] + bw_sh[j];
                            float dz = rv[j] * mk[j];
                            if (!(z > 0.f)) dz = 0.f;
                            s1[t2][j] += dz;
                            s2[t2][j] += dz * (y[j] - bw_mean[j]) * bw_inv[j];
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { s1[t2][j] += rv[j]; s2[t2][j] += rv[j] * rv[j]; }
                    }
                }
                continue;
            }
            if (my >= a.g.hm || mx >= a.g.wm) continue;
            const int oy = my * a.g.out_stride + a.g.oy0, ox = mx * a.g.out_stride + a.g.ox0;
            if (a.out_nchw) {
                float* o = reinterpret_cast<float*>(a.out) + ((int64_t)tn * a.g.co * a.g.ho + oy) * a.g.wo + ox;
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int cb = t2 * 16 + 4 * q + j;
                        if (cb < a.g.co) o[(int64_t)cb * a.g.ho * a.g.wo] = acc[m][t2][j] + bv[t2][j];
                    }
            } else {
                bf16* orow = reinterpret_cast<bf16*>(a.out) + (((int64_t)tn * a.g.ho + oy) * a.g.wo + ox) * a.g.ldo + 4 * q;
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2) {
                    const int cb = t2 * 16 + 4 * q;
                    if (cb >= a.g.co) continue;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[m][t2][j] + bv[t2][j];
                    bf16* o = orow + t2 * 16;
                    if (cb + 3 < a.g.co) {
                        uint2 pk;
                        pk.x = pack_bf16x2(v[0], v[1]);
                        pk.y = pack_bf16x2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(o) = pk;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (cb + j < a.g.co) o[j] = (bf16)v[j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float rv = round_as<bf16>(v[j]);
                        s1[t2][j] += rv;
                        s2[t2][j] += rv * rv;
                    }
                }
            }
        }
        tn = nn; tty = ny; ttx = nx;
        barrier_lds();     // NOT __syncthreads(): that would wait for this tile's output stores to reach memory
    }

    // ---- fused BN statistics: lanes sharing q hold the same 4 channels -> xor-reduce over r, then over the 4 waves
    if (a.stat_acc) {
        float* red = reinterpret_cast<float*>(smem);     // [4 waves][NT][4 q][4 j][2]
#pragma unroll
        for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = s1[t2][j], y = s2[t2][j];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64); }
                if (r == 0) {
                    red[(((wid * NT + t2) * 4 + q) * 4 + j) * 2] = x;
                    red[(((wid * NT + t2) * 4 + q) * 4 + j) * 2 + 1] = y;
                }
            }
        __syncthreads();
        for (int i = tid; i < NT * 16 * 2; i += 256) {
            const int which = i / (NT * 16), c = i - which * NT * 16;
            const int t2 = c >> 4, qq = (c >> 2) & 3, j = c & 3;
            float acc_ = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) acc_ += red[(((w * NT + t2) * 4 + qq) * 4 + j) * 2 + which];
            if (c < a.cpad) bn_acc_add(a.stat_acc, bl, a.groups, a.cpad, grp, which, c, acc_, which ? a.s2_scale : a.s1_scale);
        }
    }
}

bool conv_small_ok(const stcd_conv_geom& g, const ConvMfmaPlan& p) {
    if (!p.ok || (!p.modeB && p.nchunks != 1)) return false;
    const int K = g.ntaps * g.ci;
    const int ks = (K + 31) / 32;
    const int nt = (g.co + 15) / 16;
    int dymin, dymax, dxmin, dxmax;
    taps_extent(g, &dymin, &dymax, &dxmin, &dxmax);
    const int HH = 7 * g.in_stride + (dymax - dymin) + 1, HWp = 15 * g.in_stride + (dxmax - dxmin) + 1;
    // measured: the register-resident filter only pays for one n-tile and <= 5 k-steps (Ci <= 16, Co <= 16);
    // wider cases run faster on the generic kernel (fewer registers, more waves)
    // (STCD_SMALL_NT2=1: also two n-tiles, i.e. the 16 -> 32-channel layers -- re-measured in round 4)
    static const int nt2 = [] { const char* e = getenv("STCD_SMALL_NT2"); return e ? atoi(e) : 0; }();
    return ks <= 5 && nt <= (nt2 ? 2 : 1) && g.ci <= 32 && HH * HWp * (g.ci / 8) <= 3 * 256;
}

// blocks the launcher will use (the BN-partial slab is sized from this)
int conv_small_blocks(const stcd_conv_geom& g, int groups) {
    const int64_t ntiles = (int64_t)g.n * ((g.hm + 7) / 8) * ((g.wm + 15) / 16);
    int64_t per_group = ntiles / groups;
    int64_t bpg = std::min<int64_t>(per_group, 2048 / groups);
    if (bpg < 1) bpg = 1;
    return (int)(bpg * groups);
}

bool conv_small_bwdsum_ok(const stcd_conv_geom& g) {
    const int ks = (g.ntaps * g.ci + 31) / 32;
    return g.co == 16 && g.hm % 8 == 0 && g.wm % 16 == 0 && g.hm == g.ho && g.wm == g.wo && g.out_stride == 1 && g.oy0 == 0 && g.ox0 == 0 &&
           (ks == 3 || ks == 5);
}
int launch_conv_small(const stcd_conv_geom& g, const void* in, const void* wf_modeB, const float* bias, void* out,
                      bool out_nchw, int groups, long long* stat_acc, int cpad, hipStream_t s, const XfSrc* xf, const BwdSum* bs) {
    ConvSmallArgs a;
    a.bw_Y = nullptr; a.bw_ldy = 0; a.bw_stat = nullptr; a.bw_mask = nullptr; a.s1_scale = BN_FS1; a.s2_scale = BN_FS2;
    a.g = g;
    a.in = (const bf16*)in; a.wf = (const bf16*)wf_modeB; a.bias = bias; a.out = out; a.out_nchw = out_nchw ? 1 : 0;
    a.Ci = g.ci;
    a.KS = (g.ntaps * g.ci + 31) / 32;
    const int nt = (g.co + 15) / 16;
    a.NTtot = nt;
    int dymax, dxmax;
    taps_extent(g, &a.dymin, &dymax, &a.dxmin, &dxmax);
    a.HH = 7 * g.in_stride + (dymax - a.dymin) + 1;
    a.HWp = 15 * g.in_stride + (dxmax - a.dxmin) + 1;
    a.tiles_x = (g.wm + 15) / 16;
    a.tiles_y = (g.hm + 7) / 8;
    a.ntiles = g.n * a.tiles_x * a.tiles_y;
    a.groups = groups;
    a.halo_bytes = (a.HH * a.HWp * g.ci * 2 + 255) & ~255;
    a.stat_acc = stat_acc;
    a.cpad = cpad;
    fill_taps(g, a.tap);
    if (g.n % groups != 0) return 1;
    const int blocks = conv_small_blocks(g, groups);
    size_t lds = std::max<size_t>(2 * (size_t)a.halo_bytes, 4 * 2 * 16 * 2 * 4 * 2);
    const bool use_xf = xf && xf->on;
    if (use_xf) {
        if (xf->C != g.ci || xf->groups != groups) return 1;
        a.xf = *xf;
        lds = 2 * (size_t)a.halo_bytes + (size_t)xf->groups * 2 * xf->C * 4;
    }
    if (bs) {      // fused BatchNorm-backward sums: the tiles are partitioned by the DESTINATION layer's groups
        if (!conv_small_bwdsum_ok(g) || out_nchw || use_xf || nt != 1 || stat_acc || bs->groups != groups || !bs->acc) return 1;
        a.bw_Y = (const bf16*)bs->Y; a.bw_ldy = bs->ldy; a.bw_stat = bs->stat; a.bw_mask = bs->mask;
        a.stat_acc = bs->acc; a.cpad = g.co; a.s1_scale = BN_BS; a.s2_scale = BN_BS;
        if (a.KS == 3) k_conv_small<1, 3, false, 2, true><<<blocks, 256, lds, s>>>(a);
        else k_conv_small<1, 5, false, 2, true><<<blocks, 256, lds, s>>>(a);
        return 0;
    }
    // STCD_SMALL_FAST=0: the generic kernel for every layer; 1: exact k-step count only; 2 (default): + full-tile NHWC epilogue
    static const int fast_on = [] { const char* e = getenv("STCD_SMALL_FAST"); return e ? atoi(e) : 2; }();
    const bool full = fast_on >= 2 && !out_nchw && g.co == nt * 16 && g.hm % 8 == 0 && g.wm % 16 == 0;
#define LAUNCH_SMALL(N_, K_, F_) do { if (use_xf) k_conv_small<N_, K_, true, F_><<<blocks, 256, lds, s>>>(a); else k_conv_small<N_, K_, false, F_><<<blocks, 256, lds, s>>>(a); } while (0)
#define LAUNCH_SMALL_K(N_, K_) do { if (full) LAUNCH_SMALL(N_, K_, 2); else LAUNCH_SMALL(N_, K_, 1); } while (0)
#define LAUNCH_SMALL_N(N_) do { \
        if (fast_on >= 1 && a.KS == 3) LAUNCH_SMALL_K(N_, 3); \
        else if (fast_on >= 1 && a.KS == 5) LAUNCH_SMALL_K(N_, 5); \
        else if (a.KS <= 5) LAUNCH_SMALL(N_, 5, 0); \
        else if (fast_on >= 1 && a.KS == 9) LAUNCH_SMALL_K(N_, 9); \
        else LAUNCH_SMALL(N_, 9, 0); } while (0)
    if (nt == 1) LAUNCH_SMALL_N(1); else LAUNCH_SMALL_N(2);
#undef LAUNCH_SMALL_K
#undef LAUNCH_SMALL_N
#undef LAUNCH_SMALL
    return 0;
}

// =====================================================================================================
// 3x3 stride-1 convolutions with Ci % 32 == 0 (forward, data gradient, stride-1 transposed): resident-filter kernel.
//   * block = 4 waves, output tile 16 rows x 16 columns; wave w owns rows 4w..4w+3.  A wave reads the 6 halo rows it
//     touches ONCE per column shift and reuses them for the three row taps (18 activation-fragment reads feed
//     36*NT MFMAs per 32-channel step), so the LDS pipe stays well under the MFMA rate.
//   * the block's filter slice (NT n-tiles x all Ci x 9 taps, fragment order) is loaded into LDS once; the block is
//     persistent over tiles of its BN group, walking (tile, 32-channel chunk) steps with the next step's halo chunk
//     in flight in registers and two LDS halo buffers: one barrier per step, none inside it.
//   * epilogue: bias, bf16 NHWC store, optional fused BN statistics (per-lane running sums over all the block's
//     tiles, reduced once at the end into one partial row per block).
struct ConvResArgs {
    stcd_conv_geom g;
    const bf16* in;
    const bf16* wf;        // mode-A fragment image of conv_mfma_plan: [Ci/CiB][tap][CiB/32][NTtot][64][8]
    const float* bias;     // nullable
    bf16* out;
    int NTtot, KSp;        // n-tiles of the image, k-steps per image chunk (CiB / 32)
    int nchunks;           // Ci / 32
    int nslices, P, groups;
    int tiles_x, tiles_y, ntiles;
    int filt_bytes;        // LDS bytes of the filter slice
    long long* stat_acc;   // nullable: per-channel sum accumulators int64 [BN_REP][groups][2][cpad] (common.h)
    int cpad;              // channels [stat_c0, stat_c0 + cpad) of the output are summed (BatchNorm statistics of a forward
    int stat_c0;           // conv: all of them; bias gradient of the transposed conv that produced a concat slice: that slice)
    float s1_scale, s2_scale;
    unsigned in_bytes;     // size of the input tensor (buffer-load range check)
    int8_t tix[3][3];      // tap index of every (row shift, column shift)
    XfSrc xf;              // XF kernels: `in` is the producer's raw conv output; BN-affine + ReLU + Dropout2d applied while staging
    // BWD kernels: a data gradient that also forms the BatchNorm-backward sums of the layer it writes dA for (BwdSum, common.h)
    const bf16* bw_Y; int bw_ldy; const float* bw_stat; const float* bw_mask;
};

constexpr int RES_HW = 18;     // halo edge of the 16 x 16 output tile

// CW = channels per pipeline step (32: 64-B pixel rows; 64: full 128-B lines and half as many steps -- used whenever
// Ci % 64 == 0, where the wider co slice it leaves room for also halves the re-reads of X through L2).
// PIPE: the (32-channel chunk, column shift) stages of a step are software-pipelined -- the LDS reads of stage s + 1 (6 halo
// fragments + 3 x NT filter fragments) are issued before the MFMAs of stage s, into a second register set; without it the
// compiler issues every stage's reads right in front of their first use and the wave waits out the LDS latency 3-6 times per step.
// XF: the input is a "virtual activation" (XfSrc, common.h): every staged 16-B piece goes through xf_act8 between its buffer load
// and its LDS store; the producer's scale / shift table is built in the block's prologue (block 0 publishes it and updates the
// running statistics, as k_bn_act's block 0 did), the Dropout2d factors of the step's image ride along with the halo loads.
// BWD: see k_conv_small -- the epilogue also loads the destination layer's Y at the tile's positions and accumulates sum(dz),
// sum(dz * xhat) of the rounded dA it stores (the statistics registers and their block reduction are the forward's).
template <int NT, int CW, bool SH = false, bool PIPE = false, bool XF = false, bool BWD = false>
__global__ void __launch_bounds__(256, (CW == 64 && NT == 4) ? 1 : 2)
k_conv_res(const ConvResArgs a) {
    static_assert(!BWD || (NT <= 2 && !SH && !XF), "BWD: one or two n-tiles, double halo, plain input");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KSC = CW / 32, CH8 = CW / 8, LG8 = CW == 64 ? 3 : 2, SW = CH8 - 1;
    constexpr int HALO_BYTES = RES_HW * RES_HW * CW * 2;
    constexpr int NPIECE = RES_HW * RES_HW * CH8, MAXP = (NPIECE + 255) / 256, PIXSTEP = 256 >> LG8;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int P = a.P;
    const int bl = blockIdx.x % P, rest = blockIdx.x / P, slice = rest % a.nslices, grp = rest / a.nslices;
    char* const filt = smem;                                  // [32-ch chunk][tap][NT][64 lanes][16 B]
    char* const halo0 = smem + a.filt_bytes;                  // two halo buffers, HALO_BYTES apart
    float* const xf_tab = reinterpret_cast<float*>(smem + a.filt_bytes + (SH ? 1 : 2) * HALO_BYTES);   // XF: [groups][2][Ci] scale, shift
    const int ntaps = 9;
    const int nsteps = a.nchunks / KSC;                       // pipeline steps per tile (nchunks = Ci / 32)

    // ---- the block's filter slice (once per block): wave w takes fragments w, w+4, ...  The first 9 loads per lane
    //      (the whole slice of the common 36-fragment cases) are requested HERE, before any other work, so the staging
    //      plan below is computed while they are in flight.
    const int nfrag = a.nchunks * ntaps * NT;
    constexpr int FB = 9;
    auto filt_src = [&](int f) -> int64_t {
        const int ntl = f % NT, ft = f / NT, t = ft % ntaps, c32 = ft / ntaps;
        return ((int64_t)((c32 / a.KSp) * ntaps + t) * a.KSp + (c32 % a.KSp)) * a.NTtot + slice * NT + ntl;
    };
    // Branch-free on purpose: a fragment index past the slice is clamped to the last fragment for the load AND for the store
    // (the same bytes written twice), and the wave index is made scalar.  With `if (f < nfrag) store` the compiler had sunk
    // every load into its store's block: nine (and, for deep slices, eighteen) fully serialised load -> wait -> store round
    // trips in the prologue of every block.
    const int swid = __builtin_amdgcn_readfirstlane(wid);
    uint4 fv[FB];
#pragma unroll
    for (int k = 0; k < FB; ++k) fv[k] = *reinterpret_cast<const uint4*>(a.wf + (filt_src(min(swid + 4 * k, nfrag - 1)) * 64 + lane) * 8);
    // (no sched_barrier here: with one, the compiler keeps fv[] in scratch memory)

    // ---- halo staging plan: piece i = tid + p*256 = pixel (i >> LG8), 16-B chunk ch = tid & SW (the same for every p);
    //      only the halo coordinates are kept per piece, offsets are rebuilt from them (2 FMAs) at use.
    const int ch = tid & SW;
    int pyx[MAXP], poff[MAXP], plds[MAXP];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
        const int pix = min((tid >> LG8) + p * PIXSTEP, RES_HW * RES_HW - 1);
        const int hx = pix % RES_HW, hy = pix / RES_HW;
        pyx[p] = ((hy - 1) << 16) | ((hx - 1) & 0xffff);
        poff[p] = ((hy * a.g.wi + hx) * a.g.ldi + ch * 8) * 2;        // bytes past the halo's top-left pixel
        plds[p] = (pix * CH8 + (ch ^ ((hx >> 1) & SW))) * 16;
    }
    float bv[NT][4];
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2) {
        const int cb = (slice * NT + t2) * 16 + 4 * q;
        if (a.bias && cb + 3 < a.g.co) {                      // flat parameter tensors are 16-B aligned
            const float4 b4 = *reinterpret_cast<const float4*>(a.bias + cb);
            bv[t2][0] = b4.x; bv[t2][1] = b4.y; bv[t2][2] = b4.z; bv[t2][3] = b4.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[t2][j] = (a.bias && cb + j < a.g.co) ? a.bias[cb + j] : 0.f;
        }
    }
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
        for (int j = 0; j < 4; ++j) s1[t2][j] = s2[t2][j] = 0.f;

    // BWD: this lane's channels of the destination layer's published statistics (the block walks tiles of ONE group)
    float bw_mean[NT][4], bw_inv[NT][4], bw_sc[NT][4], bw_sh[NT][4];
    if constexpr (BWD) {
#pragma unroll
        for (int t2 = 0; t2 < NT; ++t2) {
            const float* st = a.bw_stat + (int64_t)grp * 4 * a.g.co + (slice * NT + t2) * 16 + 4 * q;
#pragma unroll
            for (int j = 0; j < 4; ++j) { bw_mean[t2][j] = st[j]; bw_inv[t2][j] = st[a.g.co + j]; bw_sc[t2][j] = st[2 * a.g.co + j]; bw_sh[t2][j] = st[3 * a.g.co + j]; }
        }
    }
    // tile walk inside the group: tiles bl, bl + P, ...
    const int tpg = a.ntiles / a.groups;
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int dn = P / tiles_img, drem = P - dn * tiles_img, dty = drem / a.tiles_x, dtx = drem - dty * a.tiles_x;
    const int tile0 = grp * tpg + bl;
    const int tile_end = (grp + 1) * tpg;

    // ---- software pipeline: the halo chunk of step s+1 is requested into registers before the MFMAs of step s and
    //      parked in the other LDS buffer after them; one barrier per step.  A step = (tile, CW-channel chunk).
    // (Round 4 tried TWO steps in flight -- step s+2 requested before the MFMAs of step s, step s+1 parked from a second register
    //  set: MEASURED SLOWER on every family (diff conv 0.985 -> 1.00-1.02 ms, SNUNet 12.7 -> 13.5 ms, SegCD 9.3 -> 9.85 ms).  The
    //  compiled loop shows why: hipcc allocates the offset temporaries of the new loads INTO the destination registers of the other
    //  set and guards them with s_waitcnt vmcnt(4) / (2) / (0) in front of the fetch, so nothing more was in flight than before,
    //  with 24-44 more live registers.  One step in flight stays.)
    struct Pos { int tile, n, y, x, c; };
    struct XfSet { unsigned ok; float4 m0, m1; };      // XF: validity bits and Dropout2d factors that travel with a register set
    uint4 preA[MAXP];
    XfSet xsA{0u, make_float4(1.f, 1.f, 1.f, 1.f), make_float4(1.f, 1.f, 1.f, 1.f)};
    // Halo loads are raw buffer loads: the descriptor starts one row + one pixel BEFORE the tensor, so offsets relative
    // to a tile's halo corner are never negative; a piece outside the image gets an offset beyond num_records and the
    // hardware returns zeros -- no address select, no data select, no 64-bit address arithmetic per piece.
    const int64_t lead = ((int64_t)a.g.wi + 1) * a.g.ldi * 2;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.in)) - lead, (short)0, (int)(a.in_bytes + (unsigned)lead), 0x00020000);
#define RES_ADV(P_)                                                                                                    \
    do {                                                                                                               \
        if (++(P_).c == nsteps) {                                                                                      \
            (P_).c = 0; (P_).tile += P;                                                                                \
            (P_).x += dtx; if ((P_).x >= a.tiles_x) { (P_).x -= a.tiles_x; ++(P_).y; }                                 \
            (P_).y += dty; if ((P_).y >= a.tiles_y) { (P_).y -= a.tiles_y; ++(P_).n; }                                 \
            (P_).n += dn;                                                                                              \
        }                                                                                                              \
    } while (0)
#define RES_FETCH(P_, PRE_, XS_)                                                                                       \
    do {                                                                                                               \
        const int gy0_ = (P_).y * 16, gx0_ = (P_).x * 16;                                                               \
        const unsigned soff_ = (unsigned)(((((int64_t)(P_).n * a.g.hi + gy0_) * a.g.wi + gx0_) * a.g.ldi + (P_).c * CW) * 2); \
        if constexpr (XF) (XS_).ok = 0;                                                                                \
        _Pragma("unroll") for (int p = 0; p < MAXP; ++p) {                                                             \
            const int hy_ = pyx[p] >> 16, hx_ = (int)(short)(pyx[p] & 0xffff);                                         \
            const bool ok_ = (unsigned)(gy0_ + hy_) < (unsigned)a.g.hi && (unsigned)(gx0_ + hx_) < (unsigned)a.g.wi;   \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok_ ? (unsigned)poff[p] : 0x80000000u, soff_, 0); \
            (PRE_)[p] = make_uint4(v_[0], v_[1], v_[2], v_[3]);                                                        \
            if constexpr (XF) (XS_).ok |= ok_ ? (1u << p) : 0u;                                                        \
        }                                                                                                              \
        if constexpr (XF) {                                                                                            \
            if (a.xf.mask) {                                                                                           \
                const float4* mp_ = reinterpret_cast<const float4*>(a.xf.mask + (int64_t)(P_).n * a.xf.C + (P_).c * CW + ch * 8); \
                (XS_).m0 = mp_[0]; (XS_).m1 = mp_[1];                                                                  \
            }                                                                                                          \
        }                                                                                                              \
    } while (0)
#define RES_STASH(BUF_, P_, PRE_, XS_)                                                                                 \
    do {                                                                                                               \
        if constexpr (XF) {                                                                                            \
            float sc_[8], sh_[8];                                                                                      \
            const float* tb_ = xf_tab + (grp * 2) * a.xf.C + (P_).c * CW + ch * 8;                                     \
            xf_fold8(tb_, tb_ + a.xf.C, (XS_).m0, (XS_).m1, sc_, sh_);                                                 \
            _Pragma("unroll") for (int p = 0; p < MAXP; ++p)                                                           \
                if ((tid >> LG8) + p * PIXSTEP < RES_HW * RES_HW)                                                      \
                    *reinterpret_cast<uint4*>(halo0 + (BUF_) * HALO_BYTES + plds[p]) = a.xf.on == 2 ? (PRE_)[p] : xf_act8((PRE_)[p], sc_, sh_, 0u - (((XS_).ok >> p) & 1u)); \
        } else {                                                                                                       \
            _Pragma("unroll") for (int p = 0; p < MAXP; ++p)                                                           \
                if ((tid >> LG8) + p * PIXSTEP < RES_HW * RES_HW)                                                      \
                    *reinterpret_cast<uint4*>(halo0 + (BUF_) * HALO_BYTES + plds[p]) = (PRE_)[p];                      \
        }                                                                                                              \
    } while (0)

    f32x4 acc[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t2 = 0; t2 < NT; ++t2) acc[m][t2] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane halo read offsets: row (4*wid + hr), column r + dx, 16-B chunk ks*4 + q (swizzled by the column)
    int aoff[KSC][3];
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
            aoff[ks][dx] = (((4 * wid) * RES_HW + r + dx) * CH8 + ((ks * 4 + q) ^ (((r + dx) >> 1) & SW))) * 16;

    Pos cur{tile0, 0, 0, 0, 0};
    cur.n = tile0 / tiles_img;
    { const int trem = tile0 - cur.n * tiles_img; cur.y = trem / a.tiles_x; cur.x = trem - cur.y * a.tiles_x; }
    Pos nxt = cur; RES_ADV(nxt);
    int buf = 0;
    if (cur.tile < tile_end) RES_FETCH(cur, preA, xsA);    // first halo chunk in flight while the filter is staged
    // ---- filter slice -> LDS: first batch was requested at kernel entry; deeper slices take more rounds
    {
#pragma unroll
        for (int k = 0; k < FB; ++k)
            *reinterpret_cast<uint4*>(filt + ((int64_t)min(swid + 4 * k, nfrag - 1) * 64 + lane) * 16) = fv[k];
        for (int f0 = swid + 4 * FB; f0 < nfrag; f0 += 4 * FB) {
            uint4 v[FB];
#pragma unroll
            for (int k = 0; k < FB; ++k) v[k] = *reinterpret_cast<const uint4*>(a.wf + (filt_src(min(f0 + 4 * k, nfrag - 1)) * 64 + lane) * 8);
#pragma unroll
            for (int k = 0; k < FB; ++k)
                *reinterpret_cast<uint4*>(filt + ((int64_t)min(f0 + 4 * k, nfrag - 1) * 64 + lane) * 16) = v[k];
        }
    }
    if constexpr (XF) {      // the producer's scale / shift for this block's group(s), before the first transformed stash
        bn_fwd_table(xf_tab, a.xf.facc, a.xf.gamma, a.xf.beta, a.xf.rmean, a.xf.rvar, a.xf.stat, a.xf.C, a.xf.groups, a.xf.ppg,
                     a.xf.momentum, a.xf.eps, a.xf.publish != 0 && blockIdx.x == 0);
        __syncthreads();
    }
    if (cur.tile < tile_end) RES_STASH(0, cur, preA, xsA);
    __syncthreads();
    while (cur.tile < tile_end) {
        const bool have_next = nxt.tile < tile_end;
        if (have_next) RES_FETCH(nxt, preA, xsA);
        uint2 bw_y[4][NT];                 // BWD: Y of this tile's four rows (this lane's channels), in flight under the last step's MFMAs
        float4 bw_mk[NT];
        if constexpr (BWD) {
            if (cur.c == nsteps - 1) {
                const int mxb = cur.x * 16 + r;
                const bf16* yb = a.bw_Y + (((int64_t)cur.n * a.g.ho + cur.y * 16 + wid * 4) * a.g.wo + mxb) * a.bw_ldy + slice * NT * 16 + 4 * q;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const bool in_ = cur.y * 16 + wid * 4 + m < a.g.hm && mxb < a.g.wm;
#pragma unroll
                    for (int t2 = 0; t2 < NT; ++t2)
                        bw_y[m][t2] = in_ ? *reinterpret_cast<const uint2*>(yb + (int64_t)m * a.g.wo * a.bw_ldy + t2 * 16) : make_uint2(0u, 0u);
                }
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2)
                    bw_mk[t2] = a.bw_mask ? *reinterpret_cast<const float4*>(a.bw_mask + (int64_t)cur.n * a.g.co + (slice * NT + t2) * 16 + 4 * q)
                                          : make_float4(1.f, 1.f, 1.f, 1.f);
            }
        }
        const char* hb = halo0 + (SH ? 0 : buf) * HALO_BYTES;
        if constexpr (PIPE) {
            constexpr int NSTG = KSC * 3;
            bf16x8 ab[2][6], wb[2][3][NT];
            const char* const fb0 = filt + (int64_t)(cur.c * KSC) * (ntaps * NT * 1024) + lane * 16;
#define RES_LOAD_STAGE(S_, B_)                                                                                        \
            do {                                                                                                      \
                constexpr int ks_ = (S_) / 3, dx_ = (S_) % 3;                                                         \
                _Pragma("unroll") for (int hr = 0; hr < 6; ++hr)                                                      \
                    ab[B_][hr] = *reinterpret_cast<const bf16x8*>(hb + aoff[ks_][dx_] + hr * (RES_HW * CW * 2));      \
                _Pragma("unroll") for (int dy = 0; dy < 3; ++dy)                                                      \
                    _Pragma("unroll") for (int t2 = 0; t2 < NT; ++t2)                                                 \
                        wb[B_][dy][t2] = *reinterpret_cast<const bf16x8*>(fb0 + ks_ * (ntaps * NT * 1024) + a.tix[dy][dx_] * (NT * 1024) + t2 * 1024); \
            } while (0)
#define RES_MMA_STAGE(B_)                                                                                             \
            do {                                                                                                      \
                _Pragma("unroll") for (int dy = 0; dy < 3; ++dy)                                                      \
                    _Pragma("unroll") for (int t2 = 0; t2 < NT; ++t2)                                                 \
                        _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                 \
                            acc[m][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[B_][dy][t2], ab[B_][m + dy], acc[m][t2], 0, 0, 0); \
            } while (0)
            RES_LOAD_STAGE(0, 0);
            if constexpr (NSTG > 1) { RES_LOAD_STAGE(1, 1); __builtin_amdgcn_sched_barrier(0); }
            RES_MMA_STAGE(0);
            if constexpr (NSTG > 2) { RES_LOAD_STAGE(2, 0); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (NSTG > 1) RES_MMA_STAGE(1);
            if constexpr (NSTG > 3) { RES_LOAD_STAGE(3, 1); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (NSTG > 2) RES_MMA_STAGE(0);
            if constexpr (NSTG > 4) { RES_LOAD_STAGE(4, 0); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (NSTG > 3) RES_MMA_STAGE(1);
            if constexpr (NSTG > 5) { RES_LOAD_STAGE(5, 1); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (NSTG > 4) RES_MMA_STAGE(0);
            if constexpr (NSTG > 5) RES_MMA_STAGE(1);
#undef RES_LOAD_STAGE
#undef RES_MMA_STAGE
        } else
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) {
            const char* fb = filt + (int64_t)(cur.c * KSC + ks) * (ntaps * NT * 1024) + lane * 16;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                bf16x8 afr[6];
#pragma unroll
                for (int hr = 0; hr < 6; ++hr)
                    afr[hr] = *reinterpret_cast<const bf16x8*>(hb + aoff[ks][dx] + hr * (RES_HW * CW * 2));
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const char* ft = fb + a.tix[dy][dx] * (NT * 1024);
#pragma unroll
                    for (int t2 = 0; t2 < NT; ++t2) {
                        const bf16x8 wfr = *reinterpret_cast<const bf16x8*>(ft + t2 * 1024);
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            acc[m][t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr, afr[m + dy], acc[m][t2], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (SH) barrier_lds();                  // one halo buffer: every wave is done reading it before it is refilled
        if (have_next) RES_STASH(SH ? 0 : (buf ^ 1), nxt, preA, xsA);
        if (cur.c == nsteps - 1) {
            // ---- epilogue of this tile: lane (q, r) holds channels 4q..4q+3 of n-tile t2 at row 4*wid + m, column r
            const int mx = cur.x * 16 + r;
            bf16* const obase = a.out + (((int64_t)cur.n * a.g.ho + cur.y * 16 + wid * 4) * a.g.wo + mx) * a.g.ldo + slice * NT * 16 + 4 * q;
            const int64_t orstep = (int64_t)a.g.wo * a.g.ldo;
            const bool want_stats = a.stat_acc != nullptr;              // data-gradient launches carry no statistics
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int my = cur.y * 16 + wid * 4 + m;
                const bool inside = my < a.g.hm && mx < a.g.wm;
                bf16* orow = obase + m * orstep;
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2) {
                    const int cb = (slice * NT + t2) * 16 + 4 * q;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[m][t2][j] + bv[t2][j];
                    acc[m][t2] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (!inside || cb >= a.g.co) continue;
                    bf16* o = orow + t2 * 16;
                    if (cb + 3 < a.g.co) {
                        uint2 pk;
                        pk.x = pack_bf16x2(v[0], v[1]);
                        pk.y = pack_bf16x2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(o) = pk;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (cb + j < a.g.co) o[j] = (bf16)v[j];
                    }
                    if constexpr (BWD) {
                        const float y[4] = {__uint_as_float(bw_y[m][t2].x << 16), __uint_as_float(bw_y[m][t2].x & 0xffff0000u),
                                            __uint_as_float(bw_y[m][t2].y << 16), __uint_as_float(bw_y[m][t2].y & 0xffff0000u)};
                        const float mk[4] = {bw_mk[t2].x, bw_mk[t2].y, bw_mk[t2].z, bw_mk[t2].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {       // the arithmetic of k_bn_reduce<T, 1> on the rounded dA
                            const float rv = round_as<bf16>(v[j]);
                            const float z = y[j] * bw_sc[t2][j] + bw_sh[t2][j];
                            float dz = rv * mk[j];
                            if (!(z > 0.f)) dz = 0.f;
                            s1[t2][j] += dz;
                            s2[t2][j] += dz * (y[j] - bw_mean[t2][j]) * bw_inv[t2][j];
                        }
                    } else if (want_stats) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float rv = round_as<bf16>(v[j]);
                            s1[t2][j] += rv;
                            s2[t2][j] += rv * rv;
                        }
                    }
                }
            }
        }
        cur = nxt; RES_ADV(nxt); buf ^= 1;
        barrier_lds();     // NOT __syncthreads(): that would wait for the tile's output stores to reach memory
    }
#undef RES_ADV
#undef RES_FETCH
#undef RES_STASH

    // ---- fused BN statistics: one atomic add per (channel, sum) and block; slices cover disjoint channel ranges
    if (a.stat_acc) {
        float* red = reinterpret_cast<float*>(smem);     // [4 waves][NT][4 q][4 j][2]  (the filter is dead by now)
        __syncthreads();
#pragma unroll
        for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = s1[t2][j], y = s2[t2][j];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64); }
                if (r == 0) {
                    red[(((wid * NT + t2) * 4 + q) * 4 + j) * 2] = x;
                    red[(((wid * NT + t2) * 4 + q) * 4 + j) * 2 + 1] = y;
                }
            }
        __syncthreads();
        for (int i = tid; i < NT * 16 * 2; i += 256) {
            const int which = i / (NT * 16), c = i - which * NT * 16;
            const int t2 = c >> 4, qq = (c >> 2) & 3, j = c & 3;
            float acc_ = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) acc_ += red[(((w * NT + t2) * 4 + qq) * 4 + j) * 2 + which];
            const int chn = slice * NT * 16 + c - a.stat_c0;
            if (chn >= 0 && chn < a.cpad) bn_acc_add(a.stat_acc, bl, a.groups, a.cpad, grp, which, chn, acc_, which ? a.s2_scale : a.s1_scale);
        }
    }
}

static int conv_res_filter_budget() {
    static const int kb = [] { const char* e = getenv("STCD_CONV_RES_KB"); return e ? atoi(e) : 38; }();
    return kb;
}

ConvResPlan conv_res_plan(const stcd_conv_geom& g, const ConvMfmaPlan& p, int groups) {
    ConvResPlan rp;
    if (!p.ok || p.modeB || g.ntaps != 9 || g.in_stride != 1 || g.out_stride != 1 || g.oy0 != 0 || g.ox0 != 0) return rp;
    if (g.ci % 32 != 0 || g.ldi % 8 != 0 || g.ldo % 4 != 0 || g.hm > g.hi || g.wm > g.wi) return rp;
    if (groups < 1 || g.n % groups != 0) return rp;
    if (((int64_t)g.n * g.hi + 2) * g.wi * g.ldi * 2 >= ((int64_t)1 << 31)) return rp;     // 32-bit buffer offsets
    bool seen[9] = {false};
    for (int t = 0; t < 9; ++t) {
        if (g.dy[t] < -1 || g.dy[t] > 1 || g.dx[t] < -1 || g.dx[t] > 1) return rp;
        seen[(g.dy[t] + 1) * 3 + g.dx[t] + 1] = true;
    }
    for (int t = 0; t < 9; ++t) if (!seen[t]) return rp;
    const int nchunks = g.ci / 32;
    static const bool wide_ok = [] { const char* e = getenv("STCD_CONV_RES_NO_CW64"); return !(e && e[0] == '1'); }();
    int CW = 32, NT = 1;
    if (wide_ok && g.ci % 64 == 0 && g.ci >= 256) {
        // 64-channel steps, one block per CU (measured: only pays for the deepest layers, whose 32-channel form has
        // twice the steps; below that two resident blocks per CU win): widest co slice that fits beside the halos
        const int halo2 = 2 * RES_HW * RES_HW * 64 * 2;
        int nt = std::min(4, p.NT);
        while (nt > 1 && (nchunks * 9 * nt * 1024 + halo2 > 160 * 1024 - 1024 || p.NTtot % nt != 0)) nt >>= 1;
        if (nchunks * 9 * nt * 1024 + halo2 <= 160 * 1024 - 1024) { CW = 64; NT = nt; }
    }
    if (CW == 32) {
        // widest co slice that still leaves two blocks per CU; if even one n-tile cannot (deep layers), the widest that
        // fits at all -- a single resident block should at least not re-read X once per slice
        const int halo2 = 2 * RES_HW * RES_HW * 64, cap = 160 * 1024 - 1024;
        auto fits = [&](int nt, int blocks_per_cu) {
            return p.NTtot % nt == 0 && (nchunks * 9 * nt * 1024 + halo2) * blocks_per_cu <= cap;
        };
        const int ntmax = std::min(4, p.NT);
        NT = 0;
        // STCD_CONV_RES_ONE=<tiles>: layers with at most that many 16 x 16 tiles take the widest slice that fits ONE block per CU
        // (fewer, longer pipeline steps and fewer re-reads of the input per output slice) -- round-4 experiment, DESIGN.md section 4
        static const int one_tiles = [] { const char* e = getenv("STCD_CONV_RES_ONE"); return e ? atoi(e) : 0; }();
        {
            const int64_t tiles_ = (int64_t)g.n * ((g.wm + 15) / 16) * ((g.hm + 15) / 16);
            if (one_tiles > 0 && tiles_ <= one_tiles)
                for (int nt = ntmax; nt >= 1 && !NT; nt >>= 1)
                    if (fits(nt, 1)) NT = nt;
        }
        for (int nt = ntmax; nt >= 1 && !NT; nt >>= 1)
            if (fits(nt, 2) && nchunks * 9 * nt <= conv_res_filter_budget()) NT = nt;
        for (int nt = ntmax; nt >= 1 && !NT; nt >>= 1)
            if (fits(nt, 1)) NT = nt;
        if (!NT) return rp;
        // Layers whose filter slice leaves ONE block per CU either way (SNUNet's 160 ... 224 concatenated input channels -> 32 on
        // 256^2 maps): with a single halo buffer (the refill waits for a second barrier) the slice can be twice as wide, so the
        // 0.3 - 0.5 GB input is read once instead of once per 16 output channels.
        static const int sh_mode = [] { const char* e = getenv("STCD_CONV_RES_SH"); return e ? atoi(e) : 1; }();
        const int64_t tiles = (int64_t)g.n * ((g.wm + 15) / 16) * ((g.hm + 15) / 16);
        if (sh_mode && tiles >= 2048 && !fits(NT, 2)) {
            const int halo1 = RES_HW * RES_HW * 64;
            for (int nt = ntmax; nt > NT; nt >>= 1)
                if (p.NTtot % nt == 0 && nchunks * 9 * nt * 1024 + halo1 <= cap) { NT = nt; rp.single_halo = 1; break; }
        }
    }
    if (p.NTtot % NT != 0) return rp;
    rp.NT = NT; rp.CW = CW; rp.nslices = p.NTtot / NT;
    rp.filt_bytes = nchunks * 9 * NT * 1024;
    rp.lds_bytes = rp.filt_bytes + (rp.single_halo ? 1 : 2) * RES_HW * RES_HW * CW * 2;
    const int64_t tiles_x = (g.wm + 15) / 16, tiles_y = (g.hm + 15) / 16, ntiles = (int64_t)g.n * tiles_x * tiles_y;
    const int64_t tpg = ntiles / groups;
    const int per_cu = std::max(1, std::min(4, (160 * 1024) / (rp.lds_bytes + 256)));
    const int64_t slots = (int64_t)per_cu * 256;
    int64_t P = std::max<int64_t>(1, slots / ((int64_t)groups * rp.nslices));
    rp.P = (int)std::min<int64_t>(P, tpg);
    rp.blocks = rp.P * rp.nslices * groups;
    rp.ok = true;
    return rp;
}

bool conv_res_bwdsum_ok(const stcd_conv_geom& g, const ConvResPlan& rp) {
    return rp.ok && !rp.single_halo && rp.NT <= 2 && g.co % (rp.NT * 16) == 0 && g.out_stride == 1 && g.oy0 == 0 && g.ox0 == 0;
}
int launch_conv_res(const stcd_conv_geom& g, const ConvMfmaPlan& p, const ConvResPlan& rp, const void* in, const void* wf,
                    const float* bias, void* out, int groups, long long* stat_acc, int cpad, hipStream_t s, int stat_c0, float s1_scale,
                    float s2_scale, const XfSrc* xf, const BwdSum* bs) {
    if (!rp.ok) return 1;
    ConvResArgs a;
    a.bw_Y = nullptr; a.bw_ldy = 0; a.bw_stat = nullptr; a.bw_mask = nullptr;
    if (bs) {
        if (!conv_res_bwdsum_ok(g, rp) || stat_acc || (xf && xf->on) || bs->groups != groups || !bs->acc) return 1;
        a.bw_Y = (const bf16*)bs->Y; a.bw_ldy = bs->ldy; a.bw_stat = bs->stat; a.bw_mask = bs->mask;
        stat_acc = bs->acc; cpad = g.co; stat_c0 = 0; s1_scale = BN_BS; s2_scale = BN_BS;
    }
    a.g = g;
    a.in = (const bf16*)in; a.wf = (const bf16*)wf; a.bias = bias; a.out = (bf16*)out;
    a.NTtot = p.NTtot; a.KSp = p.CiB / 32; a.nchunks = g.ci / 32;
    a.nslices = rp.nslices; a.P = rp.P; a.groups = groups;
    a.tiles_x = (g.wm + 15) / 16; a.tiles_y = (g.hm + 15) / 16; a.ntiles = g.n * a.tiles_x * a.tiles_y;
    a.filt_bytes = rp.filt_bytes;
    a.stat_acc = stat_acc; a.cpad = cpad; a.stat_c0 = stat_c0; a.s1_scale = s1_scale; a.s2_scale = s2_scale;
    a.in_bytes = (unsigned)((int64_t)g.n * g.hi * g.wi * g.ldi * 2);
    for (int t = 0; t < 9; ++t) a.tix[g.dy[t] + 1][g.dx[t] + 1] = (int8_t)t;
    const bool use_xf = xf && xf->on;
    size_t lds = (size_t)rp.lds_bytes;
    if (use_xf) {
        if (rp.single_halo || xf->C != g.ci || xf->groups != groups) return 1;     // (the single-halo variant has no XF instantiation)
        a.xf = *xf;
        lds += (size_t)xf->groups * 2 * xf->C * 4;
        if (lds > 160 * 1024) return 1;
    }
#define LAUNCH_RES_V(N_, W_, P_, X_)                                                                              \
    do {                                                                                                          \
        static bool attr_set = false;                                                                             \
        if (!attr_set) {                                                                                          \
            (void)hipFuncSetAttribute((const void*)k_conv_res<N_, W_, false, P_, X_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                      \
        }                                                                                                         \
        k_conv_res<N_, W_, false, P_, X_><<<(unsigned)rp.blocks, 256, lds, s>>>(a);                               \
    } while (0)
#define LAUNCH_RES(N_, W_) do { if (use_xf) LAUNCH_RES_V(N_, W_, false, true); else LAUNCH_RES_V(N_, W_, false, false); } while (0)
#define LAUNCH_RES_PIPE(N_, W_) do { if (use_xf) LAUNCH_RES_V(N_, W_, true, true); else LAUNCH_RES_V(N_, W_, true, false); } while (0)
    // measured (SNUNet / SegCD, 16 x 256^2): <1, 64> 0.67 -> 0.58 ms per step (-13 %); the CW = 32 variants do not move (their
    // layers sit on the HBM roofline: 134 MB in + 134 MB out per 32 -> 32 full-resolution layer in 66 us), <2, 64> has no registers
    // left for the second fragment set.  Default: <1, 64> only; STCD_CONV_RES_PIPE=1 pipelines every variant that fits, 0 none.
    static const int pipe_env = [] { const char* e = getenv("STCD_CONV_RES_PIPE"); return e ? atoi(e) : -1; }();
    const int pipe = pipe_env >= 0 ? pipe_env : (rp.CW == 64 && rp.NT == 1 && !use_xf);      // (<1, 64, PIPE, XF> spills: 112 B / lane)
#define LAUNCH_RES_SH(N_)                                                                                         \
    do {                                                                                                          \
        static bool attr_set = false;                                                                             \
        if (!attr_set) {                                                                                          \
            (void)hipFuncSetAttribute((const void*)k_conv_res<N_, 32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                      \
        }                                                                                                         \
        k_conv_res<N_, 32, true><<<(unsigned)rp.blocks, 256, (size_t)rp.lds_bytes, s>>>(a);                       \
    } while (0)
    if (bs) {
#define LAUNCH_RES_BWD(N_, W_, P_)                                                                                \
    do {                                                                                                          \
        static bool attr_set = false;                                                                             \
        if (!attr_set) {                                                                                          \
            (void)hipFuncSetAttribute((const void*)k_conv_res<N_, W_, false, P_, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                      \
        }                                                                                                         \
        k_conv_res<N_, W_, false, P_, false, true><<<(unsigned)rp.blocks, 256, lds, s>>>(a);                      \
    } while (0)
        if (rp.CW == 64) { if (rp.NT == 1) LAUNCH_RES_BWD(1, 64, false); else LAUNCH_RES_BWD(2, 64, false); }
        else { if (rp.NT == 1) LAUNCH_RES_BWD(1, 32, false); else LAUNCH_RES_BWD(2, 32, false); }
#undef LAUNCH_RES_BWD
    } else if (rp.single_halo) {
        if (rp.NT == 2) LAUNCH_RES_SH(2); else LAUNCH_RES_SH(4);
    } else if (rp.CW == 64) {
        switch (rp.NT) {
            case 1: if (pipe) LAUNCH_RES_PIPE(1, 64); else LAUNCH_RES(1, 64); break;
            case 2: if (pipe) LAUNCH_RES_PIPE(2, 64); else LAUNCH_RES(2, 64); break;
            default: LAUNCH_RES(4, 64); break;
        }
    } else {
        switch (rp.NT) {
            case 1: if (pipe) LAUNCH_RES_PIPE(1, 32); else LAUNCH_RES(1, 32); break;
            case 2: if (pipe) LAUNCH_RES_PIPE(2, 32); else LAUNCH_RES(2, 32); break;
            default: LAUNCH_RES(4, 32); break;       // NT = 4 has no registers for a second fragment set (it spills)
        }
    }
#undef LAUNCH_RES
#undef LAUNCH_RES_PIPE
#undef LAUNCH_RES_V
#undef LAUNCH_RES_SH
    return 0;
}


// =====================================================================================================
// Tap-list convolution as a GEMM  out[m][co] = sum_{t, ci} X[pixel(m) * stride + tap_t][ci] * W[t][ci][co]  over m = the
// positions of one BN group: 1x1 convolutions (one tap: a plain GEMM), and every wide (Ci % 64 == 0, Co % 64 == 0) layer the
// resident-filter kernel cannot take -- 3x3 with a filter slice beyond LDS (UNet decoder 3072 -> 256), stride-2 3x3, the data
// gradient of 2x2 stride-2 transposed convs -- where K simply walks over (64-channel chunk, tap).
//
// Block tile TM x TN = (32*W) x (32*W) (W = 4: 128x128, W = 2: 64x64), 4 waves as 2 (positions) x 2 (channels), each
// W x W fragments of 16x16; K walks in 64-channel steps: X rows (128 contiguous bytes each, raw buffer loads with
// hardware zero-fill past the group's last position) and the weight fragments (already in MFMA order, contiguous
// 1-KB pieces) are requested into registers before the MFMAs of the current step and parked in the other LDS buffer
// after them (one barrier per step).  X image: row pitch 128 B, 16-B chunk index XOR-swizzled by (row >> 1) & 7 (the
// layout of k_conv_res's 64-channel halo).  Weights are the MFMA A operand, so a lane's accumulator is 4 consecutive
// channels of one position; the finished tile goes through LDS once more and leaves as full 16-B pieces of complete
// output rows.  A stride (1x1 stride-2 down-sampling convs and their data gradient) only changes the row -> pixel maps.
struct ConvGemmArgs {
    stcd_conv_geom g;
    const bf16* in; const bf16* wf; const float* bias; bf16* out;
    long long* stat_acc;
    int groups, cpad, stat_c0;
    float s1_scale, s2_scale;
    int NTtot, nsteps;             // n-tiles of the fragment image; (Ci / 64) * ntaps
    unsigned lead;                 // bytes the X descriptor starts before the tensor (most negative tap offset)
    int pix_chunks;                // 1: 16-B chunk ch of a row is the pixel ch to the right (stem), bounds checked per chunk
    int Mg;                        // positions per group
    int tiles_m, tiles_m8, tiles_n;
    unsigned in_bytes;
    int tap[9];                    // (dy << 16) | (dx & 0xffff) per tap, as 32-bit words: a uniform index into them is a SCALAR
                                   // load (the int8 arrays of g compile to global_load_sbyte, whose vmcnt(0) wait also waits for
                                   // every fetch still in flight -- that had serialised the two register sets)
    // fused epilogue (ConvEpi, common.h)
    int relu;
    const bf16* gate; int ldg;
    const bf16* res; int ldr;
    float alpha, beta;
};

template <int W>
__global__ void __launch_bounds__(256, W == 2 ? 4 : 2)
k_conv_gemm(const ConvGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TM = 32 * W, TN = 32 * W, NF = TN / 16;
    constexpr int XB = TM * 128, WB = 2 * NF * 1024, STAGE = XB + WB;
    constexpr int XP = TM / 32, WP = (2 * NF * 64) / 256;           // 16-B pieces per thread and stage
    constexpr int OPITCH = TN * 2 + 16;                             // out-tile row pitch (bytes)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int wm = wid & 1, wn = wid >> 1;
    // block -> (group, m tile, n tile): the n tiles of one m tile are 8 blocks apart, i.e. on the same XCD (one L2 fetch of X)
    const int per_grp = a.tiles_m8 * 8 * a.tiles_n;
    const int grp = blockIdx.x / per_grp, idx = blockIdx.x - grp * per_grp;
    const int m8 = idx / (8 * a.tiles_n), rem = idx - m8 * 8 * a.tiles_n;
    const int nt = rem >> 3, mt = m8 * 8 + (rem & 7);
    if (mt >= a.tiles_m) return;
    const int m0 = mt * TM;                                         // first position of the tile inside the group
    const int nf0 = nt * NF;                                        // first n-fragment

    // ---- staging plan
    const int ch = tid & 7;
    const bool plain_in = a.g.ntaps == 1 && a.g.dy[0] == 0 && a.g.dx[0] == 0 && a.g.in_stride == 1 && a.g.hm == a.g.hi && a.g.wm == a.g.wi && !a.pix_chunks;
    const bool plain_out = a.g.out_stride == 1 && a.g.ho == a.g.hm && a.g.wo == a.g.wm && a.g.oy0 == 0 && a.g.ox0 == 0;
    unsigned xoff[XP]; int xlds[XP], xyx[XP];          // byte offset of the row's centre pixel (+ lead), packed (y, x) of it
#pragma unroll
    for (int p = 0; p < XP; ++p) {
        const int row = (tid >> 3) + p * 32, m = m0 + row;
        unsigned off = 0x80000000u;
        int yx = (int)0xC000C000;                      // far outside every image: all taps of a masked row read zeros
        if (m < a.Mg) {
            const int mm = grp * a.Mg + m;
            if (plain_in) {          // 1x1, stride 1: the position IS the pixel (no integer divisions in the prologue of a 1-4 step block)
                off = (unsigned)(((int64_t)mm * a.g.ldi + ch * 8) * 2) + a.lead;
                yx = 0;
            } else {
                const int x = mm % a.g.wm, t = mm / a.g.wm, y = t % a.g.hm, n = t / a.g.hm;
                off = (unsigned)(((((int64_t)n * a.g.hi + y * a.g.in_stride) * a.g.wi + x * a.g.in_stride) * a.g.ldi + ch * 8) * 2) + a.lead;
                yx = ((y * a.g.in_stride) << 16) | (x * a.g.in_stride);
            }
        }
        xoff[p] = off; xyx[p] = yx;
        xlds[p] = (row * 8 + (ch ^ ((row >> 1) & 7))) * 16;
    }
    unsigned woff[WP]; int wlds[WP];
#pragma unroll
    for (int p = 0; p < WP; ++p) {
        const int i = tid + p * 256, f = (i >> 6) % NF, ks = (i >> 6) / NF;
        woff[p] = (unsigned)(((ks * a.NTtot + nf0 + f) * 64 + (i & 63)) * 16);          // + chunk * 2 * NTtot * 1024 per step
        wlds[p] = XB + ((ks * NF + f) * 64 + (i & 63)) * 16;
    }
    // the X descriptor starts `lead` bytes before the tensor, so a tap's (possibly negative) pixel offset never wraps
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.in)) - a.lead, (short)0, (int)(a.in_bytes + a.lead), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.wf)), (short)0, a.nsteps * 2 * a.NTtot * 1024, 0x00020000);
    const unsigned wstep = (unsigned)(2 * a.NTtot * 1024);
    const int ntaps = a.g.ntaps;
    const int chx = a.pix_chunks ? ch : 0;
    // two register sets: the chunk after next is requested before the MFMAs of the current one, so every load has two
    // loop iterations to land (one block per CU on the deep layers: nothing else hides the L2 / HBM round trip)
    uint4 pxa[XP], pwa[WP], pxb[XP], pwb[WP];
    // step s = (chunk, tap), tap fastest -- the order of the fragment image [chunk][tap][ks][n-tile]
#define GM_FETCH(S_, PX_, PW_)                                                                                         \
    do {                                                                                                               \
        const int ss_ = __builtin_amdgcn_readfirstlane(S_);                     /* scalar step index */                \
        const int cc_ = ss_ / ntaps, tt_ = ss_ - cc_ * ntaps;                                                          \
        const int tw_ = a.tap[tt_];                                                                                    \
        const int tdy_ = tw_ >> 16, tdx_ = (int)(short)(tw_ & 0xffff);                                                 \
        const int toff_ = (tdy_ * a.g.wi + tdx_) * a.g.ldi * 2;                                                        \
        _Pragma("unroll") for (int p = 0; p < XP; ++p) {                                                               \
            const bool ok_ = (unsigned)((xyx[p] >> 16) + tdy_) < (unsigned)a.g.hi && (unsigned)((int)(short)(xyx[p] & 0xffff) + tdx_ + chx) < (unsigned)a.g.wi; \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok_ ? xoff[p] + (unsigned)toff_ : 0x80000000u, (unsigned)cc_ * 128u, 0); \
            PX_[p] = make_uint4(v_[0], v_[1], v_[2], v_[3]);                                                           \
        }                                                                                                              \
        _Pragma("unroll") for (int p = 0; p < WP; ++p) {                                                               \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(wrs, woff[p], (unsigned)ss_ * wstep, 0);            \
            PW_[p] = make_uint4(v_[0], v_[1], v_[2], v_[3]);                                                           \
        }                                                                                                              \
    } while (0)
#define GM_STASH(BUF_, PX_, PW_)                                                                                       \
    do {                                                                                                               \
        _Pragma("unroll") for (int p = 0; p < XP; ++p) *reinterpret_cast<uint4*>(smem + (BUF_) * STAGE + xlds[p]) = PX_[p]; \
        _Pragma("unroll") for (int p = 0; p < WP; ++p) *reinterpret_cast<uint4*>(smem + (BUF_) * STAGE + wlds[p]) = PW_[p]; \
    } while (0)
#define GM_COMPUTE(BUF_)                                                                                               \
    do {                                                                                                               \
        const char* sb = smem + (BUF_) * STAGE;                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                             \
            bf16x8 xf[W], wfr[W];                                                                                      \
            _Pragma("unroll") for (int m = 0; m < W; ++m) xf[m] = *reinterpret_cast<const bf16x8*>(sb + xr[m][ks]);    \
            _Pragma("unroll") for (int n = 0; n < W; ++n) wfr[n] = *reinterpret_cast<const bf16x8*>(sb + wr + (ks * NF + n) * 1024); \
            _Pragma("unroll") for (int m = 0; m < W; ++m)                                                              \
                _Pragma("unroll") for (int n = 0; n < W; ++n)                                                          \
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[n], xf[m], acc[m][n], 0, 0, 0);            \
        }                                                                                                              \
    } while (0)

    f32x4 acc[W][W];
#pragma unroll
    for (int m = 0; m < W; ++m)
#pragma unroll
        for (int n = 0; n < W; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    int xr[W][2];                       // X fragment offsets: row wm*16W + 16m + r, chunk 4ks + q
#pragma unroll
    for (int m = 0; m < W; ++m)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = wm * 16 * W + 16 * m + r;
            xr[m][ks] = (row * 8 + ((ks * 4 + q) ^ ((row >> 1) & 7))) * 16;
        }
    const int wr = XB + (wn * W * 64 + lane) * 16;

    // every fetch / stash is unconditional (the chunk index is clamped: the tail re-reads the last chunk into a buffer nobody
    // reads), so both register sets stay in registers across the loop
    const int nc = a.nsteps;
    GM_FETCH(0, pxa, pwa);
    GM_FETCH(min(1, nc - 1), pxb, pwb);
    GM_STASH(0, pxa, pwa);
    __syncthreads();
    // sched_barrier: without it the compiler sinks the fetch below the MFMAs (right in front of the stash of the OTHER set),
    // which (a) gives the loads half an iteration instead of one and a half to land and (b) lets it reuse the fetch registers
    // for the LDS fragment reads of the same half -- every ds_read then waits for the outstanding global loads (vmcnt) first
    for (int c = 0; c < nc; c += 2) {
        GM_FETCH(min(c + 2, nc - 1), pxa, pwa);
        __builtin_amdgcn_sched_barrier(0);
        GM_COMPUTE(0);
        GM_STASH(1, pxb, pwb);
        barrier_lds();
        if (c + 1 >= nc) break;
        GM_FETCH(min(c + 3, nc - 1), pxb, pwb);
        __builtin_amdgcn_sched_barrier(0);
        GM_COMPUTE(1);
        GM_STASH(0, pxa, pwa);
        barrier_lds();
    }
#undef GM_COMPUTE
#undef GM_FETCH
#undef GM_STASH

    // ---- epilogue: bias, rounding, statistics of the rounded values, out tile -> LDS [TM][TN] (pitch OPITCH)
    char* const ot = smem;                                           // both stage buffers are dead (barrier above)
    float s1[W][4], s2[W][4];
#pragma unroll
    for (int n = 0; n < W; ++n) {
        const int cb = (nf0 + wn * W + n) * 16 + 4 * q;
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { bv[j] = (a.bias && cb + j < a.g.co) ? a.bias[cb + j] : 0.f; s1[n][j] = s2[n][j] = 0.f; }
#pragma unroll
        for (int m = 0; m < W; ++m) {
            const int row = wm * 16 * W + 16 * m + r;
            const bool valid = m0 + row < a.Mg;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[m][n][j] + bv[j];
                if (a.relu) v[j] = fmaxf(v[j], 0.f);
                const float rv = round_as<bf16>(v[j]);
                if (valid) { s1[n][j] += rv; s2[n][j] += rv * rv; }
            }
            uint2 pk;
            pk.x = pack_bf16x2(v[0], v[1]);
            pk.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(ot + row * OPITCH + ((wn * W + n) * 16 + 4 * q) * 2) = pk;
        }
    }
    float* const red = reinterpret_cast<float*>(smem + TM * OPITCH);  // [2 wm][2 wn][W][4 q][4 j][2]
    if (a.stat_acc) {
#pragma unroll
        for (int n = 0; n < W; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = s1[n][j], y = s2[n][j];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64); }
                if (r == 0) {
                    red[((((wm * 2 + wn) * W + n) * 4 + q) * 4 + j) * 2] = x;
                    red[((((wm * 2 + wn) * W + n) * 4 + q) * 4 + j) * 2 + 1] = y;
                }
            }
    }
    __syncthreads();
    // ---- complete output rows leave as 16-B pieces (TN / 8 per row)
    constexpr int OP = (TM * (TN / 8)) / 256;
#pragma unroll
    for (int p = 0; p < OP; ++p) {
        const int i = tid + p * 256, row = i / (TN / 8), c8 = i - row * (TN / 8), m = m0 + row;
        if (m < a.Mg) {
            const int mm = grp * a.Mg + m;
            int64_t opix = mm;
            if (!plain_out) {
                const int x = mm % a.g.wm, t = mm / a.g.wm, y = t % a.g.hm, n = t / a.g.hm;
                opix = ((int64_t)n * a.g.ho + y * a.g.out_stride + a.g.oy0) * a.g.wo + x * a.g.out_stride + a.g.ox0;
            }
            uint4 pk = *reinterpret_cast<const uint4*>(ot + row * OPITCH + c8 * 16);
            if (a.gate || a.res) {      // the rounded conv output, gated and / or combined with a residual, rounded once more
                float v8[8];
                const uint32_t wv[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) { v8[2 * i2] = __uint_as_float(wv[i2] << 16); v8[2 * i2 + 1] = __uint_as_float(wv[i2] & 0xffff0000u); }
                if (a.gate) {
                    float g8[8];
                    load8(a.gate + opix * a.ldg + nt * TN + c8 * 8, g8);
#pragma unroll
                    for (int i2 = 0; i2 < 8; ++i2) v8[i2] = g8[i2] > 0.f ? v8[i2] : 0.f;
                }
                if (a.res) {
                    float r8[8];
                    load8(a.res + opix * a.ldr + nt * TN + c8 * 8, r8);
#pragma unroll
                    for (int i2 = 0; i2 < 8; ++i2) v8[i2] = a.alpha * v8[i2] + a.beta * r8[i2];
                }
                pk.x = pack_bf16x2(v8[0], v8[1]); pk.y = pack_bf16x2(v8[2], v8[3]);
                pk.z = pack_bf16x2(v8[4], v8[5]); pk.w = pack_bf16x2(v8[6], v8[7]);
            }
            *reinterpret_cast<uint4*>(a.out + opix * a.g.ldo + nt * TN + c8 * 8) = pk;
        }
    }
    if (a.stat_acc) {
        for (int i = tid; i < TN * 2; i += 256) {
            const int which = i / TN, c = i - which * TN;             // channel c of the tile: wn = c / (16W), n, q, j
            const int wn_ = c / (16 * W), n = (c >> 4) % W, qq = (c >> 2) & 3, j = c & 3;
            const float v = red[((((0 * 2 + wn_) * W + n) * 4 + qq) * 4 + j) * 2 + which] + red[((((1 * 2 + wn_) * W + n) * 4 + qq) * 4 + j) * 2 + which];
            const int chn = nt * TN + c - a.stat_c0;
            if (chn >= 0 && chn < a.cpad) bn_acc_add(a.stat_acc, mt, a.groups, a.cpad, grp, which, chn, v, which ? a.s2_scale : a.s1_scale);
        }
    }
}

ConvGemmPlan conv_gemm_plan(const stcd_conv_geom& g, const ConvMfmaPlan& p, int groups) {
    ConvGemmPlan gp;
    if (!p.ok || p.modeB || p.CiB != 64 || g.ntaps < 1) return gp;
    for (int t = 0; t < g.ntaps; ++t)
        if (g.dy[t] < -3 || g.dy[t] > 3 || g.dx[t] < -3 || g.dx[t] > 3) return gp;
    if (g.ci % 64 != 0 || g.co % 64 != 0 || g.ldi % 8 != 0 || g.ldo % 8 != 0 || groups < 1 || g.n % groups != 0) return gp;
    if (((int64_t)g.n * g.hi + 8) * g.wi * g.ldi * 2 >= ((int64_t)1 << 31) || g.hi >= 16384 || g.wi >= 16384) return gp;
    if ((g.hm - 1) * g.in_stride >= g.hi + 2 || (g.wm - 1) * g.in_stride >= g.wi + 2) return gp;
    const int64_t Mg = (int64_t)(g.n / groups) * g.hm * g.wm;
    if (Mg * groups >= ((int64_t)1 << 31)) return gp;
    // the 128x128 tile when it still fills the chip (>= 192 blocks), else 64x64
    int W = 4;
    if (g.co % 128 != 0 || p.NTtot % 8 != 0 || (int64_t)groups * ((Mg + 127) / 128) * (g.co / 128) < 192) W = 2;
    if (p.NTtot % (2 * W) != 0) return gp;
    const int T = 32 * W;
    gp.W = W;
    gp.tiles_m = (int)((Mg + T - 1) / T);
    gp.tiles_n = g.co / T;
    gp.blocks = groups * ((gp.tiles_m + 7) / 8) * 8 * gp.tiles_n;
    gp.lds_bytes = std::max(2 * (T * 128 + 2 * (T / 16) * 1024), T * (T * 2 + 16) + 2 * 2 * W * 16 * 2 * 4);
    gp.ok = true;
    return gp;
}

int launch_conv_gemm(const stcd_conv_geom& g, const ConvMfmaPlan& p, const ConvGemmPlan& gp, const void* in, const void* wf,
                   const float* bias, void* out, int groups, long long* stat_acc, int cpad, hipStream_t s, int stat_c0,
                   float s1_scale, float s2_scale, const ConvEpi* epi) {
    if (!gp.ok) return 1;
    ConvGemmArgs a;
    a.g = g;
    a.relu = epi ? epi->relu : 0;
    a.gate = epi ? (const bf16*)epi->gate : nullptr; a.ldg = epi ? epi->ldg : 0;
    a.res = epi ? (const bf16*)epi->res : nullptr; a.ldr = epi ? epi->ldr : 0;
    a.alpha = epi ? epi->alpha : 1.f; a.beta = epi ? epi->beta : 1.f;
    a.in = (const bf16*)in; a.wf = (const bf16*)wf; a.bias = bias; a.out = (bf16*)out;
    a.stat_acc = stat_acc; a.groups = groups; a.cpad = cpad; a.stat_c0 = stat_c0; a.s1_scale = s1_scale; a.s2_scale = s2_scale;
    a.NTtot = p.NTtot; a.nsteps = (g.ci / 64) * g.ntaps;
    a.lead = (unsigned)((3 * g.wi + 3) * g.ldi * 2);
    a.pix_chunks = gp.pix_chunks;
    a.Mg = (g.n / groups) * g.hm * g.wm;
    a.tiles_m = gp.tiles_m; a.tiles_m8 = (gp.tiles_m + 7) / 8; a.tiles_n = gp.tiles_n;
    a.in_bytes = (unsigned)((int64_t)g.n * g.hi * g.wi * g.ldi * 2);
    fill_taps(g, a.tap);
    if (gp.W == 4) {
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_conv_gemm<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
        k_conv_gemm<4><<<(unsigned)gp.blocks, 256, (size_t)gp.lds_bytes, s>>>(a);
    } else {
        k_conv_gemm<2><<<(unsigned)gp.blocks, 256, (size_t)gp.lds_bytes, s>>>(a);
    }
    return 0;
}


// =====================================================================================================
// One-tap weight gradient as a GEMM over positions:  dW[ci][co] = sum_m X[pixel_in(m)][ci] * dY[pixel_out(m)][co].
// Block tile T x T channels (T = 32*W: 128 or 64), 4 waves as 2 (ci) x 2 (co), each W x W accumulators; the positions are
// cut into gx contiguous slices (one block and one fp32 slab each), walked in 64-position chunks: both operand rows (T
// channels = 2T contiguous bytes) go global -> registers -> LDS with two chunks in flight, as 32-channel sub-images in the
// padded layout of k_wgrad_group (pixel p at p*64 + (p>>3)*32), and come back channel-major through ds_read_b64_tr_b16.
// Any stride / phase is only a row -> pixel map (2x2 stride-2 transposed convs, stride-2 1x1 convs).
template <int W>
__global__ void __launch_bounds__(256, 2)
k_wgrad_gemm(const WgradJob* __restrict__ jobs, int njobs, const char* base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = 32 * W, NS = T / 32, SUB = 64 * 64 + 8 * 32, IMG = NS * SUB, STAGE = 2 * IMG;
    constexpr int PP = (64 * (T / 8)) / 256;                       // 16-B pieces per thread, operand and chunk
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const WgradJob& a = jobs[lo];
    const int lb = blockIdx.x - a.start;
    // blocks of one position slice (all channel tiles) are 8 ids apart: same XCD, the slice's rows are fetched into its L2 once
    const int G = a.gy * a.gz, full = (a.gx >> 3) << 3;
    int bx, r;
    if (lb < full * G) { const int rem = lb % (8 * G); bx = (lb / (8 * G)) * 8 + (rem & 7); r = rem >> 3; }
    else { const int l2 = lb - full * G, tail = a.gx - full; bx = full + l2 % tail; r = l2 / tail; }
    const int by = r % a.gy, bz = r / a.gy;
    const int ci0 = by * T, co0 = bz * T;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wa = wid & 1, wb = wid >> 1;
    const int grp = lane >> 4, li = lane & 15, qrow = li >> 2, pcol = li & 3;
    const int M = a.ntiles;                                         // positions of the job
    const int nchunks = (M + 63) >> 6;
    const int cpb = (nchunks + a.gx - 1) / a.gx;
    const int c0 = bx * cpb, c1 = min(nchunks, c0 + cpb);

    // staging plan: piece i = tid + p*256 -> position i / (T/8) of the chunk, 16-B chunk c8 = i % (T/8).  256 is a multiple of
    // T/8, so c8 is the same for all of a thread's pieces and its rows are RP = 256/(T/8) apart: scalars, not arrays
    constexpr int RP = 256 / (T / 8), LSTEP = RP * 64 + (RP / 8) * 32;
    const int prow0 = tid / (T / 8), c8 = tid % (T / 8);
    const int plds0 = (c8 >> 2) * SUB + prow0 * 64 + (prow0 >> 3) * 32 + (c8 & 3) * 16;
    const unsigned xch = (ci0 + c8 * 8 < a.g.ci) ? (unsigned)((ci0 + c8 * 8) * 2) : 0x80000000u;
    const unsigned ych = (co0 + c8 * 8 < a.co_valid) ? (unsigned)((co0 + c8 * 8) * 2) : 0x80000000u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(base + a.in_off), (short)0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(base + a.dout_off), (short)0, (int)a.dout_bytes, 0x00020000);
    const int tdy = a.g.dy[0], tdx = a.g.dx[0];
    uint4 xa[PP], ya[PP], xb[PP], yb[PP];
    // plain layout (the common 1x1 case): position m IS the pixel index of both tensors -- no (n, y, x) decomposition per piece
    // (two integer divisions per piece and chunk made the loader, not the MFMAs, the bottleneck: ~200 VALU beside 32 MFMAs).
    // Strided / phase maps keep a running (n, y, x) per piece, advanced by 64 positions per chunk with carries.
    const bool plain = a.g.in_stride == 1 && a.g.out_stride == 1 && a.g.hm == a.g.hi && a.g.wm == a.g.wi && a.g.ho == a.g.hm && a.g.wo == a.g.wm &&
                       a.g.oy0 == 0 && a.g.ox0 == 0 && tdy == 0 && tdx == 0 && a.wi_valid == a.g.wi;
    constexpr bool INCR = W == 2;                       // (W == 4 has no registers to spare for the state: it divides per chunk)
    int px_[PP], py_[PP], pn_[PP];                      // (x, y, n) of piece p's position in chunk `pos_chunk`
    int pos_chunk = c0;
    const int adv_x = 64 % a.g.wm, adv_t = 64 / a.g.wm, adv_y = adv_t % a.g.hm, adv_n = adv_t / a.g.hm;
    if (INCR && !plain) {
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            const int m_ = c0 * 64 + prow0 + p * RP;
            px_[p] = m_ % a.g.wm; const int t_ = m_ / a.g.wm; py_[p] = t_ % a.g.hm; pn_[p] = t_ / a.g.hm;
        }
    }
#define WGG_FETCH(C_, PX_, PY_)                                                                                        \
    do {                                                                                                               \
        const int cc_ = (C_);                                                                                          \
        if (INCR && !plain) {                                                                                          \
            while (pos_chunk < cc_) {                       /* uniform: chunks are requested in non-decreasing order */ \
                _Pragma("unroll") for (int p = 0; p < PP; ++p) {                                                       \
                    px_[p] += adv_x; py_[p] += adv_y; pn_[p] += adv_n;                                                 \
                    if (px_[p] >= a.g.wm) { px_[p] -= a.g.wm; ++py_[p]; }                                              \
                    if (py_[p] >= a.g.hm) { py_[p] -= a.g.hm; ++pn_[p]; }                                              \
                }                                                                                                      \
                ++pos_chunk;                                                                                           \
            }                                                                                                          \
        }                                                                                                              \
        _Pragma("unroll") for (int p = 0; p < PP; ++p) {                                                               \
            const int m_ = cc_ * 64 + prow0 + p * RP;                                                                  \
            bool okx_; unsigned xo_, yo_;                                                                              \
            if (plain) {                                                                                               \
                okx_ = m_ < M; xo_ = (unsigned)(m_ * a.g.ldi * 2); yo_ = (unsigned)(m_ * a.g.ldo * 2);                 \
            } else {                                                                                                   \
                int x_, y_, n_;                                                                                        \
                if (INCR) { x_ = px_[p]; y_ = py_[p]; n_ = pn_[p]; }                                                   \
                else { x_ = m_ % a.g.wm; const int t_ = m_ / a.g.wm; y_ = t_ % a.g.hm; n_ = t_ / a.g.hm; }            \
                const int iy_ = y_ * a.g.in_stride + tdy, ix_ = x_ * a.g.in_stride + tdx;                              \
                okx_ = m_ < M && (unsigned)iy_ < (unsigned)a.g.hi && (unsigned)ix_ < (unsigned)a.wi_valid;             \
                xo_ = (unsigned)((((n_ * a.g.hi + iy_) * a.g.wi + ix_) * a.g.ldi) * 2);                                \
                yo_ = (unsigned)((((n_ * a.g.ho + y_ * a.g.out_stride + a.g.oy0) * a.g.wo + x_ * a.g.out_stride + a.g.ox0) * a.g.ldo) * 2); \
            }                                                                                                          \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(xrs, okx_ ? xo_ + xch : 0x80000000u, 0, 0);         \
            const u32x4 w_ = __builtin_amdgcn_raw_buffer_load_b128(yrs, m_ < M ? yo_ + ych : 0x80000000u, 0, 0);       \
            PX_[p] = make_uint4(v_[0], v_[1], v_[2], v_[3]);                                                           \
            PY_[p] = make_uint4(w_[0], w_[1], w_[2], w_[3]);                                                           \
        }                                                                                                              \
    } while (0)
#define WGG_STASH(BUF_, PX_, PY_)                                                                                      \
    do {                                                                                                               \
        _Pragma("unroll") for (int p = 0; p < PP; ++p) {                                                               \
            *reinterpret_cast<uint4*>(smem + (BUF_) * STAGE + plds0 + p * LSTEP) = PX_[p];                             \
            *reinterpret_cast<uint4*>(smem + (BUF_) * STAGE + IMG + plds0 + p * LSTEP) = PY_[p];                       \
        }                                                                                                              \
    } while (0)

    f32x4 acc[W][W];
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j = 0; j < W; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment offsets inside an image: k-step ks (32 positions), half h: position 32ks + 16(grp>>1) + 8(grp&1) + 4h + qrow
    int fo[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pp = 32 * ks + 16 * (grp >> 1) + 8 * (grp & 1) + 4 * h + qrow;
            fo[ks][h] = pp * 64 + (pp >> 3) * 32 + 8 * pcol;
        }
#define WGG_COMPUTE(BUF_)                                                                                              \
    do {                                                                                                               \
        const char* xs_ = smem + (BUF_) * STAGE;                                                                       \
        const char* ys_ = xs_ + IMG;                                                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                             \
            bf16x8 af[W], bfr[W];                                                                                      \
            _Pragma("unroll") for (int i = 0; i < W; ++i) {                                                            \
                const int t_ = wa * W + i;                                                                             \
                af[i] = tr_frag(xs_ + (t_ >> 1) * SUB + (t_ & 1) * 32 + fo[ks][0], xs_ + (t_ >> 1) * SUB + (t_ & 1) * 32 + fo[ks][1]); \
            }                                                                                                          \
            _Pragma("unroll") for (int j = 0; j < W; ++j) {                                                            \
                const int t_ = wb * W + j;                                                                             \
                bfr[j] = tr_frag(ys_ + (t_ >> 1) * SUB + (t_ & 1) * 32 + fo[ks][0], ys_ + (t_ >> 1) * SUB + (t_ & 1) * 32 + fo[ks][1]); \
            }                                                                                                          \
            _Pragma("unroll") for (int i = 0; i < W; ++i)                                                              \
                _Pragma("unroll") for (int j = 0; j < W; ++j)                                                          \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);            \
        }                                                                                                              \
    } while (0)

    if (c0 < c1) {
        const int last = c1 - 1;
        WGG_FETCH(c0, xa, ya);
        WGG_FETCH(min(c0 + 1, last), xb, yb);
        WGG_STASH(0, xa, ya);
        __syncthreads();
        for (int c = c0; c < c1; c += 2) {
            WGG_FETCH(min(c + 2, last), xa, ya);
            WGG_COMPUTE(0);
            WGG_STASH(1, xb, yb);
            barrier_lds();
            if (c + 1 >= c1) break;
            WGG_FETCH(min(c + 3, last), xb, yb);
            WGG_COMPUTE(1);
            WGG_STASH(0, xa, ya);
            barrier_lds();
        }
    }
#undef WGG_FETCH
#undef WGG_STASH
#undef WGG_COMPUTE
    // ---- slab[bx][ci][co]: lane holds D[ci_local = 4*grp + j][co_local = li] of every accumulator
    float* slab = reinterpret_cast<float*>(const_cast<char*>(base) + a.slab_off) + (int64_t)bx * a.kpad * a.wld;
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = ci0 + (wa * W + i) * 16 + 4 * grp + e, co = co0 + (wb * W + j) * 16 + li;
                if (ci < a.kpad && co < a.wld) slab[(int64_t)ci * a.wld + co] = acc[i][j][e];
            }
}

WgradMfmaPlan wgrad_gemm_plan(const stcd_conv_geom& g, int kpad, int wld) {
    WgradMfmaPlan p;
    if (g.ntaps != 1 || g.ci < 64 || g.co < 64 || g.ci % 8 != 0 || g.ldi % 8 != 0 || g.ldo % 8 != 0) return p;
    if (((int64_t)g.n * g.hi + 4) * g.wi * g.ldi * 2 >= ((int64_t)1 << 31) || (int64_t)g.n * g.ho * g.wo * g.ldo * 2 >= ((int64_t)1 << 31)) return p;
    const int64_t M = (int64_t)g.n * g.hm * g.wm;
    if (M >= ((int64_t)1 << 30)) return p;
    const int W = (g.ci >= 128 && g.co >= 128) ? 4 : 2, T = 32 * W;
    p.gemm = W; p.WCI = 0; p.NTW = 0;
    p.gy = (g.ci + T - 1) / T; p.gz = (g.co + T - 1) / T;
    const int64_t nchunks = (M + 63) / 64;
    // about 1024 blocks per launch (two rounds of two blocks per CU), at least 4 chunks per block, slabs <= 16 MB per layer
    static const int target_blocks = [] { const char* e = getenv("STCD_WGEMM_BLOCKS"); return e ? atoi(e) : 512; }();
    static const int min_chunks = [] { const char* e = getenv("STCD_WGEMM_MIN_CHUNKS"); return e ? atoi(e) : 16; }();
    int64_t gx = std::max<int64_t>(1, target_blocks / ((int64_t)p.gy * p.gz));
    gx = std::min<int64_t>(gx, std::max<int64_t>(1, nchunks / min_chunks));
    gx = std::min<int64_t>(gx, std::max<int64_t>(1, ((int64_t)16 << 20) / ((int64_t)kpad * wld * 4)));
    p.gx = (int)gx;
    p.slab_floats = gx * (int64_t)kpad * wld;
    p.ok = true;
    return p;
}

WgradJob wgrad_gemm_make_job(const stcd_conv_geom& g, const WgradMfmaPlan& p, int64_t in_off, int64_t dout_off, int64_t slab_off,
                             int kpad, int wld) {
    WgradJob a;
    memset(&a, 0, sizeof(a));
    a.g = g;
    a.in_off = in_off; a.dout_off = dout_off; a.slab_off = slab_off; a.kpad = kpad; a.wld = wld;
    a.ntiles = g.n * g.hm * g.wm;                 // positions
    a.co_valid = (g.co + 7) & ~7;
    a.in_bytes = (unsigned)((int64_t)g.n * g.hi * g.wi * g.ldi * 2);
    a.dout_bytes = (unsigned)((int64_t)g.n * g.ho * g.wo * g.ldo * 2);
    a.gx = p.gx; a.gy = p.gy; a.gz = p.gz; a.start = 0;
    a.wi_valid = p.wi_valid > 0 ? p.wi_valid : g.wi;
    const int T = 32 * p.gemm;
    a.lds_bytes = 2 * 2 * (T / 32) * (64 * 64 + 8 * 32);
    return a;
}

int launch_wgrad_gemm_group(int W, const WgradJob* jobs_dev, int njobs, int total_blocks, const char* base, hipStream_t s) {
    if (njobs <= 0 || total_blocks <= 0) return 0;
    const int T = 32 * W;
    const size_t lds = (size_t)2 * 2 * (T / 32) * (64 * 64 + 8 * 32);
    if (W == 4) {
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_wgrad_gemm<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
        k_wgrad_gemm<4><<<(unsigned)total_blocks, 256, lds, s>>>(jobs_dev, njobs, base);
    } else {
        k_wgrad_gemm<2><<<(unsigned)total_blocks, 256, lds, s>>>(jobs_dev, njobs, base);
    }
    return 0;
}

}  // namespace stcd
