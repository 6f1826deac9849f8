// kernels_conv_dma.hip -- tap-list convolutions of WIDE layers (Ci % 64 == 0, Co % 256 == 0) on LARGE maps as an implicit GEMM
// on a 256 (positions) x 256 (output channels) block tile, staged by LDS-DMA (`buffer_load_dwordx4 ... lds`): the 256 -> 256
// 3x3 convolutions and the phases of the 4x4 stride-2 transposed convolutions of ChangeFormer's decoder head
// (/root/reference/models/ChangeFormerBaseNetworks.py:85-120 inside /root/reference/models/ChangeFormer.py:1540-1631), forward
// and data gradient -- 85 % of that model's arithmetic.  Same operands, fragment image (conv_mfma_plan, CiB = 64), epilogue
// (ConvEpi) and results as k_conv_gemm; what differs is the pipeline:
//
//   * 8 waves = 2 (positions: wr) x 4 (channels: wc), each 128 positions x 64 channels = 8 x 4 accumulator fragments of the
//     16x16x32 bf16 MFMA (weights are the A operand: a lane ends up with 4 consecutive channels of one position).
//   * K walks in K-TILES (64-channel chunk, tap), tap fastest -- the order of the fragment image.  A K-tile is four HALF-TILES of
//     16 KB: A0 / A1 = the activation rows of the first / second 64 positions of BOTH wave rows (128 rows x 128 B, 16-B chunk
//     index XOR-swizzled by (row >> 1) & 7 -- applied to the SOURCE address, the LDS image of a DMA is lane-linear), B0 / B1 =
//     the filter fragments of the first / second 32 channels of ALL FOUR wave columns (16 fragments of 1 KB, lane-linear, read
//     back conflict-free).  Two LDS slots per half-tile kind: 128 KB.
//   * One K-tile = four PHASES of 16 MFMAs per wave (one quadrant of the wave's tile x K = 64):
//         phase 0: read b0 (4 x ds_read_b128), a0 (8)   MFMA (a0, b0)      stages A1 of K-tile t + 1
//         phase 1: read b1 (4)                          MFMA (a0, b1)      stages B0 of K-tile t + 2
//         phase 2: read a1 (8)                          MFMA (a1, b1)      stages A0 of K-tile t + 2
//         phase 3:                                      MFMA (a1, b0)      stages B1 of K-tile t + 2, then s_waitcnt vmcnt(6)
//     A phase is  [fragment reads, 2 DMA instructions per thread]  s_barrier  [16 MFMAs]  s_barrier.  The two wave rows run ONE
//     BARRIER APART (wave row 1 enters the loop through an extra barrier): while one row multiplies, the other -- its SIMD
//     neighbours -- reads fragments and issues DMAs.
//   * Ordering rules (programming guide section 5, "The 256^2 8-phase template"):
//       RAW: a half-tile is read at the earliest one phase after the counted vmcnt that retired its DMAs (phase 3 retires
//            K-tile t + 1, first read in phase 0 of K-tile t + 1; the three half-tiles of K-tile t + 2 stay in flight);
//       WAR: a slot is re-staged two phases after its last read, or one phase after when an lgkmcnt in front of the reading
//            phase's first barrier retired the reads (b0 in phase 0: `s_waitcnt lgkmcnt(8)` behind the 4 + 8 reads, issue order
//            pinned by sched_barrier).
//   * Zero padding: a tap outside the image gets a voffset past num_records; the DMA then writes zeros.
//   * Epilogue: bias, ReLU, rounding; the 256 x 256 tile goes through LDS once (all staging slots are dead) and leaves as full
//     16-B pieces of complete output rows, with gate / residual (ConvEpi) applied in that pass exactly as k_conv_gemm does.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace stcd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct ConvDmaArgs {
    stcd_conv_geom g;
    const bf16* in; const bf16* wf; const float* bias; bf16* out;
    int NTtot, nk;                 // n-tiles of the fragment image; K-tiles = (Ci / 64) * ntaps (even, >= 4)
    unsigned lead, in_bytes;       // the X descriptor starts `lead` bytes before the tensor (most negative tap offset)
    int M;                         // positions
    int tiles_m, tiles_n;
    unsigned long long tapbits;    // 6 bits per tap: (dy + 3) | (dx + 3) << 3 -- decoded with scalar shifts, no memory access in the loop
    int relu;
    const bf16* gate; int ldg;
    const bf16* res; int ldr;
    float alpha, beta;
};

constexpr int DM_HT = 16384;                                   // bytes of a half-tile
constexpr int DM_OPITCH = 256 * 2 + 16;                        // out-tile row pitch (bytes)
#define DM_A(H_, SLOT_) ((((SLOT_) * 2 + (H_)) * DM_HT))
#define DM_B(H_, SLOT_) ((4 * DM_HT + ((SLOT_) * 2 + (H_)) * DM_HT))
#define DM_LDS(P_) ((__attribute__((address_space(3))) void*)(P_))

__global__ void __launch_bounds__(512, 1)
k_conv_dma(const ConvDmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r = lane & 15;
    const int wr = wid >> 2, wc = wid & 3;
    // block -> tile: the blocks of one XCD (ids congruent mod 8) take a CONTIGUOUS range of tiles, so the rows neighbouring
    // tiles share (the +-1 image-row taps) and the filter meet in that XCD's L2 (bijective form of the remap)
    int tile;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
        tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    }
    const int nt = tile % a.tiles_n, mt = tile / a.tiles_n;
    const int m0 = mt * 256, nf0 = nt * 16;

    // ---- staging plan.  A: DMA instruction (h, i) of wave w moves rows rho = (2w + i) * 8 + (lane >> 3) of half-tile h, physical
    //      16-B chunk lane & 7; tile row of rho = (rho >> 6) * 128 + h * 64 + (rho & 63).
    unsigned xoff[2][2], xmask[2][2];
    {
        const int ntaps = a.g.ntaps;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rho = (wid * 2 + i) * 8 + (lane >> 3);
                const int trow = (rho >> 6) * 128 + h * 64 + (rho & 63);
                const int lc = (lane & 7) ^ ((rho >> 1) & 7);
                const int m = m0 + trow;
                unsigned off = 0x80000000u, mask = 0;
                if (m < a.M) {
                    const int x = m % a.g.wm, t = m / a.g.wm, y = t % a.g.hm, n = t / a.g.hm;
                    const int yi = y * a.g.in_stride, xi = x * a.g.in_stride;
                    off = (unsigned)(((((int64_t)n * a.g.hi + yi) * a.g.wi + xi) * a.g.ldi + lc * 8) * 2) + a.lead;
                    for (int tp = 0; tp < ntaps; ++tp) {
                        const int dy = (int)((a.tapbits >> (6 * tp)) & 7) - 3, dx = (int)((a.tapbits >> (6 * tp + 3)) & 7) - 3;
                        if ((unsigned)(yi + dy) < (unsigned)a.g.hi && (unsigned)(xi + dx) < (unsigned)a.g.wi) mask |= 1u << tp;
                    }
                }
                xoff[h][i] = off; xmask[h][i] = mask;
            }
    }
    // B: DMA instruction (h, i) of wave w moves fragment piece pi = 2w + i of half-tile h: k-step pi >> 3, wave column (pi & 7) >> 1,
    //    fragment h * 2 + (pi & 1) of that column
    unsigned woff[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pi = wid * 2 + i, ks = pi >> 3, f8 = pi & 7;
            const int ntile = (f8 >> 1) * 4 + h * 2 + (f8 & 1);
            woff[h][i] = (unsigned)(((ks * a.NTtot + nf0 + ntile) * 64 + lane) * 16);
        }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.in)) - a.lead, (short)0, (int)(a.in_bytes + a.lead), 0x00020000);
    const unsigned wstep = (unsigned)(2 * a.NTtot * 1024);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.wf)), (short)0, (int)((unsigned)a.nk * wstep), 0x00020000);
    const int ntaps = a.g.ntaps;
    const int wi_ldi2 = a.g.wi * a.g.ldi * 2, ldi2 = a.g.ldi * 2;

#define DM_STAGE_A(H_, SLOT_, CC_, TT_)                                                                                \
    do {                                                                                                               \
        const int tt_ = __builtin_amdgcn_readfirstlane(TT_);                                                                                       \
        const int tdy_ = (int)((a.tapbits >> (6 * tt_)) & 7) - 3, tdx_ = (int)((a.tapbits >> (6 * tt_ + 3)) & 7) - 3;  \
        const unsigned toff_ = (unsigned)(tdy_ * wi_ldi2 + tdx_ * ldi2);                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
            const unsigned vo_ = ((xmask[H_][i] >> tt_) & 1u) ? xoff[H_][i] + toff_ : 0x80000000u;                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, DM_LDS(smem + DM_A(H_, SLOT_) + (wid * 2 + i) * 1024), 16,   \
                                                     (int)vo_, __builtin_amdgcn_readfirstlane((CC_) * 128), 0, 0);                                     \
        }                                                                                                              \
    } while (0)
#define DM_STAGE_B(H_, SLOT_, KT_)                                                                                     \
    do {                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, DM_LDS(smem + DM_B(H_, SLOT_) + (wid * 2 + i) * 1024), 16,   \
                                                     (int)woff[H_][i], __builtin_amdgcn_readfirstlane((int)((unsigned)(KT_) * wstep)), 0, 0);          \
    } while (0)

    // ---- fragment read plan
    const int sw = (r >> 1) & 7;
    const int va0 = (wr * 64 + r) * 128 + ((q ^ sw) * 16), va1 = (wr * 64 + r) * 128 + (((4 + q) ^ sw) * 16);
    const int vb = (wc * 2 * 64 + lane) * 16;
    bf16x8 af[2][4], b0f[2][2], b1f[2][2];
    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#define DM_READ_A(H_, SLOT_)                                                                                           \
    do {                                                                                                               \
        _Pragma("unroll") for (int mm = 0; mm < 4; ++mm) {                                                             \
            af[0][mm] = *reinterpret_cast<const bf16x8*>(smem + DM_A(H_, SLOT_) + va0 + mm * 2048);                    \
            af[1][mm] = *reinterpret_cast<const bf16x8*>(smem + DM_A(H_, SLOT_) + va1 + mm * 2048);                    \
        }                                                                                                              \
    } while (0)
#define DM_READ_B(BF_, H_, SLOT_)                                                                                      \
    do {                                                                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
            _Pragma("unroll") for (int nn = 0; nn < 2; ++nn)                                                           \
                BF_[ks][nn] = *reinterpret_cast<const bf16x8*>(smem + DM_B(H_, SLOT_) + vb + (ks * 8 + nn) * 1024);    \
    } while (0)
#define DM_MMA(HA_, HB_, BF_)                                                                                          \
    do {                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
            _Pragma("unroll") for (int mm = 0; mm < 4; ++mm)                                                           \
                _Pragma("unroll") for (int nn = 0; nn < 2; ++nn)                                                       \
                    acc[(HA_) * 4 + mm][(HB_) * 2 + nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                     \
                        BF_[ks][nn], af[ks][mm], acc[(HA_) * 4 + mm][(HB_) * 2 + nn], 0, 0, 0);                        \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)
#define DM_BAR()                                                                                                       \
    do {                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        __builtin_amdgcn_s_barrier();                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)
    // one K-tile in slot SLOT_.  MODE_ 0: steady state; 1: K-tile nk - 2 (only A1 of the last K-tile is left to stage; its wait
    // is vmcnt(0)); 2: the last K-tile (nothing to stage, nothing to wait for).  (c1, t1) = (chunk, tap) of K-tile t + 1.
#define DM_KTILE(SLOT_, MODE_)                                                                                         \
    do {                                                                                                               \
        int c2_ = c1, t2_ = t1 + 1;                                                                                    \
        if (t2_ == ntaps) { t2_ = 0; ++c2_; }                                                                          \
        /* phase 0 */                                                                                                  \
        DM_READ_B(b0f, 0, SLOT_);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        DM_READ_A(0, SLOT_);                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if ((MODE_) <= 1) DM_STAGE_A(1, (SLOT_) ^ 1, c1, t1);                                                          \
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                                             \
        DM_BAR();                                                                                                      \
        DM_MMA(0, 0, b0f);                                                                                             \
        DM_BAR();                                                                                                      \
        /* phase 1 */                                                                                                  \
        DM_READ_B(b1f, 1, SLOT_);                                                                                      \
        if ((MODE_) == 0) DM_STAGE_B(0, SLOT_, kt + 2);                                                                \
        DM_BAR();                                                                                                      \
        DM_MMA(0, 1, b1f);                                                                                             \
        DM_BAR();                                                                                                      \
        /* phase 2 */                                                                                                  \
        DM_READ_A(1, SLOT_);                                                                                           \
        if ((MODE_) == 0) DM_STAGE_A(0, SLOT_, c2_, t2_);                                                              \
        DM_BAR();                                                                                                      \
        DM_MMA(1, 1, b1f);                                                                                             \
        DM_BAR();                                                                                                      \
        /* phase 3 */                                                                                                  \
        if ((MODE_) == 0) { DM_STAGE_B(1, SLOT_, kt + 2); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }           \
        if ((MODE_) == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
        DM_BAR();                                                                                                      \
        DM_MMA(1, 0, b0f);                                                                                             \
        DM_BAR();                                                                                                      \
        c1 = c2_; t1 = t2_; ++kt;                                                                                      \
    } while (0)

    // ---- prologue: K-tile 0 and B0, A0, B1 of K-tile 1 (the loop's first phase stages A1 of K-tile 1)
    int kt = 0;                                  // K-tile being multiplied
    int c1 = 0, t1 = 1;                          // (chunk, tap) of K-tile kt + 1
    if (t1 == ntaps) { t1 = 0; c1 = 1; }
    DM_STAGE_B(0, 0, 0);
    DM_STAGE_A(0, 0, 0, 0);
    DM_STAGE_B(1, 0, 0);
    DM_STAGE_A(1, 0, 0, 0);
    DM_STAGE_B(0, 1, 1);
    DM_STAGE_A(0, 1, c1, t1);
    DM_STAGE_B(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    DM_BAR();
    if (wr == 1) DM_BAR();                       // wave row 1 runs one barrier behind wave row 0
    const int nk = a.nk;
    for (int it = 0; it < nk - 2; it += 2) {
        DM_KTILE(0, 0);
        DM_KTILE(1, 0);
    }
    DM_KTILE(0, 1);
    DM_KTILE(1, 2);
    if (wr == 0) DM_BAR();
#undef DM_KTILE
#undef DM_MMA
#undef DM_READ_A
#undef DM_READ_B
#undef DM_STAGE_A
#undef DM_STAGE_B

    // ---- epilogue: bias, ReLU, rounding; the tile goes to LDS [256 rows][256 channels] (pitch DM_OPITCH)
    char* const ot = smem;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int cl = wc * 64 + n * 16 + 4 * q;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) { const float4 b4 = *reinterpret_cast<const float4*>(a.bias + nt * 256 + cl); bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w; }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int row = wr * 128 + m * 16 + r;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[m][n][j] + bv[j];
                if (a.relu) v[j] = fmaxf(v[j], 0.f);
            }
            uint2 pk;
            pk.x = pack_bf16x2(v[0], v[1]);
            pk.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(ot + row * DM_OPITCH + cl * 2) = pk;
        }
    }
    __syncthreads();
    const bool plain_out = a.g.out_stride == 1 && a.g.ho == a.g.hm && a.g.wo == a.g.wm && a.g.oy0 == 0 && a.g.ox0 == 0;
#pragma unroll 4
    for (int p = 0; p < 16; ++p) {
        const int i = tid + p * 512, row = i >> 5, c8 = i & 31, m = m0 + row;
        if (m < a.M) {
            int64_t opix = m;
            if (!plain_out) {
                const int x = m % a.g.wm, t = m / a.g.wm, y = t % a.g.hm, n = t / a.g.hm;
                opix = ((int64_t)n * a.g.ho + y * a.g.out_stride + a.g.oy0) * a.g.wo + x * a.g.out_stride + a.g.ox0;
            }
            uint4 pk = *reinterpret_cast<const uint4*>(ot + row * DM_OPITCH + c8 * 16);
            if (a.gate || a.res) {      // the rounded conv output, gated and / or combined with a residual, rounded once more
                float v8[8];
                const uint32_t wv[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) { v8[2 * i2] = __uint_as_float(wv[i2] << 16); v8[2 * i2 + 1] = __uint_as_float(wv[i2] & 0xffff0000u); }
                if (a.gate) {
                    float g8[8];
                    load8(a.gate + opix * a.ldg + nt * 256 + c8 * 8, g8);
#pragma unroll
                    for (int i2 = 0; i2 < 8; ++i2) v8[i2] = g8[i2] > 0.f ? v8[i2] : 0.f;
                }
                if (a.res) {
                    float r8[8];
                    load8(a.res + opix * a.ldr + nt * 256 + c8 * 8, r8);
#pragma unroll
                    for (int i2 = 0; i2 < 8; ++i2) v8[i2] = a.alpha * v8[i2] + a.beta * r8[i2];
                }
                pk.x = pack_bf16x2(v8[0], v8[1]); pk.y = pack_bf16x2(v8[2], v8[3]);
                pk.z = pack_bf16x2(v8[4], v8[5]); pk.w = pack_bf16x2(v8[6], v8[7]);
            }
            *reinterpret_cast<uint4*>(a.out + opix * a.g.ldo + nt * 256 + c8 * 8) = pk;
        }
    }
}

static bool conv_dma_enabled() {
    static const bool on = [] { const char* e = getenv("STCD_NO_DMA_KERNEL"); return !(e && e[0] == '1'); }();
    return on;
}

ConvDmaPlan conv_dma_plan(const stcd_conv_geom& g, const ConvMfmaPlan& p) {
    ConvDmaPlan dp;
    if (!conv_dma_enabled() || !p.ok || p.modeB || p.CiB != 64 || g.ntaps < 1) return dp;
    for (int t = 0; t < g.ntaps; ++t)
        if (g.dy[t] < -3 || g.dy[t] > 3 || g.dx[t] < -3 || g.dx[t] > 3) return dp;
    if (g.ci % 64 != 0 || g.co % 256 != 0 || g.ldi % 8 != 0 || g.ldo % 8 != 0 || p.NTtot * 16 != g.co) return dp;
    const int nk = (g.ci / 64) * g.ntaps;
    if (nk < 4 || (nk & 1)) return dp;
    if (((int64_t)g.n * g.hi + 8) * g.wi * g.ldi * 2 >= ((int64_t)1 << 31) || g.hi >= 16384 || g.wi >= 16384) return dp;
    if ((int64_t)nk * 2 * p.NTtot * 1024 >= ((int64_t)1 << 31)) return dp;
    if ((g.hm - 1) * g.in_stride >= g.hi + 2 || (g.wm - 1) * g.in_stride >= g.wi + 2) return dp;
    const int64_t M = (int64_t)g.n * g.hm * g.wm;
    if (M >= ((int64_t)1 << 31)) return dp;
    dp.tiles_m = (int)((M + 255) / 256);
    dp.tiles_n = g.co / 256;
    dp.blocks = dp.tiles_m * dp.tiles_n;
    dp.lds_bytes = std::max(8 * DM_HT, 256 * DM_OPITCH);
    dp.ok = true;
    return dp;
}

int launch_conv_dma(const stcd_conv_geom& g, const ConvMfmaPlan& p, const ConvDmaPlan& dp, const void* in, const void* wf,
                    const float* bias, void* out, hipStream_t s, const ConvEpi* epi) {
    if (!dp.ok) return 1;
    ConvDmaArgs a;
    a.g = g;
    a.in = (const bf16*)in; a.wf = (const bf16*)wf; a.bias = bias; a.out = (bf16*)out;
    a.NTtot = p.NTtot; a.nk = (g.ci / 64) * g.ntaps;
    a.lead = (unsigned)((3 * g.wi + 3) * g.ldi * 2);
    a.in_bytes = (unsigned)((int64_t)g.n * g.hi * g.wi * g.ldi * 2);
    a.M = g.n * g.hm * g.wm;
    a.tiles_m = dp.tiles_m; a.tiles_n = dp.tiles_n;
    a.tapbits = 0;
    for (int t = 0; t < g.ntaps; ++t)
        a.tapbits |= (unsigned long long)(((g.dy[t] + 3) & 7) | (((g.dx[t] + 3) & 7) << 3)) << (6 * t);
    a.relu = epi ? epi->relu : 0;
    a.gate = epi ? (const bf16*)epi->gate : nullptr; a.ldg = epi ? epi->ldg : 0;
    a.res = epi ? (const bf16*)epi->res : nullptr; a.ldr = epi ? epi->ldr : 0;
    a.alpha = epi ? epi->alpha : 1.f; a.beta = epi ? epi->beta : 1.f;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_conv_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    k_conv_dma<<<(unsigned)dp.blocks, 512, (size_t)dp.lds_bytes, s>>>(a);
    return 0;
}

}  // namespace stcd
