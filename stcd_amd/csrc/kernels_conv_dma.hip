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
//   * PERSISTENT blocks, one per CU: block b plays the virtual blocks b, b + 256, ... of a one-tile-per-block grid dealt round-robin
//     over the XCDs, so every XCD walks a contiguous eighth of the tiles, 32 neighbours at a time (tile order matters: "all XCDs
//     sweep one front" measured 6 % slower at 4 x 512^2).  After a tile's last K-tile the next tile's first 14 DMAs go out BEFORE
//     the epilogue.
//   * Epilogue: bias (kept in LDS for the kernel's life), ReLU, rounding; every wave passes its 128 x 64 sub-tile through its OWN
//     2-KB scratch in strips of 16 rows (inline-asm ds_write_b64 / ds_read_b128: an LDS access hipcc can see would be preceded by
//     `s_waitcnt vmcnt(0)`, i.e. a wait for the DMAs in flight) and stores complete 128-B row segments with buffer_store_dwordx4,
//     gate / residual (ConvEpi) applied in that pass exactly as k_conv_gemm does.  Exactly 16 store instructions per lane and
//     tile (a row past the end is dropped by the range check), so the next tile's first wait is `vmcnt(22)`: all but the 6
//     youngest DMAs and the 16 stores behind them.
//   * MEASURED (MI355X, random data, 4 x 512 x 512 x 256 -> 256, filter repack included): 1005 us = 1231 TFLOP/s (k_conv_gemm: 1686 us);
//     without the s_setprio pair around the MFMA clusters 1056-1069 us (-5 %);
//     time against the K-tile count puts the loop at ~1500-1600 TFLOP/s and the per-tile fixed cost at ~11 us, of which the
//     output stores are 2.4-3.3 us (STCD_DMA_DBG=1).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace stcd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct ConvDmaArgs {
    stcd_conv_geom g;
    const bf16* in; const bf16* wf; const float* bias; bf16* out;
    int NTtot, nk;                 // n-tiles of the fragment image; K-tiles = (Ci / 64) * ntaps (even, >= 4)
    unsigned lead, in_bytes;       // the X descriptor starts `lead` bytes before the tensor (most negative tap offset)
    unsigned out_bytes;
    int dbg;                       // development switches (STCD_DMA_DBG, compiled only with -DSTCD_DEV_SWITCHES): 1 = drop every output store (sizes the store drain: 2.4 - 3.3 us of the ~11 us per tile)
    int M;                         // positions
    int tiles_m, tiles_n;
    unsigned long long tapbits;    // 6 bits per tap: (dy + 3) | (dx + 3) << 3 -- decoded with scalar shifts, no memory access in the loop
    int relu;
    const bf16* gate; int ldg;
    const bf16* res; int ldr;
    float alpha, beta;
};

constexpr int DM_HT = 16384;                                   // bytes of a half-tile
constexpr int DM_SCR = 16 * 144;                               // bytes of one wave's epilogue scratch (16 rows x 128 B, pitch 144)
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
    // PERSISTENT blocks (one per CU).  In round j block b plays the virtual block B = j * nblk + b of a one-tile-per-block grid whose
    // blocks are dealt round-robin over the 8 XCDs (XCD = B mod 8 = b mod 8 as nblk is a multiple of 8, or the only round): XCD x
    // owns a CONTIGUOUS range of tiles (bijective split of ntiles over the 8 XCDs) and walks it front to back, 32 neighbouring
    // tiles at a time -- the rows neighbouring tiles share (the +-1 image-row taps), this round's and the last one's, and the
    // filter meet in that XCD's L2.
    const int nblk = gridDim.x, ntiles = a.tiles_m * a.tiles_n;
    int vb = blockIdx.x;                                   // virtual block id
    const int xq = ntiles >> 3, xr = ntiles & 7;
#define DM_TILE_OF(B_) ((((B_) & 7) < xr ? ((B_) & 7) * (xq + 1) : xr * (xq + 1) + (((B_) & 7) - xr) * xq) + ((B_) >> 3))
    int tile = DM_TILE_OF(vb);

    // ---- staging plan.  A: DMA instruction (h, i) of wave w moves rows rho = (2w + i) * 8 + (lane >> 3) of half-tile h, physical
    //      16-B chunk lane & 7; tile row of rho = (rho >> 6) * 128 + h * 64 + (rho & 63).
    const int ntaps = a.g.ntaps;
#define DM_DECODE(TILE_)                                                                                               \
    do {                                                                                                               \
        const int m0_ = ((TILE_) / a.tiles_n) * 256;                                                                   \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                  \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                            \
                const int rho = (wid * 2 + i) * 8 + (lane >> 3);                                                       \
                const int trow = (rho >> 6) * 128 + h * 64 + (rho & 63);                                               \
                const int lc = (lane & 7) ^ ((rho >> 1) & 7);                                                          \
                const int m = m0_ + trow;                                                                              \
                unsigned off = 0x80000000u;                                                                            \
                int yx = (int)0xC000C000;                  /* far outside every image: all taps of a masked row read zeros */ \
                if (m < a.M) {                                                                                         \
                    const int x = m % a.g.wm, t = m / a.g.wm, y = t % a.g.hm, n = t / a.g.hm;                          \
                    const int yi = y * a.g.in_stride, xi = x * a.g.in_stride;                                          \
                    off = (unsigned)(((((int64_t)n * a.g.hi + yi) * a.g.wi + xi) * a.g.ldi + lc * 8) * 2) + a.lead;    \
                    yx = (yi << 16) | xi;                                                                              \
                }                                                                                                      \
                xoff[h][i] = off; xyx[h][i] = yx;                                                                      \
            }                                                                                                          \
    } while (0)
    unsigned xoff[2][2]; int xyx[2][2];                  // byte offset of the row's centre pixel (+ lead), packed (y, x) of it
    // B: DMA instruction (h, i) of wave w moves fragment piece pi = 2w + i of half-tile h: k-step pi >> 3, wave column (pi & 7) >> 1,
    //    fragment h * 2 + (pi & 1) of that column (the tile's first n-fragment goes into the scalar offset)
    unsigned woff[2];                                    // (half-tile h: two n-fragments = 2 KB further, in the scalar offset)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pi = wid * 2 + i, ks = pi >> 3, f8 = pi & 7;
        const int ntile = (f8 >> 1) * 4 + (f8 & 1);
        woff[i] = (unsigned)(((ks * a.NTtot + ntile) * 64 + lane) * 16);
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.in)) - a.lead, (short)0, (int)(a.in_bytes + a.lead), 0x00020000);
    const unsigned wstep = (unsigned)(2 * a.NTtot * 1024);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.wf)), (short)0, (int)((unsigned)a.nk * wstep), 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(a.out), (short)0, (int)a.out_bytes, 0x00020000);
    const int wi_ldi2 = a.g.wi * a.g.ldi * 2, ldi2 = a.g.ldi * 2;
    int nt = tile % a.tiles_n, m0 = (tile / a.tiles_n) * 256;

#define DM_STAGE_A(H_, SLOT_, CC_, TT_)                                                                                \
    do {                                                                                                               \
        const int tt_ = __builtin_amdgcn_readfirstlane(TT_);                                                           \
        const int tdy_ = (int)((a.tapbits >> (6 * tt_)) & 7) - 3, tdx_ = (int)((a.tapbits >> (6 * tt_ + 3)) & 7) - 3;  \
        const unsigned toff_ = (unsigned)(tdy_ * wi_ldi2 + tdx_ * ldi2);                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
            const bool ok_ = (unsigned)((xyx[H_][i] >> 16) + tdy_) < (unsigned)a.g.hi &&                               \
                             (unsigned)((int)(short)(xyx[H_][i] & 0xffff) + tdx_) < (unsigned)a.g.wi;                  \
            const unsigned vo_ = ok_ ? xoff[H_][i] + toff_ : 0x80000000u;                                              \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, DM_LDS(smem + DM_A(H_, SLOT_) + (wid * 2 + i) * 1024), 16,   \
                                                     (int)vo_, __builtin_amdgcn_readfirstlane((CC_) * 128), 0, 0);     \
        }                                                                                                              \
    } while (0)
#define DM_STAGE_B(H_, SLOT_, KT_)                                                                                     \
    do {                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, DM_LDS(smem + DM_B(H_, SLOT_) + (wid * 2 + i) * 1024), 16,   \
                                                     (int)woff[i],                                                     \
                                                     __builtin_amdgcn_readfirstlane((int)((unsigned)(KT_) * wstep + (unsigned)nt * 16384u + (H_) * 2048u)), 0, 0); \
    } while (0)

    // ---- fragment read plan
    const int sw = (r >> 1) & 7;
    const int va0 = (wr * 64 + r) * 128 + ((q ^ sw) * 16), va1 = (wr * 64 + r) * 128 + (((4 + q) ^ sw) * 16);
    const int vbo = (wc * 2 * 64 + lane) * 16;
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
                BF_[ks][nn] = *reinterpret_cast<const bf16x8*>(smem + DM_B(H_, SLOT_) + vbo + (ks * 8 + nn) * 1024);    \
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
    // the first 14 DMA instructions of a tile: K-tile 0 and B0, A0, B1 of K-tile 1 (the loop's first phase stages A1 of K-tile 1)
    int kt, c1, t1;
#define DM_PROLOGUE()                                                                                                  \
    do {                                                                                                               \
        kt = 0; c1 = 0; t1 = 1;                                                                                        \
        if (t1 == ntaps) { t1 = 0; c1 = 1; }                                                                           \
        DM_STAGE_B(0, 0, 0);                                                                                           \
        DM_STAGE_A(0, 0, 0, 0);                                                                                        \
        DM_STAGE_B(1, 0, 0);                                                                                           \
        DM_STAGE_A(1, 0, 0, 0);                                                                                        \
        DM_STAGE_B(0, 1, 1);                                                                                           \
        DM_STAGE_A(0, 1, c1, t1);                                                                                      \
        DM_STAGE_B(1, 1, 1);                                                                                           \
    } while (0)

    const bool plain_out = a.g.out_stride == 1 && a.g.ho == a.g.hm && a.g.wo == a.g.wm && a.g.oy0 == 0 && a.g.ox0 == 0;
    // this wave's transposition scratch (16 rows x 128 B, pitch 144): LDS byte addresses of the lane's write and read slots
    const unsigned scr0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + 8 * DM_HT + wid * DM_SCR;
    const unsigned scr_w = scr0 + r * 144 + q * 8, scr_r = scr0 + (lane >> 3) * 144 + (lane & 7) * 16;
    const int nk = a.nk;
    // the bias goes to LDS once (an ordinary global load in the epilogue would make hipcc wait `vmcnt(0)`, i.e. for the next tile's DMAs)
    {
        float* const lb = reinterpret_cast<float*>(smem + 8 * DM_HT + 8 * DM_SCR);
        for (int i = tid; i < a.g.co; i += 512) lb[i] = a.bias ? a.bias[i] : 0.f;
        __syncthreads();
    }
    const unsigned bias_r = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + 8 * DM_HT + 8 * DM_SCR + (wc * 64 + 4 * q) * 4;
    DM_DECODE(tile);
    DM_PROLOGUE();
    bool first = true;
    for (;;) {
        const int nvb = vb + nblk;
        const bool has_next = nvb < ntiles;
        const int ntile = has_next ? DM_TILE_OF(nvb) : tile;
        // K-tile 0 has landed: all but the 6 youngest DMAs -- and, from the second tile on, the 16 stores of the previous tile's
        // epilogue, which were issued behind this tile's first DMAs (vmcnt counts loads, stores and DMAs together, in order)
        if (first) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
        DM_BAR();
        if (wr == 1) DM_BAR();                       // wave row 1 runs one barrier behind wave row 0
        for (int it = 0; it < nk - 2; it += 2) {
            DM_KTILE(0, 0);
            DM_KTILE(1, 0);
        }
        DM_KTILE(0, 1);
        DM_KTILE(1, 2);
        if (wr == 0) DM_BAR();
        // every staging slot is dead: the next tile's first DMAs go out before this tile's epilogue
        const int em0 = m0, ent = nt;
        if (has_next) {
            DM_DECODE(ntile);
            nt = ntile % a.tiles_n; m0 = (ntile / a.tiles_n) * 256;
            DM_PROLOGUE();
        }
        // ---- epilogue: bias, ReLU, rounding; every wave passes its 128 x 64 sub-tile through its OWN scratch in strips of 16
        //      rows and stores complete 128-B row segments (8 rows per instruction); exactly 16 store instructions per lane and
        //      tile -- a row past the last position gets a voffset past num_records and is dropped by the hardware.
        {
            f32x4 bv[4];
            {
                const unsigned ba = bias_r + (unsigned)ent * 1024u;
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\tds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(bv[0]), "=&v"(bv[1]), "=&v"(bv[2]), "=&v"(bv[3]) : "v"(ba) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            const int prow = lane >> 3, piece = lane & 7;
            int mrow = em0 + wr * 128 + prow;                        // position of this lane's first output row; the others are +8 apart
            int ox = 0, oy = 0, on = 0;
            if (!plain_out) { ox = mrow % a.g.wm; const int t = mrow / a.g.wm; oy = t % a.g.hm; on = t / a.g.hm; }
            const int cbase = ent * 256 + wc * 64 + piece * 8;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = acc[m][n][j] + bv[n][j];
                        if (a.relu) v[j] = fmaxf(v[j], 0.f);
                        acc[m][n][j] = 0.f;
                    }
                    uint2 pk;
                    pk.x = pack_bf16x2(v[0], v[1]);
                    pk.y = pack_bf16x2(v[2], v[3]);
                    // (inline asm: an LDS access hipcc can see gets `s_waitcnt vmcnt(0)` in front of it while the next tile's DMAs fly)
                    asm volatile("ds_write_b64 %0, %1" :: "v"(scr_w + n * 32), "v"(pk) : "memory");
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    u32x4 pkv;
                    if (k == 0) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pkv) : "v"(scr_r) : "memory");
                    else asm volatile("ds_read_b128 %0, %1 offset:1152\n\ts_waitcnt lgkmcnt(0)" : "=v"(pkv) : "v"(scr_r) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    uint4 pk = make_uint4(pkv[0], pkv[1], pkv[2], pkv[3]);
#ifdef STCD_DEV_SWITCHES
                    const bool valid = mrow < a.M && !(a.dbg & 1);
#else
                    const bool valid = mrow < a.M;
#endif
                    int64_t opix = mrow;
                    if (!plain_out) opix = ((int64_t)on * a.g.ho + oy * a.g.out_stride + a.g.oy0) * a.g.wo + ox * a.g.out_stride + a.g.ox0;
                    if (!valid) opix = 0;
                    if (a.gate || a.res) {      // the rounded conv output, gated and / or combined with a residual, rounded once more
                        float v8[8];
                        const uint32_t wv[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                        for (int i2 = 0; i2 < 4; ++i2) { v8[2 * i2] = __uint_as_float(wv[i2] << 16); v8[2 * i2 + 1] = __uint_as_float(wv[i2] & 0xffff0000u); }
                        if (a.gate) {
                            float g8[8];
                            load8(a.gate + opix * a.ldg + cbase, g8);
#pragma unroll
                            for (int i2 = 0; i2 < 8; ++i2) v8[i2] = g8[i2] > 0.f ? v8[i2] : 0.f;
                        }
                        if (a.res) {
                            float r8[8];
                            load8(a.res + opix * a.ldr + cbase, r8);
#pragma unroll
                            for (int i2 = 0; i2 < 8; ++i2) v8[i2] = a.alpha * v8[i2] + a.beta * r8[i2];
                        }
                        pk.x = pack_bf16x2(v8[0], v8[1]); pk.y = pack_bf16x2(v8[2], v8[3]);
                        pk.z = pack_bf16x2(v8[4], v8[5]); pk.w = pack_bf16x2(v8[6], v8[7]);
                    }
                    const unsigned vo = valid ? (unsigned)((opix * a.g.ldo + cbase) * 2) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk.x, pk.y, pk.z, pk.w}, ors, vo, 0, 0);
                    // the lane's next row: 8 positions on
                    mrow += 8;
                    if (!plain_out) {
                        ox += 8;
                        if (ox >= a.g.wm) { ox -= a.g.wm; ++oy; if (oy >= a.g.hm) { oy = 0; ++on; } }      // (the plan requires wm >= 8)
                    }
                }
            }
        }
        if (!has_next) break;
        tile = ntile; vb = nvb; first = false;
    }
#undef DM_KTILE
#undef DM_MMA
#undef DM_READ_A
#undef DM_READ_B
#undef DM_STAGE_A
#undef DM_STAGE_B
#undef DM_PROLOGUE
#undef DM_DECODE
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
    if (g.wm < 8) return dp;
    if (((int64_t)g.n * g.ho * g.wo + 8) * g.ldo * 2 >= ((int64_t)1 << 31)) return dp;      // 32-bit buffer offsets of the output stores
    dp.blocks = std::min(dp.tiles_m * dp.tiles_n, 256);       // persistent: one block per CU (a multiple of 8, or a single round)
    if (g.co > 2048) return dp;                               // the bias copy in LDS
    dp.lds_bytes = 8 * DM_HT + 8 * DM_SCR + g.co * 4;
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
    a.out_bytes = (unsigned)((int64_t)g.n * g.ho * g.wo * g.ldo * 2);
    a.M = g.n * g.hm * g.wm;
#ifdef STCD_DEV_SWITCHES      // development builds only (make DEV=1): the shipped library never drops stores
    { static const int dbg = [] { const char* e = getenv("STCD_DMA_DBG"); return e ? atoi(e) : 0; }(); a.dbg = dbg; }
#else
    a.dbg = 0;
#endif
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
