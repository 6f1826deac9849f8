// kernels_wgrad_dma.hip -- weight gradient of WIDE 3x3 / tap-list stride-1 layers (Ci % 256 == 0, Co % 256 == 0) on LARGE maps:
//     dW[t][ci][co] = sum_pos X[pos + tap_t][ci] * dY[pos][co]
// (the 256 -> 256 convolutions of ChangeFormer's decoder head, /root/reference/models/ChangeFormerBaseNetworks.py:109-120) on
// the pipeline of kernels_conv_dma.hip: a GEMM per tap with M = co, N = ci, K = positions on a 256 (ci) x 256 (co) block tile,
// K-tiles of 64 positions staged by LDS-DMA with counted vmcnt, 8 waves = 2 (ci) x 4 (co), four phases of 16 MFMAs per K-tile,
// the two wave rows one barrier apart.  What is specific here:
//
//   * block = (split s, tap t, channel tile): the positions are cut into S contiguous ranges; every block owns ONE tap and ONE
//     256 x 256 channel tile for its whole life (128 accumulator registers per lane) and writes its part of the fp32 slab
//     [s][t][ci][co] once, with 16-B stores; the stage's batched reduce launch (k_reduce_jobs) sums the S slabs.  The 9 tap
//     blocks of a split get neighbouring ids on one XCD: they read the same dY rows and overlapping X rows from its L2.
//   * K is the SLOW axis of both operands (NHWC), so every fragment comes back through ds_read_b64_tr_b16 (a 16-lane group reads
//     4 positions x 16 channels and gets them channel-major; two reads make one 8-deep fragment).  Half-tile image: [64
//     positions][16 chunks of 16 B] (X0 / X1: the first / second 64 input channels of both wave rows; Y0 / Y1: the first /
//     second 32 output channels of all four wave columns), chunk index XOR-ed with 2 * ((p & 3) | ((p >> 3) & 1) << 2) -- the 8
//     positions x 2 chunks a 32-lane half reads then fall on 16 distinct 16-B bank slots.  The swizzle is applied to the SOURCE
//     address of the DMA (its LDS image is lane-linear: one instruction = 4 positions x 256 B).
//   * positions map linearly to pixels (stride 1, full map), so a DMA's address is base + (pos + tap shift) * pitch: only the
//     VALIDITY of a row needs (y, x), which each thread carries incrementally for its two rows (64 positions per K-tile).
//     Rows past the split's end, and taps outside the image, get a voffset past num_records: the DMA writes zeros.
//   * dY may be a strided view (the sub-pixel phases of a stride-2 transposed convolution read dY at (so*y + oy0, so*x + ox0)):
//     its row offset is then not linear in the position, so each thread carries the global row index R = n*h + y of its two rows
//     and forms  R * (so*wo*pitch) + x * (so*pitch)  per K-tile (two multiply-adds per row).
//   * the loop runs an even number of K-tiles in steady state to its end (the K-tiles it stages past the range are zeros that
//     nobody reads), then drains.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace stcd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int WD_HT = 16384;
#define WD_X(H_, SLOT_) ((((SLOT_) * 2 + (H_)) * WD_HT))
#define WD_Y(H_, SLOT_) ((4 * WD_HT + ((SLOT_) * 2 + (H_)) * WD_HT))
#define WD_LDS(P_) ((__attribute__((address_space(3))) void*)(P_))

// Two transposed reads = one 8-deep fragment.  Inline asm on purpose: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of
// every LDS read it can see while an LDS-DMA is in flight (it cannot prove the DMA's destination distinct), which drains the
// staging pipeline every phase; reads it cannot see are ordered by hand (counted vmcnt + barrier before the phase that reads,
// `s_waitcnt lgkmcnt(0)` + sched_barrier in front of the MFMAs that consume them).
template <int OFF>
__device__ __forceinline__ bf16x8 wd_tr(unsigned addr) {
    bf16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(lo), "=&v"(hi) : "v"(addr), "n"(OFF), "n"(OFF + 1024) : "memory");
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

__global__ void __launch_bounds__(512, 1)
k_wgrad_dma(const WgradJob* __restrict__ jobs, int njobs, const char* base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // logical block: the blocks of one XCD (ids congruent mod 8) take a contiguous range (bijective remap)
    int lb;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
        lb = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    }
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].start <= lb) lo = mid; else hi = mid - 1;
    }
    const WgradJob& a = jobs[lo];
    const int jb = lb - a.start;
    // (split, tap, channel tile), tap fastest: the taps of a split are neighbours
    const int tap = jb % a.gy, rest = jb / a.gy, tile = rest % a.gz, split = rest / a.gz;
    const int tco = a.g.co >> 8;
    const int ci0 = (tile / tco) << 8, co0 = (tile % tco) << 8;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, li = lane & 15, qrow = li >> 2, pcol = li & 3;
    const int wr = wid >> 2, wc = wid & 3;
    const int wi = a.g.wi, hi_ = a.g.hi, ldi = a.g.ldi, ldo = a.g.ldo;
    const int M = a.ntiles;                              // positions of the layer
    const int L = a.dma_L;                               // positions per split (a multiple of 64)
    const int P0 = split * L, Pend = min(M, P0 + L);
    const int nk = a.dma_nk;                             // K-tiles of a block (even)
    const int tdy = a.g.dy[tap], tdx = a.g.dx[tap];

    // ---- staging plan: DMA instruction i of wave w covers positions 4 * (2w + i) .. + 3 of the K-tile (lane >> 4), physical chunk
    //      lane & 15 -> logical chunk (lane & 15) ^ swz(pos)
    int py[2], px[2], pr[2];                             // (y, x) and global row index n*h + y of the thread's two rows in the K-tile being staged
    unsigned xvo[2], yvo[2];                             // their byte offsets inside a K-tile (X / dY), channel part included
    int prow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pos = (wid * 2 + i) * 4 + (lane >> 4);
        const int sw = 2 * ((pos & 3) | (((pos >> 3) & 1) << 2));
        const int c16 = (lane & 15) ^ sw;
        prow[i] = pos;
        xvo[i] = (unsigned)((pos * ldi + (c16 >> 3) * 128 + (c16 & 7) * 8) * 2);
        yvo[i] = (unsigned)((((int64_t)a.g.oy0 * a.g.wo + a.g.ox0) * ldo + (c16 >> 2) * 64 + (c16 & 3) * 8) * 2);
        const int p = P0 + pos;
        px[i] = p % wi; pr[i] = p / wi; py[i] = pr[i] % hi_;
    }
    const int dq = 64 / wi, dr = 64 - dq * wi;           // a K-tile advances (y, x) by (dq, dr) with at most one carry each
    const unsigned lead = (unsigned)((wi + 1) * ldi * 2);
    const bf16* a_in = reinterpret_cast<const bf16*>(base + a.in_off);
    const bf16* a_dout = reinterpret_cast<const bf16*>(base + a.dout_off);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a_in)) - lead, (short)0, (int)(a.in_bytes + lead), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a_dout)), (short)0, (int)a.dout_bytes, 0x00020000);
    // scalar byte offsets of the staged K-tile's first position (X: tap shift and `lead` included)
    unsigned sx = (unsigned)(((int64_t)P0 + tdy * wi + tdx) * ldi * 2 + ci0 * 2) + lead;
    const unsigned sy = (unsigned)(co0 * 2);
    const unsigned dsx = (unsigned)(64 * ldi * 2);
    const unsigned yrow_b = (unsigned)(a.g.out_stride * a.g.wo * ldo * 2), ycol_b = (unsigned)(a.g.out_stride * ldo * 2);
    int pk = P0;                                         // first position of the staged K-tile
    unsigned xv[2], yv[2];                               // voffsets of the staged K-tile (valid rows) or past num_records
#define WD_STATE()                                                                                                     \
    do {                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
            const bool inr_ = pk + prow[i] < Pend;                                                                     \
            const bool okx_ = inr_ && (unsigned)(py[i] + tdy) < (unsigned)hi_ && (unsigned)(px[i] + tdx) < (unsigned)wi; \
            xv[i] = okx_ ? xvo[i] : 0x80000000u;                                                                       \
            yv[i] = inr_ ? yvo[i] + (unsigned)pr[i] * yrow_b + (unsigned)px[i] * ycol_b : 0x80000000u;                 \
        }                                                                                                              \
    } while (0)
#define WD_ADVANCE()                                                                                                   \
    do {                                                                                                               \
        pk += 64; sx += dsx;                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
            px[i] += dr; py[i] += dq; pr[i] += dq;                                                                     \
            if (px[i] >= wi) { px[i] -= wi; ++py[i]; ++pr[i]; }                                                        \
            if (py[i] >= hi_) py[i] -= hi_;                                                                            \
        }                                                                                                              \
        WD_STATE();                                                                                                    \
    } while (0)
#define WD_STAGE_X(H_, SLOT_)                                                                                          \
    do {                                                                                                               \
        const int so_ = __builtin_amdgcn_readfirstlane((int)sx + (H_) * 128);   /* (an instruction offset would also move the LDS address) */ \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, WD_LDS(smem + WD_X(H_, SLOT_) + (wid * 2 + i) * 1024), 16,   \
                                                     (int)xv[i], so_, 0, 0);                                  \
    } while (0)
#define WD_STAGE_Y(H_, SLOT_)                                                                                          \
    do {                                                                                                               \
        const int so_ = __builtin_amdgcn_readfirstlane((int)sy + (H_) * 64);                                           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(yrs, WD_LDS(smem + WD_Y(H_, SLOT_) + (wid * 2 + i) * 1024), 16,   \
                                                     (int)yv[i], so_, 0, 0);                                   \
    } while (0)

    // ---- fragment read plan: position 8q + qrow (+ 4 for the second half, + 32 per k-step), 8-B column pcol of a 16-channel block
    const int fsw = 2 * (qrow | ((q & 1) << 2));
    const int fpos = (8 * q + qrow) * 256 + (pcol & 1) * 8;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned xa[4], ya[2];                               // LDS byte addresses (the Y region starts 64 KB up: beyond a 16-bit offset)
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) xa[mm] = lds0 + (unsigned)(fpos + (((wr * 8 + mm * 2 + (pcol >> 1)) ^ fsw) * 16));
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) ya[nn] = lds0 + (unsigned)(4 * WD_HT + fpos + (((wc * 4 + nn * 2 + (pcol >> 1)) ^ fsw) * 16));
    bf16x8 xf[2][4], y0f[2][2], y1f[2][2];
    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#define WD_READ_X(H_, SLOT_, M0_, M1_)                                                                                 \
    do {                                                                                                               \
        _Pragma("unroll") for (int mm = (M0_); mm < (M1_); ++mm) {                                                     \
            xf[0][mm] = wd_tr<WD_X(H_, SLOT_)>(xa[mm]);                                                                \
            xf[1][mm] = wd_tr<WD_X(H_, SLOT_) + 8192>(xa[mm]);                                                         \
        }                                                                                                              \
    } while (0)
#define WD_READ_Y(YF_, H_, SLOT_)                                                                                      \
    do {                                                                                                               \
        _Pragma("unroll") for (int nn = 0; nn < 2; ++nn) {                                                             \
            YF_[0][nn] = wd_tr<WD_X(H_, SLOT_)>(ya[nn]);                                                               \
            YF_[1][nn] = wd_tr<WD_X(H_, SLOT_) + 8192>(ya[nn]);                                                        \
        }                                                                                                              \
    } while (0)
#define WD_MMA(HA_, HB_, YF_)                                                                                          \
    do {                                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
            _Pragma("unroll") for (int mm = 0; mm < 4; ++mm)                                                           \
                _Pragma("unroll") for (int nn = 0; nn < 2; ++nn)                                                       \
                    acc[(HA_) * 4 + mm][(HB_) * 2 + nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                     \
                        YF_[ks][nn], xf[ks][mm], acc[(HA_) * 4 + mm][(HB_) * 2 + nn], 0, 0, 0);                        \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)
#define WD_BAR()                                                                                                       \
    do {                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        __builtin_amdgcn_s_barrier();                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)
    // one K-tile in slot SLOT_ (the staged state is K-tile t + 1 on entry, t + 2 from phase 1 on)
#define WD_KTILE(SLOT_)                                                                                                \
    do {                                                                                                               \
        /* phase 0: y0 (8 reads), x0 (16 reads; lgkmcnt is 4 bits: the wait that retires y0 sits after the first 8) */ \
        WD_READ_Y(y0f, 0, SLOT_);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        WD_READ_X(0, SLOT_, 0, 2);                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        WD_READ_X(0, SLOT_, 2, 4);                                                                                     \
        WD_STAGE_X(1, (SLOT_) ^ 1);                                                                                    \
        WD_BAR();                                                                                                      \
        WD_MMA(0, 0, y0f);                                                                                             \
        WD_BAR();                                                                                                      \
        /* phase 1 */                                                                                                  \
        WD_READ_Y(y1f, 1, SLOT_);                                                                                      \
        WD_ADVANCE();                                                                                                  \
        WD_STAGE_Y(0, SLOT_);                                                                                          \
        WD_BAR();                                                                                                      \
        WD_MMA(0, 1, y1f);                                                                                             \
        WD_BAR();                                                                                                      \
        /* phase 2 */                                                                                                  \
        WD_READ_X(1, SLOT_, 0, 4);                                                                                     \
        WD_STAGE_X(0, SLOT_);                                                                                          \
        WD_BAR();                                                                                                      \
        WD_MMA(1, 1, y1f);                                                                                             \
        WD_BAR();                                                                                                      \
        /* phase 3 */                                                                                                  \
        WD_STAGE_Y(1, SLOT_);                                                                                          \
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                               \
        WD_BAR();                                                                                                      \
        WD_MMA(1, 0, y0f);                                                                                             \
        WD_BAR();                                                                                                      \
    } while (0)

    // ---- prologue: K-tile 0 and Y0, X0, Y1 of K-tile 1
    WD_STATE();
    WD_STAGE_Y(0, 0);
    WD_STAGE_X(0, 0);
    WD_STAGE_Y(1, 0);
    WD_STAGE_X(1, 0);
    WD_ADVANCE();
    WD_STAGE_Y(0, 1);
    WD_STAGE_X(0, 1);
    WD_STAGE_Y(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    WD_BAR();
    if (wr == 1) WD_BAR();
    for (int it = 0; it < nk; it += 2) {
        WD_KTILE(0);
        WD_KTILE(1);
    }
    if (wr == 0) WD_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero K-tiles staged past the range
#undef WD_KTILE
#undef WD_MMA
#undef WD_READ_X
#undef WD_READ_Y
#undef WD_STAGE_X
#undef WD_STAGE_Y
#undef WD_ADVANCE
#undef WD_STATE

    // ---- the block's part of the slab: lane (q, li) holds dW[ci = wr*128 + m*16 + li][co = wc*64 + n*16 + 4q .. + 3]
    float* slab = reinterpret_cast<float*>(const_cast<char*>(base) + a.slab_off) +
                  ((int64_t)split * a.gy + tap) * a.kpad * a.wld;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int ci = ci0 + wr * 128 + m * 16 + li;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int co = co0 + wc * 64 + n * 16 + 4 * q;
            *reinterpret_cast<float4*>(slab + (int64_t)ci * a.wld + co) = make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
        }
    }
}

static bool wgrad_dma_enabled() {
    static const bool on = [] { const char* e = getenv("STCD_NO_WGRAD_DMA"); return !(e && e[0] == '1'); }();
    return on;
}

// ok: stride-1 full-map tap list (dY plain or the (oy0, ox0) phase of a map twice as large), Ci % 256 == 0, Co % 256 == 0, maps at least 64 wide and 2 high, taps within +-1 ... (the lead of the
// X descriptor is one row + one pixel).  gx (the split count) is set by the caller (wgrad_dma_set_split).
WgradMfmaPlan wgrad_dma_plan(const stcd_conv_geom& g, int kpad, int wld) {
    WgradMfmaPlan p;
    if (!wgrad_dma_enabled()) return p;
    if (g.in_stride != 1 || g.hm != g.hi || g.wm != g.wi) return p;
    if (g.out_stride < 1 || g.out_stride > 2 || g.ho != g.out_stride * g.hi || g.wo != g.out_stride * g.wi || g.oy0 < 0 || g.ox0 < 0 ||
        g.oy0 >= g.out_stride || g.ox0 >= g.out_stride) return p;
    if (g.ci % 256 != 0 || g.co % 256 != 0 || kpad != g.ci || wld != g.co || g.ldi % 8 != 0 || g.ldo % 8 != 0 || g.ldi < g.ci || g.ldo < g.co) return p;
    if (g.wi < 64 || g.hi < 2 || g.ntaps < 1) return p;
    for (int t = 0; t < g.ntaps; ++t)
        if (g.dy[t] < -1 || g.dy[t] > 1 || g.dx[t] < -1 || g.dx[t] > 1) return p;
    const int64_t M = (int64_t)g.n * g.hi * g.wi;
    if ((M + 2 * g.wi + 256) * g.ldi * 2 >= ((int64_t)1 << 31) || ((int64_t)g.n * g.ho * g.wo + 256) * g.ldo * 2 >= ((int64_t)1 << 31)) return p;
    p.dma = 1;
    p.gy = g.ntaps; p.gz = (g.ci / 256) * (g.co / 256);
    p.ok = true;
    wgrad_dma_set_split(p, g, 1, kpad, wld);
    return p;
}

void wgrad_dma_set_split(WgradMfmaPlan& p, const stcd_conv_geom& g, int S, int kpad, int wld) {
    const int64_t M = (int64_t)g.n * g.hi * g.wi;
    S = (int)std::max<int64_t>(1, std::min<int64_t>(S, (M + 63) / 64));
    int64_t L = ((M + S - 1) / S + 63) / 64 * 64;
    S = (int)((M + L - 1) / L);                          // no empty split
    p.gx = S;
    p.slab_floats = (int64_t)S * g.ntaps * kpad * wld;
}

WgradJob wgrad_dma_make_job(const stcd_conv_geom& g, const WgradMfmaPlan& p, int64_t in_off, int64_t dout_off, int64_t slab_off,
                            int kpad, int wld) {
    WgradJob a;
    memset(&a, 0, sizeof(a));
    a.g = g;
    a.in_off = in_off; a.dout_off = dout_off; a.slab_off = slab_off; a.kpad = kpad; a.wld = wld;
    const int64_t M = (int64_t)g.n * g.hi * g.wi;
    a.ntiles = (int)M;
    a.dma_L = (int)(((M + p.gx - 1) / p.gx + 63) / 64 * 64);
    a.dma_nk = (a.dma_L / 64 + 1) & ~1;
    a.co_valid = g.co;
    a.in_bytes = (unsigned)(M * g.ldi * 2);
    a.dout_bytes = (unsigned)((int64_t)g.n * g.ho * g.wo * g.ldo * 2);
    a.gx = p.gx; a.gy = p.gy; a.gz = p.gz; a.start = 0;
    a.wi_valid = g.wi;
    a.lds_bytes = 8 * WD_HT;
    return a;
}

int launch_wgrad_dma_group(const WgradJob* jobs_dev, int njobs, int total_blocks, const char* base, hipStream_t s) {
    if (njobs <= 0 || total_blocks <= 0) return 0;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_wgrad_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    k_wgrad_dma<<<(unsigned)total_blocks, 512, (size_t)(8 * WD_HT), s>>>(jobs_dev, njobs, base);
    return 0;
}

}  // namespace stcd
