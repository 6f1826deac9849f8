// kernels_conv_halo.hip -- 3x3 stride-1 convolutions of WIDE layers on LARGE maps (Ci % 64 == 0, Co % 128 == 0): the
// 256 -> 256 convolutions of ChangeFormer's decoder head at 256^2 / 512^2 (ChangeFormerBaseNetworks.py:109-120 inside
// ChangeFormer.py:1540-1631), forward and data gradient -- 85 % of that model's arithmetic.
//
// Resident HALO, streamed FILTER (the mirror image of k_conv_res, whose resident filter slice of such a layer is 16 output
// channels wide, and the complement of k_conv_gemm, which re-fetches the activation rows for each of the 9 taps):
//   * block = 4 waves, output tile = 16 x 16 pixels x 128 channels; wave wm owns rows 4wm..4wm+3 and all 128 channels:
//     4 x 8 accumulator fragments (16x16x32 bf16 MFMA, weights = A operand, so a lane ends up
//     with 4 consecutive channels of one pixel).
//   * K walks in STAGES (64-channel chunk, tap), tap fastest.  The 18 x 18 x 64-channel halo of a chunk is staged ONCE
//     and serves all 9 taps (and all 128 output channels); the filter stage of one tap (2 k-steps x 8 fragments,
//     16 KB, fragment order of conv_mfma_plan with CiB = 64 -- the image k_conv_gemm reads) is streamed through a
//     three-slot LDS ring.  Per (tap, k-step) a wave reads 4 activation + 8 filter fragments for 32 MFMAs (k_conv_gemm:
//     8 for 16), and the activation crosses L2 -> CU once per chunk instead of once per tap.
//   * software pipeline: the loads of stage s+3 are requested at the start of stage s, parked in a 3-deep LDS ring at the end
//     of stage s+1 (two register sets, one barrier per stage), so the fragments of a stage can be read BEFORE the previous
//     stage's barrier: inside a stage, k-step 1's fragments are requested before the MFMAs of k-step 0 and the next stage's
//     k-step-0 fragments before the MFMAs of k-step 1 (one block per CU = one wave per SIMD: nothing else hides LDS latency).
//     The NEXT chunk's halo travels with the fetches of taps 2..7 (512 16-B pieces each) into the other halo buffer; blocks
//     are persistent over tiles, so the stream never drains inside a launch.  Tiles outside, stages inside: the 128
//     accumulators are plain loop-carried values of the inner loop and stay in AGPRs (as ONE flat stage loop with the
//     epilogue behind an `if`, the allocator moved all 128 to VGPRs and back every stage: 257 moves per 64 MFMAs).
//   * epilogue: bias, then ConvEpi (ReLU, rounding, gate, alpha*v + beta*res) exactly as k_conv_gemm applies it.
//
// MEASURED (MI355X, 4 x 512 x 512 x 256 -> 256): 819 - 830 TFLOP/s alone (k_conv_gemm 877, k_conv_res 720); inside the
// ChangeFormer step 683 vs 761 TFLOP/s for k_conv_gemm on the same 8 launches (step 34.1 vs 33.4 ms) -- so it is OPT-IN
// (STCD_HALO_KERNEL=1) and the GEMM kernel stays the default.  PMC (profiles/r03_conv_halo_pmc.txt): matrix pipes busy 34 %;
// per 64-MFMA stage a wave also issues 106 VALU + 81 SALU instructions (fetch addressing, halo-piece geometry, ring
// bookkeeping) that a single wave per SIMD cannot overlap with its own MFMAs across basic blocks, and waits 34 % of its
// cycles.  What would lift it: LDS-DMA (`buffer_load ... lds`) staging with counted vmcnt instead of register staging (no
// park pass, no second register set, branch-free fetch), which is the 8-phase structure of the programming guide.
#include <cstdlib>

#include "common.h"

namespace stcd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct ConvHaloArgs {
    stcd_conv_geom g;
    const bf16* in; const bf16* wf; const float* bias; bf16* out;
    int NTtot, nchunks;              // n-tiles of the fragment image; Ci / 64
    int tiles_x, tiles_y, ntiles;
    int P, nslices;                  // blocks per output-channel slice (persistent over the tiles bl, bl + P, ...); slices of 128 channels
    unsigned in_bytes;
    int tap[9];                      // (dy << 16) | (dx & 0xffff), order of the fragment image
    int relu;
    const bf16* gate; int ldg;
    const bf16* res; int ldr;
    float alpha, beta;
};

constexpr int HL_HW = 18, HL_BYTES = HL_HW * HL_HW * 128, HL_PIECES = HL_HW * HL_HW * 8;
constexpr int HL_NFB = 8;                                  // n-fragments of a block: 128 output channels
constexpr int HL_WSTAGE = 2 * HL_NFB * 1024;               // one tap's filter stage (2 k-steps x 8 fragments)

__global__ void __launch_bounds__(256, 1)
k_conv_halo(const ConvHaloArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WP = 4, HP = 2;                          // 16-B filter / halo pieces per thread and stage
    char* const halo0 = smem;                              // 2 halo buffers
    char* const wst0 = smem + 2 * HL_BYTES;                // 3 filter stages (ring)
    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    const int q = lane >> 4, r = lane & 15;
    const int P = a.P;
    // block -> (tile stream bl, output-channel slice): the slices of one stream sit 8 block ids apart, i.e. on the same XCD,
    // so the halo both read crosses the fabric once
    int bl, slice;
    if ((P & 7) == 0) { bl = (blockIdx.x / (8 * a.nslices)) * 8 + (blockIdx.x & 7); slice = (blockIdx.x >> 3) % a.nslices; }
    else { bl = blockIdx.x % P; slice = blockIdx.x / P; }
    const int nf0 = slice * HL_NFB;
    const int nchunks = a.nchunks, ntiles = a.ntiles;
    const int tiles_img = a.tiles_x * a.tiles_y;

    // raw buffer loads: the X descriptor starts one row + one pixel before the tensor (offsets relative to a halo's corner are
    // never negative); a piece outside the image gets an offset past num_records and reads zeros
    const int64_t lead = ((int64_t)a.g.wi + 1) * a.g.ldi * 2;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.in)) - lead, (short)0, (int)(a.in_bytes + (unsigned)lead), 0x00020000);
    const unsigned wstep = (unsigned)(2 * a.NTtot * 1024);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.wf)), (short)0, (int)((unsigned)nchunks * 9u * wstep), 0x00020000);

    // filter pieces: i = tid + p*256 -> fragment (i >> 6) = ks*8 + f of the stage, lane i & 63
    unsigned woff[WP];
#pragma unroll
    for (int p = 0; p < WP; ++p) {
        const int i = tid + p * 256, f = (i >> 6) % HL_NFB, ks = (i >> 6) / HL_NFB;
        woff[p] = (unsigned)(((ks * a.NTtot + nf0 + f) * 64 + (i & 63)) * 16);
    }
    const int ch = tid & 7;                                // 16-B chunk of a halo pixel this thread moves (the same for every piece)

    // halo piece `i` of the tile whose first pixel is (y0, x0): global offset past the halo corner + LDS offset (-1: no such piece)
    auto halo_piece = [&](int i, int y0, int x0, unsigned* voff, int* lds) {
        const int pix = i >> 3;
        const int hy = (pix * 3641) >> 16, hx = pix - hy * HL_HW;          // pix / 18 for 0 <= pix < 1024
        const bool in_img = (unsigned)(y0 - 1 + hy) < (unsigned)a.g.hi && (unsigned)(x0 - 1 + hx) < (unsigned)a.g.wi;
        *voff = (i >= 0 && i < HL_PIECES && in_img) ? (unsigned)(((hy * a.g.wi + hx) * a.g.ldi + ch * 8) * 2) : 0x80000000u;
        *lds = (i >= 0 && i < HL_PIECES) ? (pix * 8 + (ch ^ ((hx >> 1) & 7))) * 16 : -1;
    };
    auto tile_coords = [&](int tile, int* n, int* y0, int* x0) {
        const int tn = tile / tiles_img, trem = tile - tn * tiles_img, ty = trem / a.tiles_x;
        *n = tn; *y0 = ty * 16; *x0 = (trem - ty * a.tiles_x) * 16;
    };

    f32x4 acc[4][8];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: the first tile's chunk-0 halo goes straight to buffer 0
    {
        int n0, y0, x0;
        tile_coords(bl, &n0, &y0, &x0);
        const unsigned soff = (unsigned)((((int64_t)n0 * a.g.hi + y0) * a.g.wi + x0) * a.g.ldi * 2);
#pragma unroll
        for (int j0 = 0; j0 < 12; j0 += 4) {
            u32x4 v[4]; int ld[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned vo;
                halo_piece((j0 + j) * 256 + tid, y0, x0, &vo, &ld[j]);
                v[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, soff, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ld[j] >= 0) *reinterpret_cast<uint4*>(halo0 + ld[j]) = make_uint4(v[j][0], v[j][1], v[j][2], v[j][3]);
        }
    }

    // ---- stage streams.  Stage = (tile, 64-channel chunk, tap), tap fastest.  `c*`: the stage being computed; `f*`: the stage
    //      being fetched, THREE ahead (requested in stage s, parked in LDS at the end of stage s + 1, visible after that stage's
    //      barrier, so the fragments of stage s + 3 can be read during stage s + 2 -- before ITS barrier).
    const int nmine = (ntiles - bl + P - 1) / P;
    int ct = 0, cc = 0, ctile = bl, cgc = 0;               // compute stream: tap, chunk, tile, running chunk count (halo buffer = cgc & 1)
    int ft = 0, fc = 0, ftile = bl, fgc = 0;               // fetch stream
    int hy0 = 0, hx0 = 0, hvalid = 0;                      // halo target of the fetch stream (the chunk after its own), set at tap 2
    unsigned hsoff = 0;
    uint4 pwa[WP], pwb[WP], pha[HP], phb[HP];
    int hla[HP], hlb[HP];
#define HL_ADV(T_, C_, TILE_, GC_)                                                                                     \
    do { if (++(T_) == 9) { (T_) = 0; ++(GC_); if (++(C_) == nchunks) { (C_) = 0; (TILE_) += P; } } } while (0)
    // The NEXT chunk's halo travels with the fetches of taps 2..7 (512 pieces each) into the other halo buffer: the earliest
    // of them is parked at the end of the current chunk's stage 0, i.e. after the barrier that ended the last stage reading
    // that buffer; the latest at the end of stage 5.
#define HL_FETCH(PW_, PH_, HL_)                                                                                        \
    do {                                                                                                               \
        const int t_ = __builtin_amdgcn_readfirstlane(ft), c_ = __builtin_amdgcn_readfirstlane(fc);                    \
        const int tile_ = __builtin_amdgcn_readfirstlane(ftile);                                                       \
        const bool live_ = tile_ < ntiles;                                                                             \
        const unsigned ws_ = live_ ? (unsigned)(c_ * 9 + t_) * wstep : 0u;                                             \
        _Pragma("unroll") for (int p = 0; p < WP; ++p) {                                                               \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(wrs, woff[p], ws_, 0);                              \
            PW_[p] = make_uint4(v_[0], v_[1], v_[2], v_[3]);                                                           \
        }                                                                                                              \
        if (t_ == 2) {                                                                                                 \
            int htile_ = tile_, hc_ = c_ + 1;                                                                          \
            if (hc_ == nchunks) { hc_ = 0; htile_ += P; }                                                              \
            hvalid = live_ && htile_ < ntiles;                                                                         \
            int hn_ = 0;                                                                                               \
            if (hvalid) tile_coords(htile_, &hn_, &hy0, &hx0);                                                         \
            hsoff = (unsigned)(((((int64_t)hn_ * a.g.hi + hy0) * a.g.wi + hx0) * a.g.ldi + hc_ * 64) * 2);            \
        }                                                                                                              \
        const bool carry_ = hvalid && t_ >= 2 && t_ <= 7;                                                              \
        _Pragma("unroll") for (int h = 0; h < HP; ++h) {                                                               \
            unsigned vo_; int ld_;                                                                                     \
            halo_piece((t_ - 2) * 512 + h * 256 + tid, hy0, hx0, &vo_, &ld_);                                          \
            const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(xrs, carry_ ? vo_ : 0x80000000u, hsoff, 0);         \
            PH_[h] = make_uint4(v_[0], v_[1], v_[2], v_[3]);                                                           \
            HL_[h] = (carry_ && ld_ >= 0) ? ((fgc + 1) & 1) * HL_BYTES + ld_ : -1;                                     \
        }                                                                                                              \
        HL_ADV(ft, fc, ftile, fgc);                                                                                    \
    } while (0)
#define HL_STASH(WBUF_, PW_, PH_, HL_)                                                                                 \
    do {                                                                                                               \
        _Pragma("unroll") for (int p = 0; p < WP; ++p)                                                                 \
            *reinterpret_cast<uint4*>(wst0 + (WBUF_) * HL_WSTAGE + (tid + p * 256) * 16) = PW_[p];                     \
        _Pragma("unroll") for (int h = 0; h < HP; ++h)                                                                 \
            if (HL_[h] >= 0) *reinterpret_cast<uint4*>(halo0 + HL_[h]) = PH_[h];                                       \
    } while (0)
    // fragments of k-step KS_ of the stage (tap T_, halo buffer GC_ & 1, filter stage WBUF_): 4 pixel rows + 8 n-fragments
#define HL_LOADF(XF_, WF_, T_, GC_, WBUF_, KS_)                                                                        \
    do {                                                                                                               \
        const int tw_ = a.tap[T_];                                                                                     \
        const int col_ = r + (int)(short)(tw_ & 0xffff) + 1, row_ = 4 * wm + (tw_ >> 16) + 1;                          \
        const char* hb_ = halo0 + ((GC_) & 1) * HL_BYTES + (row_ * HL_HW + col_) * 128 + ((((KS_) * 4 + q) ^ ((col_ >> 1) & 7)) * 16); \
        const char* wb_ = wst0 + (WBUF_) * HL_WSTAGE + ((KS_) * HL_NFB * 64 + lane) * 16;                              \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) XF_[m] = *reinterpret_cast<const bf16x8*>(hb_ + m * (HL_HW * 128)); \
        _Pragma("unroll") for (int n = 0; n < 8; ++n) WF_[n] = *reinterpret_cast<const bf16x8*>(wb_ + n * 1024);       \
    } while (0)
#define HL_MMA(XF_, WF_)                                                                                               \
    do {                                                                                                               \
        _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                                  \
            _Pragma("unroll") for (int n = 0; n < 8; ++n)                                                              \
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF_[n], XF_[m], acc[m][n], 0, 0, 0);               \
    } while (0)
    // the finished tile: lane (q, r) holds channels 4q..4q+3 of fragment n at pixel (row 4wm + m, column r)
#define HL_EPILOGUE()                                                                                                  \
    do {                                                                                                               \
        {                                                                                                              \
            int n_, y0_, x0_;                                                                                          \
            tile_coords(etile_, &n_, &y0_, &x0_);                                                                      \
            const int mx_ = x0_ + r;                                                                                   \
            const int cb0_ = nf0 * 16 + 4 * q;                                                                         \
            _Pragma("unroll") for (int n = 0; n < 8; ++n) {                                                            \
                const int cb_ = cb0_ + n * 16;                                                                         \
                float bv_[4] = {0.f, 0.f, 0.f, 0.f};                                                                   \
                if (a.bias) { const float4 b4_ = *reinterpret_cast<const float4*>(a.bias + cb_); bv_[0] = b4_.x; bv_[1] = b4_.y; bv_[2] = b4_.z; bv_[3] = b4_.w; } \
                _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                        \
                    const int my_ = y0_ + 4 * wm + m;                                                                  \
                    float v_[4];                                                                                       \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
                        v_[j] = acc[m][n][j] + bv_[j];                                                                 \
                        if (a.relu) v_[j] = fmaxf(v_[j], 0.f);                                                         \
                    }                                                                                                  \
                    acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};                                                             \
                    if (my_ < a.g.hm && mx_ < a.g.wm) {                                                                \
                        const int64_t opix_ = ((int64_t)n_ * a.g.ho + my_) * a.g.wo + mx_;                             \
                        if (a.gate || a.res) {                                                                         \
                            _Pragma("unroll") for (int j = 0; j < 4; ++j) v_[j] = round_as<bf16>(v_[j]);               \
                            if (a.gate) {                                                                              \
                                const uint2 g_ = *reinterpret_cast<const uint2*>(a.gate + opix_ * a.ldg + cb_);        \
                                const float g4_[4] = {__uint_as_float(g_.x << 16), __uint_as_float(g_.x & 0xffff0000u), \
                                                      __uint_as_float(g_.y << 16), __uint_as_float(g_.y & 0xffff0000u)}; \
                                _Pragma("unroll") for (int j = 0; j < 4; ++j) v_[j] = g4_[j] > 0.f ? v_[j] : 0.f;      \
                            }                                                                                          \
                            if (a.res) {                                                                               \
                                const uint2 r_ = *reinterpret_cast<const uint2*>(a.res + opix_ * a.ldr + cb_);         \
                                const float r4_[4] = {__uint_as_float(r_.x << 16), __uint_as_float(r_.x & 0xffff0000u), \
                                                      __uint_as_float(r_.y << 16), __uint_as_float(r_.y & 0xffff0000u)}; \
                                _Pragma("unroll") for (int j = 0; j < 4; ++j) v_[j] = a.alpha * v_[j] + a.beta * r4_[j]; \
                            }                                                                                          \
                        }                                                                                              \
                        uint2 pk_;                                                                                     \
                        pk_.x = pack_bf16x2(v_[0], v_[1]);                                                             \
                        pk_.y = pack_bf16x2(v_[2], v_[3]);                                                             \
                        *reinterpret_cast<uint2*>(a.out + opix_ * a.g.ldo + cb_) = pk_;                                \
                    }                                                                                                  \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    } while (0)
    // one stage: k-step 1's fragments are requested before the MFMAs of k-step 0, the NEXT stage's k-step-0 fragments before
    // the MFMAs of k-step 1 (its filter stage was parked one stage ago and is visible since the last barrier) -- one wave per
    // SIMD, so nothing else hides the LDS latency
    // (placing the fetch / the park inside the MFMA scheduling regions instead measured the same: 814 - 829 TFLOP/s either way)
#define HL_STAGE(PWF_, PHF_, HLF_, PWS_, PHS_, HLS_)                                                                   \
    do {                                                                                                               \
        HL_FETCH(PWF_, PHF_, HLF_);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        const int t_ = __builtin_amdgcn_readfirstlane(ct);                                                             \
        HL_LOADF(xf1, wf1, t_, cgc, wcur, 1);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        HL_MMA(xf0, wf0);                                                                                              \
        HL_EPILOGUE_PRE();                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        HL_LOADF(xf0, wf0, nt_, ngc_, wnext_, 0);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        HL_MMA(xf1, wf1);                                                                                              \
        HL_ADV(ct, cc, ctile, cgc);                                                                                    \
        HL_STASH(wpark_, PWS_, PHS_, HLS_);                                                                            \
        wcur = wnext_;                                                                                                 \
        barrier_lds();                                                                                                 \
    } while (0)
    // (the tap / halo buffer / filter stage of the stage after the current one)
#define HL_EPILOGUE_PRE()                                                                                              \
    int nt_ = __builtin_amdgcn_readfirstlane(ct) + 1, ngc_ = cgc;                                                      \
    if (nt_ == 9) { nt_ = 0; ++ngc_; }                                                                                 \
    const int wnext_ = wcur == 2 ? 0 : wcur + 1, wpark_ = wnext_ == 2 ? 0 : wnext_ + 1

    bf16x8 xf0[4], wf0[8], xf1[4], wf1[8];
    int wcur = 0;
    HL_FETCH(pwa, pha, hla);                     // stage 0
    HL_STASH(0, pwa, pha, hla);
    HL_FETCH(pwb, phb, hlb);                     // stage 1
    HL_STASH(1, pwb, phb, hlb);
    HL_FETCH(pwa, pha, hla);                     // stage 2 (its tap carries the first 512 pieces of the next chunk's halo)
    __syncthreads();
    HL_LOADF(xf0, wf0, 0, 0, 0, 0);
    // tiles outside, stages inside (an even number per tile: the plan requires an even chunk count), so the accumulators are
    // plain loop-carried values of the inner loop -- zeroed and read out only between tiles
    const int spt = nchunks * 9;
    for (int k = 0; k < nmine; ++k) {
        const int etile_ = ctile;
        for (int s = 0; s < spt; s += 2) {
            HL_STAGE(pwb, phb, hlb, pwa, pha, hla);  // fetches stage s + 3, parks stage s + 2
            HL_STAGE(pwa, pha, hla, pwb, phb, hlb);
        }
        HL_EPILOGUE();
    }
#undef HL_ADV
#undef HL_FETCH
#undef HL_STASH
#undef HL_LOADF
#undef HL_MMA
#undef HL_EPILOGUE
#undef HL_EPILOGUE_PRE
#undef HL_STAGE
}

static bool conv_halo_enabled() {
    static const bool on = [] { const char* e = getenv("STCD_NO_HALO_KERNEL"); return !(e && e[0] == '1'); }();
    return on;
}

ConvHaloPlan conv_halo_plan(const stcd_conv_geom& g, const ConvMfmaPlan& p) {
    ConvHaloPlan hp;
    if (!conv_halo_enabled() || !p.ok || p.modeB || p.CiB != 64 || g.ntaps != 9) return hp;
    if (g.in_stride != 1 || g.out_stride != 1 || g.oy0 != 0 || g.ox0 != 0 || g.hm > g.hi || g.wm > g.wi) return hp;
    if (g.ci % 128 != 0 || g.co % 128 != 0 || g.ldi % 8 != 0 || g.ldo % 4 != 0 || p.NTtot * 16 != g.co) return hp;
    if (((int64_t)g.n * g.hi + 2) * g.wi * g.ldi * 2 >= ((int64_t)1 << 31)) return hp;      // 32-bit buffer offsets
    if ((int64_t)(g.ci / 64) * 9 * 2 * p.NTtot * 1024 >= ((int64_t)1 << 31)) return hp;
    bool seen[9] = {false};
    for (int t = 0; t < 9; ++t) {
        if (g.dy[t] < -1 || g.dy[t] > 1 || g.dx[t] < -1 || g.dx[t] > 1) return hp;
        seen[(g.dy[t] + 1) * 3 + g.dx[t] + 1] = true;
    }
    for (int t = 0; t < 9; ++t) if (!seen[t]) return hp;
    hp.NH = 1;
    hp.nslices = g.co / 128;
    const int64_t ntiles = (int64_t)g.n * ((g.hm + 15) / 16) * ((g.wm + 15) / 16);
    // one block per CU (the LDS holds two halos and three filter stages); every block needs work
    int64_t P = std::max<int64_t>(1, 256 / hp.nslices);
    if (P >= 8) P &= ~(int64_t)7;
    hp.P = (int)std::min<int64_t>(P, ntiles);
    hp.blocks = hp.P * hp.nslices;
    hp.lds_bytes = 2 * HL_BYTES + 3 * HL_WSTAGE;
    hp.ok = true;
    return hp;
}

int launch_conv_halo(const stcd_conv_geom& g, const ConvMfmaPlan& p, const ConvHaloPlan& hp, const void* in, const void* wf,
                     const float* bias, void* out, hipStream_t s, const ConvEpi* epi) {
    if (!hp.ok) return 1;
    ConvHaloArgs a;
    a.g = g;
    a.in = (const bf16*)in; a.wf = (const bf16*)wf; a.bias = bias; a.out = (bf16*)out;
    a.NTtot = p.NTtot; a.nchunks = g.ci / 64;
    a.tiles_x = (g.wm + 15) / 16; a.tiles_y = (g.hm + 15) / 16; a.ntiles = g.n * a.tiles_x * a.tiles_y;
    a.P = hp.P; a.nslices = hp.nslices;
    a.in_bytes = (unsigned)((int64_t)g.n * g.hi * g.wi * g.ldi * 2);
    for (int t = 0; t < 9; ++t) a.tap[t] = (int)(((unsigned)(int)g.dy[t] << 16) | ((unsigned)(int)g.dx[t] & 0xffffu));
    a.relu = epi ? epi->relu : 0;
    a.gate = epi ? (const bf16*)epi->gate : nullptr; a.ldg = epi ? epi->ldg : 0;
    a.res = epi ? (const bf16*)epi->res : nullptr; a.ldr = epi ? epi->ldr : 0;
    a.alpha = epi ? epi->alpha : 1.f; a.beta = epi ? epi->beta : 1.f;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_conv_halo, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    k_conv_halo<<<(unsigned)hp.blocks, 256, (size_t)hp.lds_bytes, s>>>(a);
    return 0;
}

}  // namespace stcd
