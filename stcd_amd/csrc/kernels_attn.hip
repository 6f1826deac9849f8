// kernels_attn.hip -- spatial-reduction attention of ChangeFormer on the matrix cores (bf16 storage, fp32 accumulation).
// Attention.forward, /root/reference/models/ChangeFormer.py:347-354:  attn = softmax(q k^T * scale) ; attn_drop ; x = attn v.
//
// Shapes (BASELINE.json configs[4], 512 x 512): N = 16384 / 4096 / 1024 / 256 queries per image against Nkv = 256 reduced keys, head
// dimension 64 (80 in stage 3).  The work is tiny next to the decoder's convolutions (~0.2 TFLOP of a 13 TFLOP step) but N x Nkv
// probabilities per head must never touch HBM: the forward keeps one query tile's scores in registers, the backward recomputes them
// from the stored log-sum-exp.
//
// MFMA form: v_mfma_f32_16x16x16_bf16 (lane l: A[row l & 15][k = 4 (l >> 4) + j], B[k = 4 (l >> 4) + j][col l & 15], D[row = 4 (l >> 4) + j]
// [col l & 15]).  A result tile D[key][query] therefore already IS the B operand [k = key][col = query] of the next product: the
// forward computes S^T = K Q^T, turns it into P^T in registers and feeds it straight into O^T = V^T P^T -- no shuffle, no LDS round trip.
// Because the A fragment of X and the B fragment of X^T hold the same registers, swapping the two operands of an MFMA transposes its
// result: the backward gets S and S^T (dP and dP^T) from the same fragments, S for the products that contract over queries (dK, dV),
// S^T for the one that contracts over keys (dQ).
#include <algorithm>

#include "common.h"

namespace stcd {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a_, b_, c_, 0, 0, 0)

static inline unsigned cdiv_u(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

__device__ __forceinline__ s16x4 pack4(float a, float b, float c, float d) {
    uint2 u;
    u.x = pack_bf16x2(a, b);
    u.y = pack_bf16x2(c, d);
    return __builtin_bit_cast(s16x4, u);
}
__device__ __forceinline__ s16x4 ld4(const bf16* p) { return *reinterpret_cast<const s16x4*>(p); }

// ------------------------------------------------------------------------------------------------ forward
// grid (query blocks, heads, images); 4 waves, each walks 16-query tiles; K [Nkv][D + 8] and V^T [D][Nkv16 + 8] of the (image, head)
// stay in LDS for the whole block (Nkv <= 256).
template <int DT>
__global__ void __launch_bounds__(256, 2)
k_attn_fwd_mfma(const bf16* __restrict__ q, int ldq, const bf16* __restrict__ kv, int ldkv, bf16* __restrict__ out, int ldo, float* __restrict__ lse,
                int N, int Nkv, int heads, float scale, DropSite drop) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = 16 * DT, KP = D + 8;
    const int NKT = (Nkv + 15) >> 4, NK16 = NKT * 16, VP = NK16 + 8;
    bf16* Ks = reinterpret_cast<bf16*>(smem);
    bf16* Vt = Ks + NK16 * KP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
    const int h = blockIdx.y, img = blockIdx.z, C = heads * D;
    for (int e = tid; e < NK16 * (D / 8); e += 256) {
        const int key = e / (D / 8), pc = e - key * (D / 8);
        uint4 kk = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (key < Nkv) {
            const bf16* p = kv + ((int64_t)img * Nkv + key) * ldkv + h * D + pc * 8;
            kk = *reinterpret_cast<const uint4*>(p);
            vv = *reinterpret_cast<const uint4*>(p + C);
        }
        *reinterpret_cast<uint4*>(Ks + key * KP + pc * 8) = kk;
        const unsigned short* ve = reinterpret_cast<const unsigned short*>(&vv);
#pragma unroll
        for (int j = 0; j < 8; ++j) reinterpret_cast<unsigned short*>(Vt)[(pc * 8 + j) * VP + key] = ve[j];
    }
    __syncthreads();
    const int ntiles = (N + 15) >> 4;
    const float sl2 = scale * 1.44269504088896340736f;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int qi = tile * 16 + r, qc = qi < N ? qi : N - 1;
        s16x4 qf[DT];
        {
            const bf16* qp = q + ((int64_t)img * N + qc) * ldq + h * D + 4 * g;
#pragma unroll
            for (int ds = 0; ds < DT; ++ds) qf[ds] = ld4(qp + ds * 16);
        }
        f32x4 s[16];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
            if (kt < NKT) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const bf16* kp = Ks + (kt * 16 + r) * KP + 4 * g;
#pragma unroll
                for (int ds = 0; ds < DT; ++ds) acc = MFMA16(ld4(kp + ds * 16), qf[ds], acc);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (kt * 16 + 4 * g + j >= Nkv) acc[j] = -INFINITY;
                    m = fmaxf(m, acc[j]);
                }
                s[kt] = acc;
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
        s16x4 pf[16];
        const uint32_t base = (uint32_t)(((int64_t)(img * heads + h) * N + qc) * Nkv);
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
            if (kt < NKT) {
                float pm[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p = exp2f((s[kt][j] - m) * sl2);
                    l += p;
                    pm[j] = p * cf_keep(base + (uint32_t)(kt * 16 + 4 * g + j), drop);
                }
                pf[kt] = pack4(pm[0], pm[1], pm[2], pm[3]);
            }
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.f / l;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const bf16* vp = Vt + (dt * 16 + r) * VP + 4 * g;
#pragma unroll
            for (int kt = 0; kt < 16; ++kt)
                if (kt < NKT) acc = MFMA16(ld4(vp + kt * 16), pf[kt], acc);
            if (qi < N)
                *reinterpret_cast<s16x4*>(out + ((int64_t)img * N + qi) * ldo + h * D + dt * 16 + 4 * g) = pack4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
        }
        if (g == 0 && qi < N) lse[(int64_t)(img * heads + h) * N + qi] = m * scale + __logf(l);
    }
}

// ------------------------------------------------------------------------------------------------ backward
// row terms D_i = dO_i . O_i per (image, head, query)
__global__ void __launch_bounds__(256)
k_attn_rowdot(const bf16* __restrict__ o, int ldo, const bf16* __restrict__ dout, int lddo, float* __restrict__ Drow, int N, int heads, int d, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // (img, query, head)
    if (idx >= total) return;
    const int h = (int)(idx % heads);
    const int64_t row = idx / heads;                                    // img * N + query
    const bf16* a = o + row * ldo + h * d;
    const bf16* b = dout + row * lddo + h * d;
    float acc = 0.f;
    for (int p = 0; p < d; p += 8) {
        float x[8], y[8];
        load8(a + p, x);
        load8(b + p, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += x[j] * y[j];
    }
    const int64_t img = row / N, qi = row - img * N;
    Drow[(img * heads + h) * N + qi] = acc;
}

// grid (query splits, key blocks of 128, images * heads); wave w owns keys [128 kb + 32 w, + 32); all four waves work on the SAME
// 16-query tile.  Partial results: dq_part [key blocks][n * N][C] fp32 (summed over the key blocks), dkv_part [query splits][n * Nkv][2C].
template <int DT>
__global__ void __launch_bounds__(256, DT <= 4 ? 2 : 1)     // <= 256 registers where they suffice: VGPR-form MFMAs (kernels_conv_mfma.hip, k_conv_small)
k_attn_bwd_mfma(const bf16* __restrict__ q, int ldq, const bf16* __restrict__ kv, int ldkv, const bf16* __restrict__ dout, int lddo,
                const float* __restrict__ lse, const float* __restrict__ Drow, float* __restrict__ dq_part, float* __restrict__ dkv_part, int n, int N,
                int Nkv, int heads, float scale, DropSite drop, int qper) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = 16 * DT, KP = D + 8, TP = 128 + 8, QP = 16 + 8;
    bf16* Ks = reinterpret_cast<bf16*>(smem);                 // [128][KP]
    bf16* Vs = Ks + 128 * KP;                                 // [128][KP]
    bf16* Kt = Vs + 128 * KP;                                 // [D][TP]
    bf16* Qt = Kt + D * TP;                                   // [D][QP]
    bf16* Ot = Qt + D * QP;                                   // [D][QP]  (dO^T)
    float* red = reinterpret_cast<float*>(Ot + D * QP);       // [4 waves][D][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
    const int img = blockIdx.z / heads, h = blockIdx.z - img * heads, C = heads * D;
    const int key0 = blockIdx.y * 128;
    for (int e = tid; e < 128 * (D / 8); e += 256) {
        const int kl = e / (D / 8), pc = e - kl * (D / 8), key = key0 + kl;
        uint4 kk = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (key < Nkv) {
            const bf16* p = kv + ((int64_t)img * Nkv + key) * ldkv + h * D + pc * 8;
            kk = *reinterpret_cast<const uint4*>(p);
            vv = *reinterpret_cast<const uint4*>(p + C);
        }
        *reinterpret_cast<uint4*>(Ks + kl * KP + pc * 8) = kk;
        *reinterpret_cast<uint4*>(Vs + kl * KP + pc * 8) = vv;
        const unsigned short* ke = reinterpret_cast<const unsigned short*>(&kk);
#pragma unroll
        for (int j = 0; j < 8; ++j) reinterpret_cast<unsigned short*>(Kt)[(pc * 8 + j) * TP + kl] = ke[j];
    }
    f32x4 dKt[DT][2], dVt[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) { dKt[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dVt[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int i_begin = blockIdx.x * qper, i_end = i_begin + qper < N ? i_begin + qper : N;
    const float l2e = 1.44269504088896340736f;
    const int64_t hrow = (int64_t)(img * heads + h) * N;
    for (int i0 = i_begin; i0 < i_end; i0 += 16) {
        __syncthreads();                                       // the previous tile's Qt / Ot / red are free
        // stage Q^T and dO^T of the tile: thread -> (query, 8-dim piece)
        for (int e = tid; e < 16 * (D / 8); e += 256) {
            const int qq = e / (D / 8), pc = e - qq * (D / 8);
            const int qi = i0 + qq < N ? i0 + qq : N - 1;
            const uint4 a = *reinterpret_cast<const uint4*>(q + ((int64_t)img * N + qi) * ldq + h * D + pc * 8);
            const uint4 b = *reinterpret_cast<const uint4*>(dout + ((int64_t)img * N + qi) * lddo + h * D + pc * 8);
            const unsigned short* ae = reinterpret_cast<const unsigned short*>(&a);
            const unsigned short* be = reinterpret_cast<const unsigned short*>(&b);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                reinterpret_cast<unsigned short*>(Qt)[(pc * 8 + j) * QP + qq] = ae[j];
                reinterpret_cast<unsigned short*>(Ot)[(pc * 8 + j) * QP + qq] = be[j];
            }
        }
        // row-major fragments straight from global: lane = (query r, dims 4 g ..)
        s16x4 qf[DT], of[DT];
        {
            const int qi = i0 + r < N ? i0 + r : N - 1;
            const bf16* qp = q + ((int64_t)img * N + qi) * ldq + h * D + 4 * g;
            const bf16* op = dout + ((int64_t)img * N + qi) * lddo + h * D + 4 * g;
#pragma unroll
            for (int ds = 0; ds < DT; ++ds) { qf[ds] = ld4(qp + ds * 16); of[ds] = ld4(op + ds * 16); }
        }
        // per-query terms in both layouts: X: queries 4 g + j (key r); Y: query r (keys 4 g + j)
        float Lx[4], Dx[4], Ly, Dy;
        bool vx[4], vy;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qi = i0 + 4 * g + j;
            vx[j] = qi < i_end;
            Lx[j] = vx[j] ? lse[hrow + qi] : 0.f;
            Dx[j] = vx[j] ? Drow[hrow + qi] : 0.f;
        }
        vy = i0 + r < i_end;
        Ly = vy ? lse[hrow + i0 + r] : 0.f;
        Dy = vy ? Drow[hrow + i0 + r] : 0.f;
        __syncthreads();
        f32x4 dQt[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dQt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int kl0 = wave * 32 + kt * 16;               // first local key of the tile
            f32x4 sx = {0.f, 0.f, 0.f, 0.f}, sy = sx, px = sx, py = sx;
            {
                const bf16* kp = Ks + (kl0 + r) * KP + 4 * g;
                const bf16* vp = Vs + (kl0 + r) * KP + 4 * g;
#pragma unroll
                for (int ds = 0; ds < DT; ++ds) {
                    const s16x4 kf = ld4(kp + ds * 16), vf = ld4(vp + ds * 16);
                    sx = MFMA16(qf[ds], kf, sx);               // S   [query 4g+j][key r]
                    sy = MFMA16(kf, qf[ds], sy);               // S^T [key 4g+j][query r]
                    px = MFMA16(of[ds], vf, px);               // dP
                    py = MFMA16(vf, of[ds], py);               // dP^T
                }
            }
            float pmx[4], dsx[4], dsy[4];
            {
                const int keyx = key0 + kl0 + r;
                const bool kvx = keyx < Nkv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = kvx && vx[j];
                    const float p = ok ? exp2f((sx[j] * scale - Lx[j]) * l2e) : 0.f;
                    const float keep = cf_keep((uint32_t)((hrow + i0 + 4 * g + j) * Nkv + keyx), drop);
                    pmx[j] = p * keep;
                    dsx[j] = p * (px[j] * keep - Dx[j]) * scale;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int keyy = key0 + kl0 + 4 * g + j;
                    const bool ok = keyy < Nkv && vy;
                    const float p = ok ? exp2f((sy[j] * scale - Ly) * l2e) : 0.f;
                    const float keep = cf_keep((uint32_t)((hrow + i0 + r) * Nkv + keyy), drop);
                    dsy[j] = p * (py[j] * keep - Dy) * scale;
                }
            }
            const s16x4 pmX = pack4(pmx[0], pmx[1], pmx[2], pmx[3]);      // B [k = query][col = key]
            const s16x4 dsX = pack4(dsx[0], dsx[1], dsx[2], dsx[3]);
            const s16x4 dsY = pack4(dsy[0], dsy[1], dsy[2], dsy[3]);      // B [k = key][col = query]
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const s16x4 otf = ld4(Ot + (dt * 16 + r) * QP + 4 * g);   // A [row = dim][k = query]
                const s16x4 qtf = ld4(Qt + (dt * 16 + r) * QP + 4 * g);
                const s16x4 ktf = ld4(Kt + (dt * 16 + r) * TP + kl0 + 4 * g);   // A [row = dim][k = key]
                dVt[dt][kt] = MFMA16(otf, pmX, dVt[dt][kt]);              // dV^T [dim][key]
                dKt[dt][kt] = MFMA16(qtf, dsX, dKt[dt][kt]);              // dK^T [dim][key]
                dQt[dt] = MFMA16(ktf, dsY, dQt[dt]);                      // dQ^T [dim][query], this wave's 32 keys
            }
        }
        // dQ of the tile = sum over the four waves' key slices
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(wave * D + dt * 16 + 4 * g + j) * 16 + r] = dQt[dt][j];
        __syncthreads();
        for (int e = tid; e < 16 * D; e += 256) {
            const int qq = e / D, dim = e - qq * D;
            if (i0 + qq < i_end) {
                const float v = red[(0 * D + dim) * 16 + qq] + red[(1 * D + dim) * 16 + qq] + red[(2 * D + dim) * 16 + qq] + red[(3 * D + dim) * 16 + qq];
                dq_part[((int64_t)blockIdx.y * n * N + (int64_t)img * N + i0 + qq) * C + h * D + dim] = v;
            }
        }
    }
    // the block's dK / dV slab: lane = (key r of tile kt, dims 4 g + j of tile dt)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = key0 + wave * 32 + kt * 16 + r;
        if (key < Nkv) {
            float* o = dkv_part + (((int64_t)blockIdx.x * n + img) * Nkv + key) * 2 * C + h * D + 4 * g;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                *reinterpret_cast<float4*>(o + dt * 16) = make_float4(dKt[dt][kt][0], dKt[dt][kt][1], dKt[dt][kt][2], dKt[dt][kt][3]);
                *reinterpret_cast<float4*>(o + C + dt * 16) = make_float4(dVt[dt][kt][0], dVt[dt][kt][1], dVt[dt][kt][2], dVt[dt][kt][3]);
            }
        }
    }
}
__global__ void k_sum_slabs_bf16(const float* __restrict__ partial, int nslab, int64_t count, int row_elems, bf16* __restrict__ out, int ldo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    for (int b = 0; b < nslab; ++b) s += partial[(int64_t)b * count + i];
    out[(i / row_elems) * ldo + (i % row_elems)] = (bf16)s;
}

// ------------------------------------------------------------------------------------------------ host side
static inline bool attn_mfma_shape_ok(int Nkv, int d) { return d % 16 == 0 && d >= 16 && d <= 128 && Nkv >= 1 && Nkv <= 256; }
static inline int attn_mfma_qsplit(int n, int heads, int N, int Nkv) {
    const int kb = (Nkv + 127) / 128;
    const int64_t tiles = (N + 15) / 16;
    int64_t qs = std::max<int64_t>(1, 1024 / std::max(1, n * heads * kb));      // ~4 blocks per CU over the launch
    qs = std::min<int64_t>(qs, std::max<int64_t>(1, tiles / 8));                // >= 8 query tiles per block (the K / V staging amortises)
    return (int)std::min<int64_t>(qs, 64);
}
bool attn_mfma_ok(int dt, int Nkv, int d) {
    static const bool off = [] { const char* e = getenv("STCD_NO_MFMA_ATTENTION"); return e && e[0] == '1'; }();
    return dt == BF16 && !off && attn_mfma_shape_ok(Nkv, d);
}
int64_t attn_mfma_bwd_scratch_floats(int n, int N, int Nkv, int heads, int d) {
    const int kb = (Nkv + 127) / 128, C = heads * d;
    return (int64_t)n * heads * N + 64 + (int64_t)kb * n * N * C + (int64_t)attn_mfma_qsplit(n, heads, N, Nkv) * n * Nkv * 2 * C + 64;
}

template <int DT>
static void fwd_launch(const void* q, int ldq, const void* kv, int ldkv, void* out, int ldo, float* lse, int n, int N, int Nkv, int heads,
                       float scale, DropSite drop, hipStream_t s) {
    constexpr int D = 16 * DT;
    const int NK16 = ((Nkv + 15) / 16) * 16;
    const size_t lds = (size_t)(NK16 * (D + 8) + D * (NK16 + 8)) * 2;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_attn_fwd_mfma<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const int64_t tiles = (N + 15) / 16;
    int64_t qb = std::max<int64_t>(1, 1024 / std::max(1, n * heads));
    qb = std::min<int64_t>(qb, std::max<int64_t>(1, tiles / 16));                // each wave >= 4 tiles
    dim3 grid((unsigned)qb, heads, n);
    k_attn_fwd_mfma<DT><<<grid, 256, lds, s>>>((const bf16*)q, ldq, (const bf16*)kv, ldkv, (bf16*)out, ldo, lse, N, Nkv, heads, scale, drop);
}
int launch_attn_fwd_mfma(const void* q, int ldq, const void* kv, int ldkv, void* out, int ldo, float* lse, int n, int N, int Nkv, int heads,
                         int d, float scale, DropSite drop, hipStream_t s) {
    if (!attn_mfma_shape_ok(Nkv, d)) return 1;
    switch (d / 16) {
        case 1: fwd_launch<1>(q, ldq, kv, ldkv, out, ldo, lse, n, N, Nkv, heads, scale, drop, s); break;
        case 2: fwd_launch<2>(q, ldq, kv, ldkv, out, ldo, lse, n, N, Nkv, heads, scale, drop, s); break;
        case 3: fwd_launch<3>(q, ldq, kv, ldkv, out, ldo, lse, n, N, Nkv, heads, scale, drop, s); break;
        case 4: fwd_launch<4>(q, ldq, kv, ldkv, out, ldo, lse, n, N, Nkv, heads, scale, drop, s); break;
        case 5: fwd_launch<5>(q, ldq, kv, ldkv, out, ldo, lse, n, N, Nkv, heads, scale, drop, s); break;
        case 6: fwd_launch<6>(q, ldq, kv, ldkv, out, ldo, lse, n, N, Nkv, heads, scale, drop, s); break;
        case 8: fwd_launch<8>(q, ldq, kv, ldkv, out, ldo, lse, n, N, Nkv, heads, scale, drop, s); break;
        default: return 1;
    }
    return 0;
}

template <int DT>
static void bwd_launch(const void* q, int ldq, const void* kv, int ldkv, const void* dout, int lddo, const float* lse, const float* Drow,
                       float* dq_part, float* dkv_part, int n, int N, int Nkv, int heads, float scale, DropSite drop, int QS, int qper,
                       hipStream_t s) {
    constexpr int D = 16 * DT;
    const size_t lds = (size_t)(2 * 128 * (D + 8) + D * (128 + 8) + 2 * D * 24) * 2 + (size_t)4 * D * 16 * 4;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_attn_bwd_mfma<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    dim3 grid(QS, (Nkv + 127) / 128, n * heads);
    k_attn_bwd_mfma<DT><<<grid, 256, lds, s>>>((const bf16*)q, ldq, (const bf16*)kv, ldkv, (const bf16*)dout, lddo, lse, Drow, dq_part, dkv_part, n,
                                               N, Nkv, heads, scale, drop, qper);
}
int launch_attn_bwd_mfma(const void* q, int ldq, const void* kv, int ldkv, const void* out, int ldo, const void* dout, int lddo,
                         const float* lse, void* dq, int lddq, void* dkv, int lddkv, float* scratch, int n, int N, int Nkv, int heads, int d,
                         float scale, DropSite drop, hipStream_t s) {
    if (!attn_mfma_shape_ok(Nkv, d)) return 1;
    const int C = heads * d, kb = (Nkv + 127) / 128;
    const int QS = attn_mfma_qsplit(n, heads, N, Nkv), qper = (((N + QS - 1) / QS) + 15) & ~15;
    float* Drow = scratch;
    float* dq_part = scratch + (((int64_t)n * heads * N + 63) & ~(int64_t)63);
    float* dkv_part = dq_part + (int64_t)kb * n * N * C;
    const int64_t rows = (int64_t)n * N * heads;
    k_attn_rowdot<<<cdiv_u(rows, 256), 256, 0, s>>>((const bf16*)out, ldo, (const bf16*)dout, lddo, Drow, N, heads, d, rows);
    switch (d / 16) {
#define BWD_CASE(DT_) case DT_: bwd_launch<DT_>(q, ldq, kv, ldkv, dout, lddo, lse, Drow, dq_part, dkv_part, n, N, Nkv, heads, scale, drop, QS, qper, s); break
        BWD_CASE(1); BWD_CASE(2); BWD_CASE(3); BWD_CASE(4); BWD_CASE(5); BWD_CASE(6); BWD_CASE(8);
#undef BWD_CASE
        default: return 1;
    }
    const int64_t cq = (int64_t)n * N * C, ck = (int64_t)n * Nkv * 2 * C;
    k_sum_slabs_bf16<<<cdiv_u(cq, 256), 256, 0, s>>>(dq_part, kb, cq, C, (bf16*)dq, lddq);
    k_sum_slabs_bf16<<<cdiv_u(ck, 256), 256, 0, s>>>(dkv_part, QS, ck, 2 * C, (bf16*)dkv, lddkv);
    return 0;
}

}  // namespace stcd
