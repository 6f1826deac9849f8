// kernels_loss.hip -- fused loss forward+backward and the on-device confusion matrix (gfx950).
//
// Restated from (semantics, not code):
//   cross_entropy(ignore_index=255)          /root/reference/models/losses.py:6-21
//   cd_loss(sigmoid(x), y) == BCE_DICE       /root/reference/models/losses.py:24-34; train_pse_cd.py:227-228,436-462
//   SegmentationMetric.genConfusionMatrix    /root/reference/train_pse_cd.py:361-368
// The reference runs each of these as 3-8 tiny ATen kernels plus a host sync for the metric; here each loss is
// reduce -> finalize -> gradient (three launches, wave-reduced, no host sync) and the metric is one launch.
#include "common.h"

namespace stcd {

#define LOSS_BLOCKS 1024

struct LossScratch {
    double part[LOSS_BLOCKS][5];
    double fin[5];
};
int64_t loss_scratch_bytes() { return (int64_t)sizeof(LossScratch); }

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// block-reduce K doubles (sm: 4 waves x K); result valid in thread 0
template <int K>
__device__ __forceinline__ void block_sum4(double (&v)[K], double* sm) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = wave_sum_d(v[k]);
        if (lane == 0) sm[wid * K + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double a = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += sm[w * K + k];
            v[k] = a;
        }
    }
}

// ------------------------------------------------------------------ cross entropy
__global__ void __launch_bounds__(256)
k_ce_reduce(const float* __restrict__ logits, const int64_t* __restrict__ target, int Cn, int64_t HW, int64_t npix, int ignore,
            LossScratch* sc) {
    __shared__ double sm[16];
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = target[i];
        if (t == ignore) continue;
        // a label outside [0, classes) that is not the ignore index is an ERROR (F.cross_entropy raises a device assert):
        // it is never dereferenced; it poisons the loss (and, below, the gradient) with NaN so the caller cannot miss it
        if (t < 0 || t >= Cn) { v[0] += (double)NAN; v[1] += 1.0; continue; }
        int64_t n = i / HW, p = i - n * HW;
        const float* l = logits + n * Cn * HW + p;
        float mx = l[0];
        for (int c = 1; c < Cn; ++c) mx = fmaxf(mx, l[(int64_t)c * HW]);
        float se = 0.f;
        for (int c = 0; c < Cn; ++c) se += expf(l[(int64_t)c * HW] - mx);
        float lse = mx + logf(se);
        v[0] += (double)(lse - l[t * HW]);
        v[1] += 1.0;
    }
    block_sum4(v, sm);
    if (threadIdx.x == 0) { sc->part[blockIdx.x][0] = v[0]; sc->part[blockIdx.x][1] = v[1]; }
}
__global__ void k_ce_finalize(LossScratch* sc, int nblocks, float* loss) {
    __shared__ double sm[16];
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) { v[0] += sc->part[b][0]; v[1] += sc->part[b][1]; }
    block_sum4(v, sm);
    if (threadIdx.x == 0) {
        sc->fin[0] = v[0]; sc->fin[1] = v[1];
        *loss = (float)(v[0] / v[1]);
    }
}
__global__ void __launch_bounds__(256)
k_ce_grad(const float* __restrict__ logits, const int64_t* __restrict__ target, int Cn, int64_t HW, int64_t npix, int ignore,
          const LossScratch* sc, float* __restrict__ dlogits) {
    const float inv = (float)(1.0 / sc->fin[1]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = target[i];
        int64_t n = i / HW, p = i - n * HW;
        const float* l = logits + n * Cn * HW + p;
        float* d = dlogits + n * Cn * HW + p;
        if (t == ignore) {
            for (int c = 0; c < Cn; ++c) d[(int64_t)c * HW] = 0.f;
            continue;
        }
        if (t < 0 || t >= Cn) {      // out-of-range label: NaN gradient, no out-of-bounds read
            for (int c = 0; c < Cn; ++c) d[(int64_t)c * HW] = NAN;
            continue;
        }
        float mx = l[0];
        for (int c = 1; c < Cn; ++c) mx = fmaxf(mx, l[(int64_t)c * HW]);
        float se = 0.f;
        for (int c = 0; c < Cn; ++c) se += expf(l[(int64_t)c * HW] - mx);
        float rs = 1.f / se;
        for (int c = 0; c < Cn; ++c)
            d[(int64_t)c * HW] = (expf(l[(int64_t)c * HW] - mx) * rs - (c == t ? 1.f : 0.f)) * inv;
    }
}
void launch_loss_ce(const float* logits, const int64_t* target, int B, int Cn, int64_t HW, int ignore, float* loss,
                    float* dlogits, void* scratch, hipStream_t s) {
    LossScratch* sc = (LossScratch*)scratch;
    int64_t npix = (int64_t)B * HW;
    int nb = (int)std::min<int64_t>(LOSS_BLOCKS, (npix + 255) / 256);
    k_ce_reduce<<<nb, 256, 0, s>>>(logits, target, Cn, HW, npix, ignore, sc);
    k_ce_finalize<<<1, 256, 0, s>>>(sc, nb, loss);
    if (dlogits) k_ce_grad<<<nb, 256, 0, s>>>(logits, target, Cn, HW, npix, ignore, sc, dlogits);
}

// ------------------------------------------------------------------ sigmoid + BCE(mean) + Dice(smooth=1)
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void __launch_bounds__(256)
k_bd_reduce(const float* __restrict__ logits, const float* __restrict__ target, int64_t n, int from_logits, LossScratch* sc) {
    __shared__ double sm[20];
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};   // sum p, sum t, sum p*t, sum bce, valid pixels
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float p = (from_logits & 1) ? sigmoidf(logits[i]) : logits[i], t = target[i];
        // targets are probabilities in [0,1]; the cutout label 255 (stcd_pseudo_pair; data/dataset.py:24-57) marks pixels
        // that take no part in the loss -- torch's BCELoss would assert on it.  Any other value outside [0,1]: NaN.
        if (t == 255.f) continue;
        if (!(t >= 0.f && t <= 1.f)) { v[3] += (double)NAN; v[4] += 1.0; continue; }
        float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);   // torch clamps log at -100
        v[0] += p; v[1] += t; v[2] += (double)p * t;
        v[3] -= (double)(t * lp + (1.f - t) * l1p);
        v[4] += 1.0;
    }
    block_sum4(v, sm);
    if (threadIdx.x == 0)
        for (int k = 0; k < 5; ++k) sc->part[blockIdx.x][k] = v[k];
}
__global__ void k_bd_finalize(LossScratch* sc, int nblocks, int64_t n, int flags, float* loss) {
    __shared__ double sm[20];
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
        for (int k = 0; k < 5; ++k) v[k] += sc->part[b][k];
    block_sum4(v, sm);
    if (threadIdx.x == 0) {
        double den = v[0] + v[1] + 1.0, num = 2.0 * v[2] + 1.0;
        sc->fin[0] = den; sc->fin[1] = num; sc->fin[2] = v[4];
        *loss = (float)(((flags & 2) ? 0.0 : v[3] / fmax(v[4], 1.0)) + 1.0 - num / den);      // bit 1: Dice term only
    }
}
__global__ void __launch_bounds__(256)
k_bd_grad(const float* __restrict__ logits, const float* __restrict__ target, int64_t n, int from_logits,
          const LossScratch* sc, float* __restrict__ dlogits) {
    const float den = (float)sc->fin[0], num = (float)sc->fin[1], invn = 1.f / (float)fmax(sc->fin[2], 1.0), invd2 = 1.f / (den * den);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float p = (from_logits & 1) ? sigmoidf(logits[i]) : logits[i], t = target[i];
        if (t == 255.f) { dlogits[i] = 0.f; continue; }
        if (!(t >= 0.f && t <= 1.f)) { dlogits[i] = NAN; continue; }
        float q = p * (1.f - p);
        float dbce = (from_logits & 2) ? 0.f : (p - t) / fmaxf(q, 1e-12f) * invn;   // ATen binary_cross_entropy_backward (EPSILON 1e-12)
        float ddice = -(2.f * t * den - num) * invd2;
        dlogits[i] = (from_logits & 1) ? (dbce + ddice) * q : (dbce + ddice);   // chain through sigmoid only for logits
    }
}
void launch_loss_bce_dice(const float* logits, const float* target, int64_t n, int from_logits, float* loss, float* dlogits,
                          void* scratch, hipStream_t s) {
    LossScratch* sc = (LossScratch*)scratch;
    int nb = (int)std::min<int64_t>(LOSS_BLOCKS, (n + 255) / 256);
    k_bd_reduce<<<nb, 256, 0, s>>>(logits, target, n, from_logits, sc);
    k_bd_finalize<<<1, 256, 0, s>>>(sc, nb, n, from_logits, loss);
    if (dlogits) k_bd_grad<<<nb, 256, 0, s>>>(logits, target, n, from_logits, sc, dlogits);
}

// ------------------------------------------------------------------ contrastive loss of the semi-supervised stage
// /root/reference/train_stcd.py:334-385: pred [2b,1,H,W] probabilities = cat(real change pairs, pseudo change pairs);
// M = (cd_label == pse_label), N = (cd_label != pse_label);
//   loss = sum((pse - cd)^2 * M) / (sum M + 1e-8) + sum((pse - |cd - 1|)^2 * N) / (sum N + 1e-8)
// with gradients to BOTH halves (d|cd-1|/dcd = sign(cd - 1)).
__global__ void __launch_bounds__(256)
k_ct_reduce(const float* __restrict__ pred, const int64_t* __restrict__ cd_label, const int64_t* __restrict__ pse_label, int64_t n,
            LossScratch* sc) {
    __shared__ double sm[16];
    double v[4] = {0.0, 0.0, 0.0, 0.0};   // sum pos, sum M, sum neg, sum N
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float cd = pred[i], ps = pred[n + i];
        if (cd_label[i] == pse_label[i]) { const float d = ps - cd; v[0] += (double)(d * d); v[1] += 1.0; }
        else { const float d = ps - fabsf(cd - 1.f); v[2] += (double)(d * d); v[3] += 1.0; }
    }
    block_sum4(v, sm);
    if (threadIdx.x == 0)
        for (int k = 0; k < 4; ++k) sc->part[blockIdx.x][k] = v[k];
}
__global__ void k_ct_finalize(LossScratch* sc, int nblocks, float* loss) {
    __shared__ double sm[16];
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
        for (int k = 0; k < 4; ++k) v[k] += sc->part[b][k];
    block_sum4(v, sm);
    if (threadIdx.x == 0) {
        sc->fin[0] = v[1] + 1e-8; sc->fin[1] = v[3] + 1e-8;
        *loss = (float)(v[0] / (v[1] + 1e-8) + v[2] / (v[3] + 1e-8));
    }
}
__global__ void __launch_bounds__(256)
k_ct_grad(const float* __restrict__ pred, const int64_t* __restrict__ cd_label, const int64_t* __restrict__ pse_label, int64_t n,
          const LossScratch* sc, float* __restrict__ dpred) {
    const float im = (float)(1.0 / sc->fin[0]), in_ = (float)(1.0 / sc->fin[1]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float cd = pred[i], ps = pred[n + i];
        if (cd_label[i] == pse_label[i]) {
            const float g = 2.f * (ps - cd) * im;
            dpred[n + i] = g; dpred[i] = -g;
        } else {
            const float t = cd - 1.f, g = 2.f * (ps - fabsf(t)) * in_;
            dpred[n + i] = g;
            dpred[i] = -g * (float)((t > 0.f) - (t < 0.f));
        }
    }
}
void launch_loss_contrastive(const float* pred, const int64_t* cd_label, const int64_t* pse_label, int64_t n_half, float* loss,
                             float* dpred, void* scratch, hipStream_t s) {
    LossScratch* sc = (LossScratch*)scratch;
    int nb = (int)std::min<int64_t>(LOSS_BLOCKS, (n_half + 255) / 256);
    k_ct_reduce<<<nb, 256, 0, s>>>(pred, cd_label, pse_label, n_half, sc);
    k_ct_finalize<<<1, 256, 0, s>>>(sc, nb, loss);
    if (dpred) k_ct_grad<<<nb, 256, 0, s>>>(pred, cd_label, pse_label, n_half, sc, dpred);
}

// ------------------------------------------------------------------ 2x2 confusion matrix: cm[2*label+pred] += count
__global__ void __launch_bounds__(256)
k_confusion(const float* __restrict__ logits, const int64_t* __restrict__ target, int Cn, int64_t HW, int64_t npix,
            unsigned long long* __restrict__ cm) {
    __shared__ unsigned int bins[4];
    if (threadIdx.x < 4) bins[threadIdx.x] = 0;
    __syncthreads();
    unsigned int loc[4] = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = target[i];
        if (t < 0 || t > 1) continue;
        int64_t n = i / HW, p = i - n * HW;
        const float* l = logits + n * Cn * HW + p;
        int pred = Cn == 1 ? (l[0] > 0.f) : (l[HW] > l[0]);   // argmax picks class 0 on ties, sigmoid(x) > 0.5 <=> x > 0
        loc[2 * (int)t + pred]++;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned int v = loc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&bins[k], v);
    }
    __syncthreads();
    if (threadIdx.x < 4 && bins[threadIdx.x]) atomicAdd(cm + threadIdx.x, (unsigned long long)bins[threadIdx.x]);
}
void launch_confusion(const float* logits, const int64_t* target, int B, int Cn, int64_t HW, int64_t* cm, hipStream_t s) {
    int64_t npix = (int64_t)B * HW;
    int nb = (int)std::min<int64_t>(512, (npix + 255) / 256);
    k_confusion<<<nb, 256, 0, s>>>(logits, target, Cn, HW, npix, (unsigned long long*)cm);
}

// ------------------------------------------------------------------ fused Adam / AdamW over the flat buffers
// One launch updates every parameter (torch.optim.Adam / AdamW, amsgrad=False, maximize=False; trainer.py:46-50,
// train_pse_cd.py:431): 4 floats per thread, fp32 arithmetic in torch's operation order.
__global__ void __launch_bounds__(256)
k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
       float lr, float omb1, float beta2, float omb2, float eps, float wd, int decoupled, float step_size, float inv_bc2_sqrt) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    float pv[4], gv[4], mv[4], vv[4];
    const bool full = i0 + 3 < n;
    if (full) {
        *reinterpret_cast<float4*>(pv) = *reinterpret_cast<const float4*>(p + i0);
        *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(g + i0);
        *reinterpret_cast<float4*>(mv) = *reinterpret_cast<const float4*>(m + i0);
        *reinterpret_cast<float4*>(vv) = *reinterpret_cast<const float4*>(v + i0);
    } else {
        for (int j = 0; j < 4; ++j) {
            const bool ok = i0 + j < n;
            pv[j] = ok ? p[i0 + j] : 0.f; gv[j] = ok ? g[i0 + j] : 0.f; mv[j] = ok ? m[i0 + j] : 0.f; vv[j] = ok ? v[i0 + j] : 0.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float gr = gv[j];
        if (decoupled) pv[j] *= 1.f - lr * wd;          // AdamW: param.mul_(1 - lr * weight_decay)
        else gr = gr + wd * pv[j];                      // Adam: grad.add(param, alpha=weight_decay)
        mv[j] = mv[j] + (gr - mv[j]) * omb1;            // exp_avg.lerp_(grad, 1 - beta1)
        vv[j] = vv[j] * beta2 + omb2 * gr * gr;          // mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(vv[j]) * inv_bc2_sqrt + eps;
        pv[j] = pv[j] - step_size * (mv[j] / denom);
    }
    if (full) {
        *reinterpret_cast<float4*>(p + i0) = *reinterpret_cast<const float4*>(pv);
        *reinterpret_cast<float4*>(m + i0) = *reinterpret_cast<const float4*>(mv);
        *reinterpret_cast<float4*>(v + i0) = *reinterpret_cast<const float4*>(vv);
    } else {
        for (int j = 0; j < 4; ++j)
            if (i0 + j < n) { p[i0 + j] = pv[j]; m[i0 + j] = mv[j]; v[i0 + j] = vv[j]; }
    }
}
// the same update with its step-dependent scalars read from DEVICE memory (hyper fp32 [8] = lr, 1 - beta1, beta2, 1 - beta2, eps,
// weight decay, step size, 1 / sqrt(bias correction 2)): a captured hipGraph of the training step replays with this step's values,
// which the host writes into a pinned buffer whose copy is part of the graph
__global__ void __launch_bounds__(256)
k_adam_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
           const float* __restrict__ hyper, int decoupled) {
    const float lr = hyper[0], omb1 = hyper[1], beta2 = hyper[2], omb2 = hyper[3], eps = hyper[4], wd = hyper[5], step_size = hyper[6],
                inv_bc2_sqrt = hyper[7];
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    for (int j = 0; j < 4; ++j) {
        const int64_t i = i0 + j;
        if (i >= n) break;
        float pv = p[i], gr = g[i], mv = m[i], vv = v[i];
        if (decoupled) pv *= 1.f - lr * wd;
        else gr = gr + wd * pv;
        mv = mv + (gr - mv) * omb1;
        vv = vv * beta2 + omb2 * gr * gr;
        const float denom = sqrtf(vv) * inv_bc2_sqrt + eps;
        pv = pv - step_size * (mv / denom);
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}
void launch_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, int decoupled, hipStream_t s) {
    const int64_t threads = (n + 3) / 4;
    k_adam_dev<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(p, g, m, v, n, hyper, decoupled);
}
void launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float omb1, float beta2, float omb2, float eps,
                 float wd, int decoupled, float step_size, float inv_bc2_sqrt, hipStream_t s) {
    const int64_t threads = (n + 3) / 4;
    k_adam<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(p, g, m, v, n, lr, omb1, beta2, omb2, eps, wd, decoupled, step_size, inv_bc2_sqrt);
}

// ------------------------------------------------------------------ pseudo-change pair synthesis (own specification)
// The reference assembles a pseudo-change pair from files (data/dataset.py:468-482): B := an in-painted copy of A and
// change-label := A's building mask when the tile is in the change list, else B := A and label := 0; then
// ToTensor + Normalize (dataset.py:499-500) and, optionally, the paired cutout erase (dataset.py:24-57).  No generator
// arithmetic exists there, so this kernel defines it: B = A outside the mask, round(alpha*donor + (1-alpha)*A) inside.
// One thread per pixel: 3 uint8 channels in, 2 x 3 normalised fp32 NCHW values + 3 int64 labels out.
__device__ __forceinline__ uint32_t pc_hash(uint64_t seed, uint64_t i) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + 1);   // splitmix64 (same generator as the dropout masks)
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 40);               // 24 random bits
}
__global__ void __launch_bounds__(256)
k_pseudo_pair(const uint8_t* __restrict__ A, const uint8_t* __restrict__ donor, const uint8_t* __restrict__ mask,
              const uint8_t* __restrict__ change, const float* __restrict__ alpha, const int32_t* __restrict__ erase,
              uint64_t seed, int B, int H, int W, float m0, float m1, float m2, float is0, float is1, float is2,
              float* __restrict__ x1, float* __restrict__ x2, int64_t* __restrict__ c_label, int64_t* __restrict__ s_label_a,
              int64_t* __restrict__ s_label_b) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t HW = (int64_t)H * W;
    if (i >= (int64_t)B * HW) return;
    const int n = (int)(i / HW);
    const int64_t pix = i - (int64_t)n * HW;
    const int y = (int)(pix / W), x = (int)(pix - (int64_t)y * W);
    const bool ch = change[n] != 0;
    const bool m = mask[i] >= 1;                                      // label[label >= 1] = 1   (dataset.py:461)
    const float al = alpha ? alpha[n] : 1.f;
    bool er = false;
    if (erase) {                                                      // (x, y, w, h) per sample; w == 0: no erase
        const int ex = erase[4 * n], ey = erase[4 * n + 1], ew = erase[4 * n + 2], eh = erase[4 * n + 3];
        er = ew > 0 && eh > 0 && x >= ex && x < ex + ew && y >= ey && y < ey + eh;
    }
    const float mean[3] = {m0, m1, m2}, istd[3] = {is0, is1, is2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = (float)A[i * 3 + c];
        float b = a;
        if (ch && m) b = rintf(al * (float)donor[i * 3 + c] + (1.f - al) * a);
        if (er) {                                                     // pixel-level erase, the SAME value in A and B
            a = (float)(pc_hash(seed, (uint64_t)i * 3 + c) & 255u);
            b = a;
        }
        x1[((int64_t)n * 3 + c) * HW + pix] = (a * (1.f / 255.f) - mean[c]) * istd[c];
        x2[((int64_t)n * 3 + c) * HW + pix] = (b * (1.f / 255.f) - mean[c]) * istd[c];
    }
    const int64_t lab = m ? 1 : 0;
    c_label[i] = er ? 255 : (ch ? lab : 0);                           // cutout marks its rectangle 255 (dataset.py:52)
    if (s_label_a) s_label_a[i] = lab;
    if (s_label_b) s_label_b[i] = ch ? 0 : lab;
}
void launch_pseudo_pair(const uint8_t* A, const uint8_t* donor, const uint8_t* mask, const uint8_t* change, const float* alpha,
                        const int32_t* erase, uint64_t seed, int B, int H, int W, const float* mean, const float* std_,
                        float* x1, float* x2, int64_t* c_label, int64_t* s_label_a, int64_t* s_label_b, hipStream_t s) {
    const int64_t total = (int64_t)B * H * W;
    k_pseudo_pair<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(A, donor, mask, change, alpha, erase, seed, B, H, W, mean[0], mean[1],
                                                                  mean[2], 1.f / std_[0], 1.f / std_[1], 1.f / std_[2], x1, x2,
                                                                  c_label, s_label_a, s_label_b);
}

// ------------------------------------------------------------------ photometric augmentation on the device
// Counterpart of the PIL / torchvision host path of the pseudo-change / change datasets
// (/root/reference/data/dataset.py:488-495: T.ColorJitter(0.5, 0.5, 0.5, 0.25) w.p. 0.5, T.RandomGrayscale(p=0.2),
// blur(): GaussianBlur(sigma ~ U(0.1, 2)) w.p. 0.5 (:120-124); then ToTensor + Normalize :499-500), on normalised fp32
// NCHW images that are already resident on the device.  Per image n the caller supplies
//   params[n] = {jitter_on, brightness, contrast, saturation, hue, gray_on, sigma, 0}
// (stcd_amd.augment draws them with the reference's probabilities).  Arithmetic = torchvision's float-tensor functional
// ops (F.adjust_brightness / _contrast / _saturation / _hue, rgb_to_grayscale) in the fixed order brightness -> contrast
// -> saturation -> hue (ColorJitter shuffles that order: this build fixes it, PARITY UNPINNED beyond the per-op formulas,
// which tests pin to PIL's ImageEnhance on the CPU), then RandomGrayscale, then a separable true Gaussian of radius
// ceil(3 sigma) with replicated edges (PIL approximates its GaussianBlur by box filters).
__device__ __forceinline__ float aug_gray(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

__global__ void __launch_bounds__(256)
k_aug_mean(const float* __restrict__ x, const float* __restrict__ params, int64_t HW, float m0, float m1, float m2, float s0, float s1,
           float s2, unsigned long long* __restrict__ acc) {
    const int n = blockIdx.y;
    const float* p = params + n * 8;
    if (p[0] == 0.f) return;                                        // no jitter for this image: the mean is not needed
    const float br = p[1];
    const float* xr = x + (int64_t)n * 3 * HW;
    float sum = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (int64_t)gridDim.x * blockDim.x) {
        const float r = clamp01(br * (xr[i] * s0 + m0)), g = clamp01(br * (xr[HW + i] * s1 + m1)), b = clamp01(br * (xr[2 * HW + i] * s2 + m2));
        sum += aug_gray(r, g, b);
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) atomicAdd(acc + n, (unsigned long long)llrint((double)sum * 1048576.0));   // 2^20 fixed point: exact, order-free
}

__device__ __forceinline__ void aug_hue(float& r, float& g, float& b, float hf) {
    // torchvision _rgb2hsv / _hsv2rgb on floats
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    const float eqc = mx == mn ? 1.f : 0.f, cr = mx - mn;
    const float sat = cr / (eqc != 0.f ? 1.f : mx), crd = eqc != 0.f ? 1.f : cr;
    const float rc = (mx - r) / crd, gc = (mx - g) / crd, bc = (mx - b) / crd;
    float h = (mx == r ? (bc - gc) : 0.f) + ((mx == g && mx != r) ? (2.f + rc - bc) : 0.f) + ((mx != g && mx != r) ? (4.f + gc - rc) : 0.f);
    h = h / 6.f + 1.f;
    h = h - floorf(h);
    h = h + hf;
    h = h - floorf(h);
    const float v = mx, i6 = floorf(h * 6.f), f = h * 6.f - i6;
    const int i = ((int)i6) % 6;
    const float p = clamp01(v * (1.f - sat)), q = clamp01(v * (1.f - sat * f)), t = clamp01(v * (1.f - sat * (1.f - f)));
    switch (i) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

__global__ void __launch_bounds__(256)
k_aug_point(const float* __restrict__ x, const float* __restrict__ params, int64_t HW, float m0, float m1, float m2, float s0, float s1,
            float s2, const unsigned long long* __restrict__ acc, float* __restrict__ out01) {
    const int n = blockIdx.y;
    const float* p = params + n * 8;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float* xr = x + (int64_t)n * 3 * HW;
    float r = xr[i] * s0 + m0, g = xr[HW + i] * s1 + m1, b = xr[2 * HW + i] * s2 + m2;     // back to [0,1] RGB
    if (p[0] != 0.f) {
        const float br = p[1], ct = p[2], st = p[3], hf = p[4];
        r = clamp01(br * r); g = clamp01(br * g); b = clamp01(br * b);
        const float mean = (float)((double)acc[n] / 1048576.0 / (double)HW);
        r = clamp01(ct * r + (1.f - ct) * mean); g = clamp01(ct * g + (1.f - ct) * mean); b = clamp01(ct * b + (1.f - ct) * mean);
        const float gr = aug_gray(r, g, b);
        r = clamp01(st * r + (1.f - st) * gr); g = clamp01(st * g + (1.f - st) * gr); b = clamp01(st * b + (1.f - st) * gr);
        if (hf != 0.f) aug_hue(r, g, b, hf);
    }
    if (p[5] != 0.f) { const float gr = aug_gray(r, g, b); r = g = b = gr; }
    float* o = out01 + (int64_t)n * 3 * HW;
    o[i] = r; o[HW + i] = g; o[2 * HW + i] = b;
}

// one separable pass (dir 0: along x, 1: along y); the last pass also re-normalises
__global__ void __launch_bounds__(256)
k_aug_blur(const float* __restrict__ in, const float* __restrict__ params, int H, int W, int dir, int normalise, float m0, float m1,
           float m2, float is0, float is1, float is2, float* __restrict__ out) {
    const int n = blockIdx.z, c = blockIdx.y;
    const int64_t HW = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float sigma = params[n * 8 + 6];
    const float* pl = in + ((int64_t)n * 3 + c) * HW;
    const int y = (int)(i / W), xq = (int)(i - (int64_t)y * W);
    float v;
    if (sigma > 0.f) {
        const int rad = min(8, (int)ceilf(3.f * sigma));
        const float k = -0.5f / (sigma * sigma);
        float acc_ = 0.f, wsum = 0.f;
        for (int d = -rad; d <= rad; ++d) {
            const float wgt = expf(k * (float)(d * d));
            const int yy = dir ? min(max(y + d, 0), H - 1) : y, xx = dir ? xq : min(max(xq + d, 0), W - 1);
            acc_ += wgt * pl[(int64_t)yy * W + xx];
            wsum += wgt;
        }
        v = acc_ / wsum;
    } else {
        v = pl[i];
    }
    if (normalise) {
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), istd = c == 0 ? is0 : (c == 1 ? is1 : is2);
        v = (v - mean) * istd;
    }
    out[((int64_t)n * 3 + c) * HW + i] = v;
}

void launch_augment(const float* x, const float* params, int N, int H, int W, const float* mean, const float* std_, float* out,
                    void* scratch, hipStream_t s) {
    const int64_t HW = (int64_t)H * W;
    unsigned long long* acc = (unsigned long long*)scratch;                     // [N] fixed-point gray sums
    float* t0 = (float*)((char*)scratch + (((size_t)N * 8 + 255) & ~(size_t)255));
    float* t1 = t0 + (size_t)N * 3 * HW;
    (void)hipMemsetAsync(acc, 0, (size_t)N * 8, s);
    const int chunks = (int)std::min<int64_t>(64, (HW + 1023) / 1024);
    k_aug_mean<<<dim3(chunks, N), 256, 0, s>>>(x, params, HW, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2], acc);
    k_aug_point<<<dim3((unsigned)((HW + 255) / 256), N), 256, 0, s>>>(x, params, HW, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2], acc, t0);
    dim3 gb((unsigned)((HW + 255) / 256), 3, N);
    k_aug_blur<<<gb, 256, 0, s>>>(t0, params, H, W, 0, 0, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, t1);
    k_aug_blur<<<gb, 256, 0, s>>>(t1, params, H, W, 1, 1, mean[0], mean[1], mean[2], 1.f / std_[0], 1.f / std_[1], 1.f / std_[2], out);
}
int64_t augment_scratch_bytes(int N, int H, int W) { return (((int64_t)N * 8 + 255) & ~(int64_t)255) + (int64_t)2 * N * 3 * H * W * 4 + 256; }

}  // namespace stcd
