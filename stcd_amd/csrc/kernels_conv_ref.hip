// kernels_conv_ref.hip -- reference (plain FMA) implementation of the engine's generic tap-list convolution.
//
// This is the fp32 parity path and the on-device cross-check of the MFMA kernels (kernels_conv_mfma.hip): one
// thread per (output position, output channel), fp32 accumulation in tap-major / channel-minor order.
// It serves, through the tap list + strides of stcd_conv_geom:
//   nn.Conv2d(k=3,p=1) fwd / dgrad                     /root/reference/models/SiamUnet_diff.py:18-48
//   nn.ConvTranspose2d(k=3,p=1) fwd / dgrad            SiamUnet_diff.py:54-90   (flipped, transposed filter)
//   nn.ConvTranspose2d(k=3,p=1,s=2,op=1) fwd / dgrad   SiamUnet_diff.py:52      (4 sub-pixel phases / stride-2 conv)
//   nn.ConvTranspose2d(k=2,s=2), nn.Conv2d(k=1)        /root/reference/models/SNUNet.py:38,106
#include "common.h"

namespace stcd {

template <typename T, bool NCHW_OUT>
__global__ void __launch_bounds__(256)
k_conv_ref(stcd_conv_geom g, const T* __restrict__ in, const float* __restrict__ w, int kpad, int wld,
           const float* __restrict__ bias, void* __restrict__ out, int64_t total) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int co = (int)(idx % g.co);
    int64_t m = idx / g.co;
    int mx = (int)(m % g.wm); m /= g.wm;
    int my = (int)(m % g.hm);
    int n = (int)(m / g.hm);
    float acc = bias ? bias[co] : 0.f;
    for (int t = 0; t < g.ntaps; ++t) {
        int iy = my * g.in_stride + g.dy[t], ix = mx * g.in_stride + g.dx[t];
        if (iy < 0 || iy >= g.hi || ix < 0 || ix >= g.wi) continue;
        const T* px = in + (((int64_t)n * g.hi + iy) * g.wi + ix) * g.ldi;
        const float* wt = w + ((int64_t)t * kpad) * wld + co;
        for (int c0 = 0; c0 < g.ci; c0 += 8) {
            float v[8];
            load8<T>(px + c0, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j] * round_as<T>(wt[(int64_t)(c0 + j) * wld]);
        }
    }
    int oy = my * g.out_stride + g.oy0, ox = mx * g.out_stride + g.ox0;
    if (NCHW_OUT)
        ((float*)out)[(((int64_t)n * g.co + co) * g.ho + oy) * g.wo + ox] = acc;
    else
        ((T*)out)[(((int64_t)n * g.ho + oy) * g.wo + ox) * g.ldo + co] = (T)acc;
}

void launch_conv_ref(int dt, const stcd_conv_geom& g, const void* in, const float* w, int kpad, int wld,
                     const float* bias, void* out, bool nchw, hipStream_t s) {
    int64_t total = (int64_t)g.n * g.hm * g.wm * g.co;
    if (total == 0) return;
    int grid = (int)((total + 255) / 256);
    if (dt == BF16) {
        if (nchw) k_conv_ref<bf16, true><<<grid, 256, 0, s>>>(g, (const bf16*)in, w, kpad, wld, bias, out, total);
        else k_conv_ref<bf16, false><<<grid, 256, 0, s>>>(g, (const bf16*)in, w, kpad, wld, bias, out, total);
    } else {
        if (nchw) k_conv_ref<float, true><<<grid, 256, 0, s>>>(g, (const float*)in, w, kpad, wld, bias, out, total);
        else k_conv_ref<float, false><<<grid, 256, 0, s>>>(g, (const float*)in, w, kpad, wld, bias, out, total);
    }
}

// dw[t][ci][co] += sum over a chunk of output positions of in(m, tap t)[ci] * dout(m)[co]
#define WG_CHUNK 2048
// ACC = double for the engine's parity path: the chunks meet through atomics, and in double the summation order no
// longer shows in the fp32 result (reproducible gradients run to run); float for the per-op entry point.
template <typename T, typename ACC>
__global__ void __launch_bounds__(256)
k_wgrad_ref(stcd_conv_geom g, const T* __restrict__ in, const T* __restrict__ dout, ACC* __restrict__ dw, int kpad,
            int wld, int64_t positions) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)g.ntaps * g.ci * g.co) return;
    int co = (int)(idx % g.co);
    int ci = (int)((idx / g.co) % g.ci);
    int t = (int)(idx / ((int64_t)g.co * g.ci));
    int64_t m0 = (int64_t)blockIdx.y * WG_CHUNK, m1 = min(positions, m0 + WG_CHUNK);
    ACC acc = 0;
    for (int64_t m = m0; m < m1; ++m) {
        int mx = (int)(m % g.wm);
        int64_t r = m / g.wm;
        int my = (int)(r % g.hm);
        int n = (int)(r / g.hm);
        int iy = my * g.in_stride + g.dy[t], ix = mx * g.in_stride + g.dx[t];
        if (iy < 0 || iy >= g.hi || ix < 0 || ix >= g.wi) continue;
        int oy = my * g.out_stride + g.oy0, ox = mx * g.out_stride + g.ox0;
        float a = (float)in[(((int64_t)n * g.hi + iy) * g.wi + ix) * g.ldi + ci];
        float b = (float)dout[(((int64_t)n * g.ho + oy) * g.wo + ox) * g.ldo + co];
        acc += (ACC)a * (ACC)b;
    }
    atomicAdd(dw + ((int64_t)t * kpad + ci) * wld + co, acc);
}

void launch_wgrad_ref(int dt, const stcd_conv_geom& g, const void* in, const void* dout, float* dw, int kpad, int wld,
                      hipStream_t s) {
    int64_t positions = (int64_t)g.n * g.hm * g.wm;
    int64_t outs = (int64_t)g.ntaps * g.ci * g.co;
    if (positions == 0 || outs == 0) return;
    dim3 grid((unsigned)((outs + 255) / 256), (unsigned)((positions + WG_CHUNK - 1) / WG_CHUNK));
    if (dt == BF16) k_wgrad_ref<bf16, float><<<grid, 256, 0, s>>>(g, (const bf16*)in, (const bf16*)dout, dw, kpad, wld, positions);
    else k_wgrad_ref<float, float><<<grid, 256, 0, s>>>(g, (const float*)in, (const float*)dout, dw, kpad, wld, positions);
}
void launch_wgrad_ref_f64(int dt, const stcd_conv_geom& g, const void* in, const void* dout, double* dw, int kpad, int wld,
                          hipStream_t s) {
    int64_t positions = (int64_t)g.n * g.hm * g.wm;
    int64_t outs = (int64_t)g.ntaps * g.ci * g.co;
    if (positions == 0 || outs == 0) return;
    dim3 grid((unsigned)((outs + 255) / 256), (unsigned)((positions + WG_CHUNK - 1) / WG_CHUNK));
    if (dt == BF16) k_wgrad_ref<bf16, double><<<grid, 256, 0, s>>>(g, (const bf16*)in, (const bf16*)dout, dw, kpad, wld, positions);
    else k_wgrad_ref<float, double><<<grid, 256, 0, s>>>(g, (const float*)in, (const float*)dout, dw, kpad, wld, positions);
}

}  // namespace stcd
