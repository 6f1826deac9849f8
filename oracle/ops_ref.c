/* Plain-C restatement of the per-op arithmetic on the change-detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): compiled by oracle/Makefile into
 * oracle/_build/libstcd_oracle.so and called from tests/ through oracle/ops_c.py.
 * Never linked into, or called by, the product (stcd_amd/).
 *
 * Layout: NCHW fp32, like the reference; sums are carried in double so the oracle is the
 * closest thing to exact arithmetic the tests have.  Every backward is written from the
 * defining formula (no autograd) -- that is the point: it pins the explicit gradients the
 * HIP kernels implement.
 *
 * Follows (semantics, not code):
 *   nn.Conv2d(k=3,p=1)                         /root/reference/models/SiamUnet_diff.py:18-48
 *   nn.ConvTranspose2d(k=3,p=1[,s=2,op=1])     /root/reference/models/SiamUnet_diff.py:52-90
 *   nn.ConvTranspose2d(k=2,s=2), Conv2d(k=1)   /root/reference/models/SNUNet.py:38,106
 *   nn.BatchNorm2d train/eval                  /root/reference/models/SiamUnet_diff.py:19
 *   F.max_pool2d(2,2)                          /root/reference/models/SiamUnet_diff.py:101
 *   abs / sub skip fusion                      SiamUnet_diff.py:150, SiamUnet_sub.py:150
 *   cross_entropy / cd_loss                    /root/reference/models/losses.py:6-21, :24-34
 * Parity status: pinned by tests/golden/g1_ops.npz (tests/test_oracle_c.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX4(n, c, h, w, C, H, W) ((((int64_t)(n) * (C) + (c)) * (H) + (h)) * (W) + (w))

/* y[n,co,oy,ox] = b[co] + sum_{ci,ky,kx} x[n,ci,oy-pad+ky,ox-pad+kx] * w[co,ci,ky,kx]   (stride 1) */
void ref_conv2d_fwd(const float* x, const float* w, const float* b, float* y,
                    int N, int Ci, int H, int W, int Co, int K, int pad) {
    int Ho = H + 2 * pad - K + 1, Wo = W + 2 * pad - K + 1;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    double acc = b ? b[co] : 0.0;
                    for (int ci = 0; ci < Ci; ++ci)
                        for (int ky = 0; ky < K; ++ky) {
                            int iy = oy - pad + ky;
                            if (iy < 0 || iy >= H) continue;
                            for (int kx = 0; kx < K; ++kx) {
                                int ix = ox - pad + kx;
                                if (ix < 0 || ix >= W) continue;
                                acc += (double)x[IDX4(n, ci, iy, ix, Ci, H, W)] *
                                       w[(((int64_t)co * Ci + ci) * K + ky) * K + kx];
                            }
                        }
                    y[IDX4(n, co, oy, ox, Co, Ho, Wo)] = (float)acc;
                }
}

/* dx = full correlation of gy with w; dw[co,ci,ky,kx] = sum x[..] gy[..]; db[co] = sum gy */
void ref_conv2d_bwd(const float* x, const float* w, const float* gy, float* dx, float* dw, float* db,
                    int N, int Ci, int H, int W, int Co, int K, int pad) {
    int Ho = H + 2 * pad - K + 1, Wo = W + 2 * pad - K + 1;
    int64_t nx = (int64_t)N * Ci * H * W, nw = (int64_t)Co * Ci * K * K;
    double* dxa = (double*)calloc(nx, sizeof(double));
    double* dwa = (double*)calloc(nw, sizeof(double));
    for (int co = 0; co < Co; ++co) {
        double sb = 0.0;
        for (int n = 0; n < N; ++n)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    double g = gy[IDX4(n, co, oy, ox, Co, Ho, Wo)];
                    sb += g;
                    for (int ci = 0; ci < Ci; ++ci)
                        for (int ky = 0; ky < K; ++ky) {
                            int iy = oy - pad + ky;
                            if (iy < 0 || iy >= H) continue;
                            for (int kx = 0; kx < K; ++kx) {
                                int ix = ox - pad + kx;
                                if (ix < 0 || ix >= W) continue;
                                int64_t wi = (((int64_t)co * Ci + ci) * K + ky) * K + kx;
                                int64_t xi = IDX4(n, ci, iy, ix, Ci, H, W);
                                dxa[xi] += g * w[wi];
                                dwa[wi] += g * x[xi];
                            }
                        }
                }
        if (db) db[co] = (float)sb;
    }
    if (dx) for (int64_t i = 0; i < nx; ++i) dx[i] = (float)dxa[i];
    if (dw) for (int64_t i = 0; i < nw; ++i) dw[i] = (float)dwa[i];
    free(dxa); free(dwa);
}

/* out[n,co,s*iy-pad+ky, s*ix-pad+kx] += x[n,ci,iy,ix] * w[ci,co,ky,kx];  Ho = (H-1)s - 2pad + K + opad */
void ref_convT2d_fwd(const float* x, const float* w, const float* b, float* y,
                     int N, int Ci, int H, int W, int Co, int K, int stride, int pad, int opad) {
    int Ho = (H - 1) * stride - 2 * pad + K + opad, Wo = (W - 1) * stride - 2 * pad + K + opad;
    int64_t ny = (int64_t)N * Co * Ho * Wo;
    double* ya = (double*)calloc(ny, sizeof(double));
    for (int n = 0; n < N; ++n)
        for (int ci = 0; ci < Ci; ++ci)
            for (int iy = 0; iy < H; ++iy)
                for (int ix = 0; ix < W; ++ix) {
                    double v = x[IDX4(n, ci, iy, ix, Ci, H, W)];
                    for (int co = 0; co < Co; ++co)
                        for (int ky = 0; ky < K; ++ky) {
                            int oy = stride * iy - pad + ky;
                            if (oy < 0 || oy >= Ho) continue;
                            for (int kx = 0; kx < K; ++kx) {
                                int ox = stride * ix - pad + kx;
                                if (ox < 0 || ox >= Wo) continue;
                                ya[IDX4(n, co, oy, ox, Co, Ho, Wo)] +=
                                    v * w[(((int64_t)ci * Co + co) * K + ky) * K + kx];
                            }
                        }
                }
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co)
            for (int64_t p = 0; p < (int64_t)Ho * Wo; ++p) {
                int64_t i = ((int64_t)n * Co + co) * Ho * Wo + p;
                y[i] = (float)(ya[i] + (b ? b[co] : 0.0));
            }
    free(ya);
}

void ref_convT2d_bwd(const float* x, const float* w, const float* gy, float* dx, float* dw, float* db,
                     int N, int Ci, int H, int W, int Co, int K, int stride, int pad, int opad) {
    int Ho = (H - 1) * stride - 2 * pad + K + opad, Wo = (W - 1) * stride - 2 * pad + K + opad;
    int64_t nw = (int64_t)Ci * Co * K * K;
    double* dwa = (double*)calloc(nw, sizeof(double));
    for (int n = 0; n < N; ++n)
        for (int ci = 0; ci < Ci; ++ci)
            for (int iy = 0; iy < H; ++iy)
                for (int ix = 0; ix < W; ++ix) {
                    int64_t xi = IDX4(n, ci, iy, ix, Ci, H, W);
                    double v = x[xi], acc = 0.0;
                    for (int co = 0; co < Co; ++co)
                        for (int ky = 0; ky < K; ++ky) {
                            int oy = stride * iy - pad + ky;
                            if (oy < 0 || oy >= Ho) continue;
                            for (int kx = 0; kx < K; ++kx) {
                                int ox = stride * ix - pad + kx;
                                if (ox < 0 || ox >= Wo) continue;
                                double g = gy[IDX4(n, co, oy, ox, Co, Ho, Wo)];
                                int64_t wi = (((int64_t)ci * Co + co) * K + ky) * K + kx;
                                acc += g * w[wi];
                                dwa[wi] += g * v;
                            }
                        }
                    if (dx) dx[xi] = (float)acc;
                }
    if (dw) for (int64_t i = 0; i < nw; ++i) dw[i] = (float)dwa[i];
    if (db)
        for (int co = 0; co < Co; ++co) {
            double s = 0.0;
            for (int n = 0; n < N; ++n)
                for (int64_t p = 0; p < (int64_t)Ho * Wo; ++p) s += gy[((int64_t)n * Co + co) * Ho * Wo + p];
            db[co] = (float)s;
        }
    free(dwa);
}

/* Train-mode batch norm over (N,H,W).  save_mean/save_invstd out; running stats updated in place with
 * the UNBIASED variance (momentum m).  y = (x-mean)*invstd*gamma + beta, invstd = 1/sqrt(var_biased+eps). */
void ref_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* rmean, float* rvar,
                      float* y, float* save_mean, float* save_invstd,
                      int N, int C, int HW, float momentum, float eps) {
    int64_t cnt = (int64_t)N * HW;
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int n = 0; n < N; ++n)
            for (int p = 0; p < HW; ++p) s += x[((int64_t)n * C + c) * HW + p];
        double mean = s / cnt, v = 0.0;
        for (int n = 0; n < N; ++n)
            for (int p = 0; p < HW; ++p) {
                double d = x[((int64_t)n * C + c) * HW + p] - mean;
                v += d * d;
            }
        double var = v / cnt, invstd = 1.0 / sqrt(var + eps);
        if (save_mean) save_mean[c] = (float)mean;
        if (save_invstd) save_invstd[c] = (float)invstd;
        if (rmean) rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
        if (rvar) rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * (cnt > 1 ? v / (cnt - 1) : var));
        for (int n = 0; n < N; ++n)
            for (int p = 0; p < HW; ++p) {
                int64_t i = ((int64_t)n * C + c) * HW + p;
                y[i] = (float)((x[i] - mean) * invstd * gamma[c] + beta[c]);
            }
    }
}

void ref_bn_eval_fwd(const float* x, const float* gamma, const float* beta, const float* rmean, const float* rvar,
                     float* y, int N, int C, int HW, float eps) {
    for (int c = 0; c < C; ++c) {
        double invstd = 1.0 / sqrt((double)rvar[c] + eps);
        for (int n = 0; n < N; ++n)
            for (int p = 0; p < HW; ++p) {
                int64_t i = ((int64_t)n * C + c) * HW + p;
                y[i] = (float)((x[i] - rmean[c]) * invstd * gamma[c] + beta[c]);
            }
    }
}

/* dbeta = sum gy; dgamma = sum gy*xhat; dx = gamma*invstd*(gy - dbeta/M - xhat*dgamma/M) */
void ref_bn_train_bwd(const float* x, const float* gy, const float* gamma, const float* mean, const float* invstd,
                      float* dx, float* dgamma, float* dbeta, int N, int C, int HW) {
    int64_t cnt = (int64_t)N * HW;
    for (int c = 0; c < C; ++c) {
        double sg = 0.0, sgx = 0.0;
        for (int n = 0; n < N; ++n)
            for (int p = 0; p < HW; ++p) {
                int64_t i = ((int64_t)n * C + c) * HW + p;
                double xh = ((double)x[i] - mean[c]) * invstd[c];
                sg += gy[i];
                sgx += gy[i] * xh;
            }
        if (dbeta) dbeta[c] = (float)sg;
        if (dgamma) dgamma[c] = (float)sgx;
        for (int n = 0; n < N; ++n)
            for (int p = 0; p < HW; ++p) {
                int64_t i = ((int64_t)n * C + c) * HW + p;
                double xh = ((double)x[i] - mean[c]) * invstd[c];
                dx[i] = (float)((double)gamma[c] * invstd[c] * (gy[i] - sg / cnt - xh * sgx / cnt));
            }
    }
}

/* 2x2/2 max pool, floor; arg = index 0..3 (ky*2+kx) of the FIRST maximum in scan order */
void ref_maxpool2_fwd(const float* x, float* y, uint8_t* arg, int N, int C, int H, int W) {
    int Ho = H / 2, Wo = W / 2;
    for (int64_t nc = 0; nc < (int64_t)N * C; ++nc)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                float best = -INFINITY;
                int bi = 0;
                for (int k = 0; k < 4; ++k) {
                    float v = x[(nc * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)];
                    if (k == 0 || v > best) { best = v; bi = k; }
                }
                y[(nc * Ho + oy) * Wo + ox] = best;
                if (arg) arg[(nc * Ho + oy) * Wo + ox] = (uint8_t)bi;
            }
}

void ref_maxpool2_bwd(const float* x, const float* gy, float* dx, int N, int C, int H, int W) {
    int Ho = H / 2, Wo = W / 2;
    memset(dx, 0, sizeof(float) * (size_t)N * C * H * W);
    for (int64_t nc = 0; nc < (int64_t)N * C; ++nc)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                float best = 0.f;
                int bi = 0;
                for (int k = 0; k < 4; ++k) {
                    float v = x[(nc * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)];
                    if (k == 0 || v > best) { best = v; bi = k; }
                }
                dx[(nc * H + 2 * oy + (bi >> 1)) * W + 2 * ox + (bi & 1)] = gy[(nc * Ho + oy) * Wo + ox];
            }
}

/* nn.ReplicationPad2d((0, W-w0, 0, H-h0)) (/root/reference/models/SiamUnet_diff.py:149): x [N,C,h0,w0] -> y [N,C,H,W],
 * y(i,j) = x(min(i,h0-1), min(j,w0-1)); backward: every replica's gradient folds into its source pixel */
void ref_rep_pad_fwd(const float* x, float* y, int N, int C, int h0, int w0, int H, int W) {
    for (int64_t nc = 0; nc < (int64_t)N * C; ++nc)
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j)
                y[(nc * H + i) * W + j] = x[(nc * h0 + (i < h0 ? i : h0 - 1)) * w0 + (j < w0 ? j : w0 - 1)];
}
void ref_rep_pad_bwd(const float* gy, float* dx, int N, int C, int h0, int w0, int H, int W) {
    memset(dx, 0, sizeof(float) * (size_t)N * C * h0 * w0);
    for (int64_t nc = 0; nc < (int64_t)N * C; ++nc)
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j)
                dx[(nc * h0 + (i < h0 ? i : h0 - 1)) * w0 + (j < w0 ? j : w0 - 1)] += gy[(nc * H + i) * W + j];
}

/* mode 0: |a-b| ; mode 1: b-a.   backward: da, db from g (abs' = sign, 0 at ties) */
void ref_fuse_fwd(const float* a, const float* b, float* y, int64_t n, int mode) {
    for (int64_t i = 0; i < n; ++i) y[i] = mode == 0 ? fabsf(a[i] - b[i]) : b[i] - a[i];
}
void ref_fuse_bwd(const float* a, const float* b, const float* g, float* da, float* db, int64_t n, int mode) {
    for (int64_t i = 0; i < n; ++i) {
        float s = mode == 0 ? (float)((a[i] > b[i]) - (a[i] < b[i])) : -1.f;
        da[i] = s * g[i];
        db[i] = -s * g[i];
    }
}

/* mean over non-ignored pixels of -log softmax(logits)[target]; logits [N,C,HW], target int64 [N,HW] */
double ref_ce_fwd_bwd(const float* logits, const int64_t* target, float* dlogits, int N, int C, int HW, int ignore) {
    double total = 0.0;
    int64_t valid = 0;
    for (int n = 0; n < N; ++n)
        for (int p = 0; p < HW; ++p) valid += target[(int64_t)n * HW + p] != ignore;
    for (int n = 0; n < N; ++n)
        for (int p = 0; p < HW; ++p) {
            int64_t t = target[(int64_t)n * HW + p];
            double mx = -INFINITY, se = 0.0;
            for (int c = 0; c < C; ++c) { double v = logits[((int64_t)n * C + c) * HW + p]; if (v > mx) mx = v; }
            for (int c = 0; c < C; ++c) se += exp(logits[((int64_t)n * C + c) * HW + p] - mx);
            double lse = mx + log(se);
            for (int c = 0; c < C; ++c) {
                int64_t i = ((int64_t)n * C + c) * HW + p;
                if (t == ignore) { if (dlogits) dlogits[i] = 0.f; continue; }
                double sm = exp(logits[i] - lse);
                if (dlogits) dlogits[i] = (float)((sm - (c == t)) / valid);
            }
            if (t != ignore) total += lse - logits[((int64_t)n * C + t) * HW + p];
        }
    return total / valid;
}

/* cd_loss(sigmoid(logits), target): BCE(mean, log clamped at -100, grad denominator clamped at 1e-12) + Dice(smooth 1) */
double ref_bce_dice_fwd_bwd(const float* logits, const float* target, float* dlogits, int64_t n) {
    double sp = 0.0, st = 0.0, spt = 0.0, bce = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        float p = 1.f / (1.f + expf(-logits[i]));          /* fp32 sigmoid, as torch computes it */
        double lp = fmax(log((double)p), -100.0), l1p = fmax(log(1.0 - (double)p), -100.0);
        bce -= target[i] * lp + (1.0 - target[i]) * l1p;
        sp += p; st += target[i]; spt += (double)p * target[i];
    }
    bce /= n;
    double den = sp + st + 1.0, num = 2.0 * spt + 1.0;
    if (dlogits)
        for (int64_t i = 0; i < n; ++i) {
            float pf = 1.f / (1.f + expf(-logits[i]));
            double p = pf, q = p * (1.0 - p);
            double dbce = (p - target[i]) / fmax(q, 1e-12) / n;
            double ddice = -(2.0 * target[i] * den - num) / (den * den);
            dlogits[i] = (float)((dbce + ddice) * q);
        }
    return bce + 1.0 - num / den;
}
