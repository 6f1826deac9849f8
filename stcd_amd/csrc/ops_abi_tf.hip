// ops_abi_tf.hip -- per-op C-ABI entry points of the ChangeFormer kernels (include/stcd_hip.h, "stcd_op_cf_*"): each runs exactly the
// launch sequence the engine runs for that step, on caller-provided contiguous NHWC / token tensors, so the parity tests can pin
// every kernel against torch's fp32 implementation of the same op (tests/test_cf_ops_gpu.py).
// Reference semantics: /root/reference/models/ChangeFormer.py (OverlapPatchEmbed :195-236, Attention :298-358, Mlp / DWConv :260-295,
// :512-523, Block :472-509, conv_diff :1138-1148, DecoderTransformer_v3 :1563-1631), ChangeFormerBaseNetworks.py:109-120.
#include "common.h"

using namespace stcd;

#define CF_COMMON(dtype)                                                                                  \
    STCD_CHECK(dtype == STCD_DTYPE_F32 || dtype == STCD_DTYPE_BF16, "unknown dtype");                    \
    hipStream_t s = (hipStream_t)hip_stream

extern "C" {

int64_t stcd_op_cf_scratch_bytes(int64_t rows, int channels, int n, int q_tokens, int kv_tokens) {
    int64_t f = 4096;
    f = std::max(f, layernorm_bwd_scratch_floats(rows, channels));
    f = std::max(f, colsum_scratch_floats(rows, channels));
    f = std::max(f, dwgelu_bwd_scratch_floats(1, 1, (int)std::min<int64_t>(rows, 1 << 30), channels));
    if (n > 0 && q_tokens > 0 && kv_tokens > 0) f = std::max(f, attn_bwd_scratch_floats(n, q_tokens, kv_tokens, 1, channels));
    return f * 4 + 256;
}

int stcd_op_cf_im2col(int dtype, const void* x, void* col, int ldc, int n, int h, int w, int c, int k, int stride, int pad, void* hip_stream) {
    CF_COMMON(dtype);
    STCD_CHECK(x && col && n >= 1 && h >= 1 && w >= 1 && c >= 1 && k >= 1 && stride >= 1 && pad >= 0 && ldc >= c * k * k, "bad argument");
    STCD_CHECK((int64_t)k * k * (c + 2) * (int64_t)dsize(dtype) <= 64 * 1024, "patch too large for the LDS tile");
    const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    launch_im2col(dtype, x, c, col, ldc, n, h, w, c, k, stride, pad, ho, wo, s);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_op_cf_col2im(int dtype, const void* dcol, int ldc, void* dx, int n, int h, int w, int c, int k, int stride, int pad, int accumulate,
                      void* hip_stream) {
    CF_COMMON(dtype);
    STCD_CHECK(dcol && dx && n >= 1 && h >= 1 && w >= 1 && c >= 1 && k >= 1 && stride >= 1 && pad >= 0 && ldc >= c * k * k, "bad argument");
    const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    launch_col2im(dtype, dcol, ldc, dx, c, n, h, w, c, k, stride, pad, ho, wo, accumulate, s);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_op_cf_layernorm(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* stats, int64_t rows, int c, float eps,
                         void* hip_stream) {
    CF_COMMON(dtype);
    STCD_CHECK(x && y && gamma && beta && stats && rows >= 1 && c >= 8 && c % 8 == 0 && c <= 2048, "bad argument");
    launch_layernorm(dtype, x, c, y, c, gamma, beta, stats, rows, c, eps, s);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_op_cf_layernorm_bwd(int dtype, const void* dy, const void* dy2, const void* x, const float* stats, const float* gamma, const void* add,
                             void* dx, float* dgamma, float* dbeta, void* scratch, int64_t rows, int c, void* hip_stream) {
    CF_COMMON(dtype);
    STCD_CHECK(dy && x && stats && gamma && dx && dgamma && dbeta && scratch && rows >= 1 && c >= 8 && c % 8 == 0 && c <= 2048, "bad argument");
    launch_layernorm_bwd(dtype, dy, c, dy2, c, x, c, stats, gamma, add, c, dx, c, dgamma, dbeta, (float*)scratch, rows, c, s);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_op_cf_colsum(int dtype, const void* x, int64_t rows, int c, float* out, void* scratch, void* hip_stream) {
    CF_COMMON(dtype);
    STCD_CHECK(x && out && scratch && rows >= 1 && c >= 8 && c % 8 == 0, "bad argument");
    launch_colsum(dtype, x, c, rows, c, out, (float*)scratch, s);
    STCD_HIP(hipGetLastError());
    return 0;
}
static int attn_check(int n, int q_tokens, int kv_tokens, int heads, int d) {
    STCD_CHECK(n >= 1 && q_tokens >= 1 && kv_tokens >= 1 && heads >= 1 && d >= 8 && d % 8 == 0 && d <= 128, "bad attention shape (head dim: multiple of 8, <= 128)");
    STCD_CHECK((int64_t)n * heads * q_tokens * (int64_t)kv_tokens < ((int64_t)1 << 32), "attention map too large for 32-bit dropout indices");
    return 0;
}
int stcd_op_cf_attention(int dtype, const void* q, const void* kv, void* out, float* lse, int n, int q_tokens, int kv_tokens, int heads, int d,
                         float p, uint64_t seed, void* hip_stream) {
    CF_COMMON(dtype);
    STCD_CHECK(q && kv && out && lse, "null pointer argument");
    if (attn_check(n, q_tokens, kv_tokens, heads, d)) return 1;
    const int C = heads * d;
    launch_attn_fwd(dtype, q, C, kv, 2 * C, out, C, lse, n, q_tokens, kv_tokens, heads, d, 1.f / sqrtf((float)d), cf_make_site(seed, 0, p, true), s);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_op_cf_attention_bwd(int dtype, const void* q, const void* kv, const void* out, const void* dout, const float* lse, void* dq, void* dkv,
                             void* scratch, int n, int q_tokens, int kv_tokens, int heads, int d, float p, uint64_t seed, void* hip_stream) {
    CF_COMMON(dtype);
    STCD_CHECK(q && kv && out && dout && lse && dq && dkv && scratch, "null pointer argument");
    if (attn_check(n, q_tokens, kv_tokens, heads, d)) return 1;
    const int C = heads * d;
    launch_attn_bwd(dtype, q, C, kv, 2 * C, out, C, dout, C, lse, dq, C, dkv, 2 * C, (float*)scratch, n, q_tokens, kv_tokens, heads, d,
                    1.f / sqrtf((float)d), cf_make_site(seed, 0, p, true), s);
    STCD_HIP(hipGetLastError());
    return 0;
}
int64_t stcd_op_cf_attention_scratch_bytes(int n, int q_tokens, int kv_tokens, int heads, int d) {
    return attn_bwd_scratch_floats(n, q_tokens, kv_tokens, heads, d) * 4 + 256;
}
int stcd_op_cf_dwgelu(int dtype, const void* h, void* u, void* a, const float* w, const float* b, int n, int height, int width, int ch, float p,
                      uint64_t seed, void* hip_stream) {
    CF_COMMON(dtype);
    STCD_CHECK(h && u && a && w && b && n >= 1 && height >= 1 && width >= 1 && ch >= 8 && ch % 8 == 0, "bad argument");
    launch_dwgelu_fwd(dtype, h, u, a, w, b, n, height, width, ch, cf_make_site(seed, 0, p, true), s);
    STCD_HIP(hipGetLastError());
    return 0;
}
int64_t stcd_op_cf_dwgelu_scratch_bytes(int n, int height, int width, int ch) { return dwgelu_bwd_scratch_floats(n, height, width, ch) * 4 + 256; }
int stcd_op_cf_dwgelu_bwd(int dtype, const void* h, const void* u, void* da, void* dh, const float* w, float* dw, float* db, void* scratch, int n,
                          int height, int width, int ch, float p, uint64_t seed, void* hip_stream) {
    CF_COMMON(dtype);
    STCD_CHECK(h && u && da && dh && w && dw && db && scratch && n >= 1 && height >= 1 && width >= 1 && ch >= 8 && ch % 8 == 0, "bad argument");
    launch_dwgelu_bwd(dtype, h, u, da, dh, w, dw, db, (float*)scratch, n, height, width, ch, cf_make_site(seed, 0, p, true), s);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_op_cf_resid_drop(int dtype, const void* x, const void* y, void* out, int n, int64_t rows_per_img, int c, float p, float path_p,
                          uint64_t seed, int backward, void* hip_stream) {
    CF_COMMON(dtype);
    STCD_CHECK(y && out && (backward || x) && n >= 1 && rows_per_img >= 1 && c >= 8 && c % 8 == 0, "bad argument");
    const DropSite d0 = cf_make_site(seed, 0, p, true), d1 = cf_make_site(seed, 1, path_p, true);
    if (backward) launch_resid_drop_bwd(dtype, y, out, n, rows_per_img, c, d0, d1, s);
    else launch_resid_drop(dtype, x, y, out, n, rows_per_img, c, d0, d1, s);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_op_cf_bilinear(int dtype, const void* src, void* dst, int n, int h, int w, int out_h, int out_w, int c, int accumulate, int backward,
                        void* hip_stream) {
    CF_COMMON(dtype);
    STCD_CHECK(src && dst && n >= 1 && h >= 1 && w >= 1 && out_h >= 1 && out_w >= 1 && c >= 8 && c % 8 == 0, "bad argument");
    // backward: src = d(dst map) [n, out_h, out_w, c], dst = d(src map) [n, h, w, c]
    if (backward) launch_bilinear_bwd(dtype, src, c, dst, c, n, h, w, out_h, out_w, c, accumulate, s);
    else launch_bilinear(dtype, src, c, dst, c, n, h, w, out_h, out_w, c, accumulate, s);
    STCD_HIP(hipGetLastError());
    return 0;
}
/* op: 0 PReLU(x; alpha) | 1 x * dropout(p, seed) | 2 relu(x) | 3 y-gated relu gradient: x * [y > 0] | 4 alpha_f * x + beta_f * y */
int stcd_op_cf_elementwise(int dtype, int op, const void* x, const void* y, void* out, int64_t rows, int c, const float* alpha, float alpha_f,
                           float beta_f, float p, uint64_t seed, void* hip_stream) {
    CF_COMMON(dtype);
    STCD_CHECK(x && out && rows >= 1 && c >= 8 && c % 8 == 0 && op >= 0 && op <= 4, "bad argument");
    if (op == 0) { STCD_CHECK(alpha != nullptr, "alpha is null"); launch_prelu(dtype, x, c, out, c, alpha, rows, c, s); }
    else if (op == 1) launch_dropout_ew(dtype, x, c, out, c, rows, c, cf_make_site(seed, 0, p, true), s);
    else if (op == 2) launch_relu(dtype, x, c, out, c, rows, c, s);
    else if (op == 3) { STCD_CHECK(y != nullptr, "y is null"); launch_relu_bwd(dtype, x, c, y, c, out, c, rows, c, s); }
    else launch_axpby(dtype, alpha_f, x, c, beta_f, y, c, out, c, rows, c, s);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_op_cf_prelu_bwd(int dtype, const void* dz, const void* y, void* dy, const float* alpha, float* dalpha, void* scratch, int64_t rows, int c,
                         void* hip_stream) {
    CF_COMMON(dtype);
    STCD_CHECK(dz && y && dy && alpha && dalpha && scratch && rows >= 1 && c >= 8 && c % 8 == 0, "bad argument");
    launch_prelu_bwd(dtype, dz, c, y, c, dy, c, alpha, dalpha, (float*)scratch, rows, c, s);
    STCD_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
