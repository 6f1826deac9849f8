// ops_abi.hip -- per-op C-ABI entry points of the normalisation / pooling / fusion kernels (include/stcd_hip.h).
//
// These run EXACTLY the launch sequences the engine runs for one layer (statistics -> finalize -> apply, ...), on
// caller-provided NHWC tensors, so the parity tests can pin every elementwise kernel against the reference's per-op
// vectors (tests/golden/g1_ops.npz) and the plain-C oracle at per-op tolerance.  Reference semantics:
//   BatchNorm2d + ReLU + Dropout2d   /root/reference/models/SiamUnet_diff.py:19-20, applied :99
//   F.max_pool2d(2,2)                /root/reference/models/SiamUnet_diff.py:101
//   |T1-T2| / T2-T1 skip fusion      SiamUnet_diff.py:150 / SiamUnet_sub.py:150
//   ReplicationPad2d                 SiamUnet_diff.py:149
#include <algorithm>

#include "common.h"

using namespace stcd;

namespace {

int check_map(const stcd_map_geom* g) {
    STCD_CHECK(g != nullptr, "geometry is null");
    STCD_CHECK(g->n >= 1 && g->h >= 1 && g->w >= 1, "bad sizes");
    STCD_CHECK(g->c >= 8 && g->c % 8 == 0 && g->c <= 2048 && (g->c & (g->c - 1)) == 0, "channels must be a power of two in [8, 2048]");
    STCD_CHECK(g->groups >= 1 && g->groups <= 2 && g->n % g->groups == 0, "groups must be 1 or 2 and divide n");
    STCD_CHECK((int64_t)g->n * g->h * g->w * (g->c / 8) < ((int64_t)1 << 31), "tensor too large for 32-bit indexing");
    return 0;
}

int64_t acc_bytes(const stcd_map_geom& g) { return bn_acc_bytes(g.groups, g.c); }

}  // namespace

extern "C" {

int64_t stcd_op_ew_scratch_bytes(const stcd_map_geom* g) {
    if (!g || g->c < 8 || g->groups < 1) return 0;
    return acc_bytes(*g) + 256;
}

int stcd_op_bn_act(int dtype, const stcd_map_geom* g, const void* y, int ldy, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, const float* mask, int relu, int training, void* a, int lda,
                   void* pool, int ldp, float* stat, void* scratch, int64_t scratch_bytes, void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(dtype == STCD_DTYPE_F32 || dtype == STCD_DTYPE_BF16, "unknown dtype");
    STCD_CHECK(y && gamma && beta && running_mean && running_var && a && stat && scratch, "null pointer argument");
    STCD_CHECK(ldy >= g->c && lda >= g->c && ldy % 8 == 0 && lda % 8 == 0 && (!pool || (ldp >= g->c && ldp % 8 == 0)), "bad pixel stride");
    STCD_CHECK(scratch_bytes >= stcd_op_ew_scratch_bytes(g), "scratch too small");
    hipStream_t s = (hipStream_t)hip_stream;
    const int npg = g->n / g->groups;
    const int64_t ppg = (int64_t)npg * g->h * g->w;
    long long* acc = (long long*)scratch;
    BnActArgs aa;
    if (training) {
        STCD_HIP(hipMemsetAsync(acc, 0, (size_t)acc_bytes(*g), s));
        launch_bn_stats(dtype, y, ldy, g->c, g->groups, ppg, acc, s);
        aa.facc = acc; aa.gamma = gamma; aa.beta = beta; aa.running_mean = running_mean; aa.running_var = running_var;
    } else {
        launch_bn_eval_prepare(g->c, g->groups, gamma, beta, running_mean, running_var, stat, 1e-5f, s);
    }
    aa.Y = y; aa.ldy = ldy; aa.A = a; aa.lda = lda; aa.a_group_off = ppg * lda;
    aa.P = pool; aa.ldp = ldp; aa.stat = stat; aa.mask = mask;
    aa.C = g->c; aa.groups = g->groups; aa.npg = npg; aa.H = g->h; aa.W = g->w; aa.relu = relu ? 1 : 0;
    launch_bn_act(dtype, aa, s);
    STCD_HIP(hipGetLastError());
    return 0;
}

int stcd_op_bn_act_pair(int dtype, const stcd_map_geom* g, const void* y, int ldy, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, const float* mask, int fuse_mode, void* a, int lda, void* pool,
                        int ldp, void* fused, int ldf, float* stat, void* scratch, int64_t scratch_bytes, void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(dtype == STCD_DTYPE_F32 || dtype == STCD_DTYPE_BF16, "unknown dtype");
    STCD_CHECK(g->groups == 2, "the pair kernel handles the two dates of a pair: groups must be 2");
    STCD_CHECK(y && gamma && beta && running_mean && running_var && fused && stat && scratch, "null pointer argument");
    STCD_CHECK(fuse_mode == 0 || fuse_mode == 1, "fuse_mode must be 0 (|a1-a2|) or 1 (a2-a1)");
    STCD_CHECK(scratch_bytes >= stcd_op_ew_scratch_bytes(g), "scratch too small");
    hipStream_t s = (hipStream_t)hip_stream;
    const int npg = g->n / 2;
    const int64_t ppg = (int64_t)npg * g->h * g->w;
    long long* acc = (long long*)scratch;
    STCD_HIP(hipMemsetAsync(acc, 0, (size_t)acc_bytes(*g), s));
    launch_bn_stats(dtype, y, ldy, g->c, 2, ppg, acc, s);
    BnActArgs aa;
    aa.facc = acc; aa.gamma = gamma; aa.beta = beta; aa.running_mean = running_mean; aa.running_var = running_var;
    aa.Y = y; aa.ldy = ldy; aa.A = a; aa.lda = lda; aa.a_group_off = ppg * lda;
    aa.P = pool; aa.ldp = ldp; aa.stat = stat; aa.mask = mask;
    aa.C = g->c; aa.groups = 2; aa.npg = npg; aa.H = g->h; aa.W = g->w; aa.relu = 1;
    launch_bn_act_pair(dtype, aa, fused, ldf, fuse_mode, s);
    STCD_HIP(hipGetLastError());
    return 0;
}

int stcd_op_bn_act_bwd(int dtype, const stcd_map_geom* g, const void* da, int ldda, const void* y, int ldy, const float* stat,
                       const float* mask, int relu, void* dy, int lddy, float* dgamma, float* dbeta, void* scratch,
                       int64_t scratch_bytes, void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(dtype == STCD_DTYPE_F32 || dtype == STCD_DTYPE_BF16, "unknown dtype");
    STCD_CHECK(da && y && stat && dy && dgamma && dbeta && scratch, "null pointer argument");
    STCD_CHECK(scratch_bytes >= stcd_op_ew_scratch_bytes(g), "scratch too small");
    hipStream_t s = (hipStream_t)hip_stream;
    const int npg = g->n / g->groups;
    const int64_t HW = (int64_t)g->h * g->w, ppg = npg * HW;
    long long* acc = (long long*)scratch;
    STCD_HIP(hipMemsetAsync(acc, 0, (size_t)acc_bytes(*g), s));
    launch_bn_bwd_reduce(dtype, da, ldda, ppg * ldda, y, ldy, stat, mask, g->c, g->groups, npg, HW, relu ? 1 : 0, acc, s);
    launch_bn_bwd_apply(dtype, da, ldda, ppg * ldda, dy, lddy, y, ldy, stat, acc, dgamma, dbeta, mask, g->c, g->groups, npg, HW,
                        relu ? 1 : 0, s);
    STCD_HIP(hipGetLastError());
    return 0;
}

int stcd_op_maxpool(int dtype, const stcd_map_geom* g, const void* a, int lda, void* pool, int ldp, void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(a && pool, "null pointer argument");
    launch_maxpool(dtype, a, lda, pool, ldp, g->n, g->h, g->w, g->c, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

int stcd_op_maxpool_bwd(int dtype, const stcd_map_geom* g, const void* a, int lda, const void* dpool, int ldp, void* da, int ldda,
                        int accumulate, void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(a && dpool && da, "null pointer argument");
    const int npg = g->n / g->groups;
    const int64_t ppg = (int64_t)npg * g->h * g->w;
    launch_pool_bwd(dtype, a, lda, ppg * lda, dpool, ldp, da, ldda, ppg * ldda, g->groups, npg, g->h, g->w, g->c, accumulate ? 1 : 0,
                    (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

/* 3x3 stride-2 padding-1 max-pool of the ResNet stem (F.max_pool2d(x, 3, 2, 1), models/resnet.py:176) and its gradient through the
 * recorded winners.  g: the INPUT map (h, w even); pool / dpool: [n, h/2, w/2, c]; idx: n*(h/2)*(w/2)*c bytes. */
int stcd_op_maxpool3(int dtype, const stcd_map_geom* g, const void* a, int lda, void* pool, int ldp, void* idx, void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(a && pool && g->h % 2 == 0 && g->w % 2 == 0, "null pointer argument or odd size");
    launch_maxpool3(dtype, a, lda, pool, ldp, g->n, g->h, g->w, g->c, (hipStream_t)hip_stream, (unsigned char*)idx);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_op_maxpool3_bwd(int dtype, const stcd_map_geom* g, const void* idx, const void* dpool, int ldp, void* da, int ldda, void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(idx && dpool && da && g->h % 2 == 0 && g->w % 2 == 0, "null pointer argument or odd size");
    launch_maxpool3_bwd(dtype, (const unsigned char*)idx, dpool, ldp, da, ldda, g->n, g->h, g->w, g->c, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

int stcd_op_fuse(int dtype, int mode, const stcd_map_geom* g, const void* a, int lda, void* d, int ldd, void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(g->groups == 2 && (mode == 0 || mode == 1) && a && d, "bad argument (groups must be 2, mode 0 or 1)");
    const int b = g->n / 2;
    const int64_t HW = (int64_t)g->h * g->w;
    launch_fuse(dtype, mode, a, lda, (int64_t)b * HW * lda, d, ldd, b, HW, g->c, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

int64_t stcd_op_pairdw_scratch_bytes(const stcd_map_geom* g) {
    if (!g || g->n < 2 || g->c < 8) return 0;
    return pairdw_partial_floats(g->n / 2, g->h, g->w, g->c) * 4 + 256;
}
int stcd_op_pairdw(int dtype, const stcd_map_geom* g, const void* a, int lda, const float* w, const float* b, void* out, int ldo,
                   void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(g->groups == 2 && a && w && out, "bad argument (groups must be 2: the two dates stacked in the batch dimension)");
    const int n = g->n / 2;
    launch_pairdw_fwd(dtype, a, lda, (int64_t)n * g->h * g->w * lda, out, ldo, w, b, n, g->h, g->w, g->c, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_op_pairdw_bwd(int dtype, const stcd_map_geom* g, const void* a, int lda, const void* dout, int lddo, const float* w, void* da,
                       int ldda, float* dw, void* scratch, int64_t scratch_bytes, void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(g->groups == 2 && a && dout && w && da && dw && scratch, "bad argument (groups must be 2)");
    STCD_CHECK(scratch_bytes >= stcd_op_pairdw_scratch_bytes(g), "scratch too small");
    const int n = g->n / 2;
    launch_pairdw_bwd_data(dtype, dout, lddo, da, ldda, (int64_t)n * g->h * g->w * ldda, w, n, g->h, g->w, g->c, (hipStream_t)hip_stream);
    launch_pairdw_bwd_filter(dtype, a, lda, (int64_t)n * g->h * g->w * lda, dout, lddo, dw, (float*)scratch, n, g->h, g->w, g->c,
                             (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

int stcd_op_fuse_bwd(int dtype, int mode, const stcd_map_geom* g, const void* a, int lda, const void* dd, int ldd, void* da, int ldda,
                     void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(g->groups == 2 && (mode == 0 || mode == 1) && a && dd && da, "bad argument (groups must be 2, mode 0 or 1)");
    const int b = g->n / 2;
    const int64_t HW = (int64_t)g->h * g->w;
    launch_fuse_bwd(dtype, mode, a, lda, (int64_t)b * HW * lda, dd, ldd, da, ldda, (int64_t)b * HW * ldda, b, HW, g->c,
                    (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

int stcd_op_rep_pad(int dtype, const stcd_map_geom* g, void* d, int ld, int h0, int w0, void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(d && h0 >= 1 && w0 >= 1 && h0 <= g->h && w0 <= g->w, "bad argument");
    launch_rep_pad(dtype, d, ld, g->n, g->h, g->w, h0, w0, g->c, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

int stcd_op_rep_pad_bwd(int dtype, const stcd_map_geom* g, void* dd, int ld, int h0, int w0, void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(dd && h0 >= 1 && w0 >= 1 && h0 <= g->h && w0 <= g->w, "bad argument");
    launch_rep_pad_bwd(dtype, dd, ld, g->n, g->h, g->w, h0, w0, g->c, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

int stcd_op_skip_bwd(int dtype, int mode, const stcd_map_geom* g, const void* a, int lda, const void* y, int ldy, const void* dd,
                     int ldd, const void* dpool, int ldp, const float* stat, const float* mask, void* da, int ldda, void* dy,
                     int lddy, float* dgamma, float* dbeta, void* scratch, int64_t scratch_bytes, void* hip_stream) {
    if (check_map(g)) return 1;
    STCD_CHECK(g->groups == 2 && (mode == 0 || mode == 1), "groups must be 2, mode 0 or 1");
    STCD_CHECK(y && dd && dpool && stat && da && dy && dgamma && dbeta && scratch, "null pointer argument");
    STCD_CHECK(scratch_bytes >= stcd_op_ew_scratch_bytes(g), "scratch too small");
    hipStream_t s = (hipStream_t)hip_stream;
    const int b = g->n / 2;
    STCD_CHECK(a || skip_pair_supported(b, g->h, g->w, g->c), "a == NULL (recompute the activations from y): unsupported shape");
    const int64_t HW = (int64_t)g->h * g->w, ppg = b * HW;
    long long* acc = (long long*)scratch;
    STCD_HIP(hipMemsetAsync(acc, 0, (size_t)acc_bytes(*g), s));
    if (a)
        launch_skip_bwd(dtype, mode, a, lda, ppg * lda, y, ldy, dd, ldd, dpool, ldp, da, ldda, ppg * ldda, stat, mask, b, g->h, g->w,
                        g->c, acc, s);
    else        // the engine's default plan: the forward stored no activations, k_skip_bwd_pair recomputes them from y
        launch_skip_bwd_pair(dtype, mode, y, ldy, dd, ldd, dpool, ldp, da, ldda, ppg * ldda, stat, mask, b, g->h, g->w, g->c, acc, s);
    launch_bn_bwd_apply(dtype, da, ldda, ppg * ldda, dy, lddy, y, ldy, stat, acc, dgamma, dbeta, mask, g->c, 2, b, HW, 1, s);
    STCD_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
