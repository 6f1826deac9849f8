// engine.hip -- network plans, workspace layout and the C ABI (include/stcd_hip.h) of the stcd engine.
//
// The engine owns the *graph* of the FC-Siam family (and SNUNet, engine_snunet.hip): buffers are carved out of
// one caller-provided workspace at stcd_configure() time, forward/backward are straight-line launch sequences
// on the caller's stream (no allocation, no sync -> hipGraph-capturable).  Data layout in HBM:
//   activations  NHWC, bf16 (or fp32 in parity mode), T1 images then T2 images in one batch of 2B for the
//                shared encoder (BN statistics stay per date: "groups"),
//   concat       never materialised by a copy: producers write channel slices of the decoder's input buffer,
//   parameters   caller's flat fp32 buffer in the reference's layouts; repacked per call into [tap][K][N],
//   gradients    caller's flat fp32 buffer, same layout as parameters.
// Layer tables restate /root/reference/models/SiamUnet_diff.py:13-92 (+ SiamUnet_conc.py:54-87); the forward
// order follows SiamUnet_diff.py:94-181.
#include <algorithm>
#include <climits>
#include <cstdint>
#include <array>
#include <cstring>
#include <memory>
#include <vector>

#include "common.h"

namespace stcd {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

enum ConvKind { K_CONV3 = 0, K_CONVT3_S1 = 1, K_UPCONV = 2, K_CONV1 = 3, K_CONVT2 = 4, K_CONV3_S2 = 5, K_CONV1_S2 = 6, K_STEM7 = 7 };

struct TRef { int64_t off = -1; int ld = 0; };                 // byte offset in workspace, pixel stride (elements)
struct GRef { int64_t off = -1; int ld = 0; int64_t goff = 0; }; // grouped view (see common.h)

static inline int round8(int c) { return (c + 7) & ~7; }

struct ConvW {
    std::string name;
    int kind = 0, cin = 0, cout = 0;
    int kin_p = 0, nout_p = 0;       // padded K (input channels incl. zero pad) and N
    int64_t w_off = 0, b_off = 0;    // into flat params (elements)
    PackSpec fwd{}, dgrad{};         // dgrad.ntaps == 0: no data gradient needed
    int64_t wpk_fwd = -1, wpk_dgrad = -1, dwe = -1;   // workspace byte offsets
    int64_t dwe_floats = 0;
};
struct BnP {
    std::string name;
    int C = 0, calls = 1;
    int64_t g_off = 0, b_off = 0;    // gamma/beta in flat params
    int64_t run_off = 0;             // running mean (var at +C) in flat bn buffer
};
struct DropP { std::string name; int rows = 0, C = 0; int64_t off = 0; };

// one conv-shaped launch, fully bound at configure time
struct ConvOp {
    stcd_conv_geom g{};
    int conv = -1; bool dgrad = false;   // which packed filter of which ConvW
    int tap0 = 0;                        // first tap inside that filter's PackSpec (sub-pixel phases)
    int kreal = 0, nreal = 0;            // un-padded channels (algorithmic work accounting)
    ConvMfmaPlan plan; int64_t wf = -1;  // MFMA path: plan + fragment-order weight image (workspace offset)
    bool small = false;                  // eligible for the small-channel persistent kernel
    ConvResPlan res; int res_groups = 1; // resident-filter persistent kernel (3x3 stride 1, Ci % 32 == 0)
    ConvGemmPlan gemm;                    // 1x1 convs with Ci % 64 == 0, Co % 64 == 0: tiled GEMM
    ConvDmaPlan dma;                      // wide layers on large maps: 256 x 256 implicit-GEMM tile staged by LDS-DMA (kernels_conv_dma.hip)
    ConvHaloPlan halo;                    // wide 3x3 layers on large maps: resident halo, streamed filter (opt-in, STCD_HALO_KERNEL=1)
};
struct WgradOp {
    stcd_conv_geom g{};
    int conv = -1, tap0 = 0, kreal = 0, nreal = 0;
    WgradMfmaPlan plan;
    int64_t slab = -1;                   // this launch's own K-split slabs (reduced in one batched launch per stage)
    int stage = 0;
    int64_t in_off = -1, dout_off = -1;  // workspace offsets of X and dY (grouped launch at the end of the stage)
    bool grouped = false;
    int group = -1;                      // index of its WgradGroup inside e.wgroups[stage]
    // X is a virtual activation (the producer's raw conv output, transformed while staged: XfSrc / wgrad_job_set_xf); xf_C == 0: plain
    int64_t xf_stat_off = -1, xf_mask_off = -1; int xf_C = 0, xf_groups = 1, xf_npg = 1;
    bool tail = false;                   // the network's first conv: its dY is the LAST tensor of the backward, so it gets a group of
                                         // its own -- its stage-mates' grid goes out as soon as THEY are ready, not with it
    bool own_taps = false;               // filter taps (ky, kx) of this launch given here instead of conv.fwd.ky/kx[tap0 + t]
    int8_t oky[9] = {0}, okx[9] = {0};   // (the 7x7 stem: 49 taps spread over 7 launches)
};
struct WgradGroup {                      // all launches of one kernel variant in one backward stage
    int WCI = 1, NTW = 1; bool t9 = false;
    bool xf = false;                     // its jobs stage X through the virtual-activation transform (k_wgrad_group<.., XF = true>)
    bool tail = false;                   // holds only the network's first conv (WgradOp::tail)
    int gemm = 0;                        // > 0: k_wgrad_gemm<gemm> group (one-tap launches; WCI / NTW / t9 unused)
    int dma = 0;                         // 1: k_wgrad_dma group (256 x 256 channel tiles, LDS-DMA staging; splits chosen for the group)
    std::vector<WgradJob> jobs;
    int total_blocks = 0, lds_bytes = 0;
    int64_t table_off = -1;
    double flops = 0.0, bytes = 0.0;
    int seen = 0; bool launched = false;  // run time: members whose operands exist so far in this backward; the grid went out
};

struct Cbrd {                         // conv -> BN -> ReLU -> Dropout2d [-> pool]
    ConvOp fwd, dgr; WgradOp wg;
    int conv = -1, bn = -1, drop = -1;
    int N = 0, H = 0, W = 0, groups = 1, npg = 0;
    TRef in; int K = 0;               // input view and its channel count
    TRef Y; GRef A; TRef P; TRef dPool; bool pool = false;
    GRef dA; TRef dY; TRef dIn; bool has_dIn = false;
    int64_t stat = -1;
    int64_t facc = -1, bacc = -1;     // BatchNorm accumulators (forward statistics / backward sums), see common.h
    const Cbrd* dgr_bwd = nullptr;    // this layer's data-gradient launch also forms the BatchNorm-backward sums of THAT layer (BwdSum)
    bool bwd_sums_fused = false;      // ... and that layer then skips its k_bn_reduce launch
    int64_t dgr_sum_acc = -1; int dgr_sum_c0 = 0, dgr_sum_C = 0;   // the data-gradient launch also sums a channel slice of its
                                                                    // output (bias gradient of the transposed conv feeding it)
    TRef fuse_dst;                    // skip layers of diff / sub: the decoder's concat slice that receives |a1-a2| / a2-a1
    // "virtual activation" (round 4): virt = this layer's A is never written -- its ONE consumer (the next conv of the stage: forward
    // launch and weight gradient) reads Y and applies scale / shift / ReLU / Dropout2d while staging (no k_bn_act launch, one tensor
    // pass less); xsrc = the producer whose raw output this layer reads that way (nullptr: a materialised input)
    bool virt = false; const Cbrd* xsrc = nullptr;
};
struct XConc {                        // cross_conc skip block of SiamUnet_cross_conc (SiamUnet_crossconc.py:11-33): pairwise depthwise
    int level = 0, C = 0;             // conv -> BN -> ReLU, then conv3x3 C -> C -> BN -> ReLU (the Cbrd `res`, no dropout) into the concat slice
    int64_t w_off = 0, b_off = 0;     // diff.0 weight [C][2][3][3] / bias in the flat parameters
    int bn1 = -1;
    TRef G, dG, R, dR;                // pairwise-conv output (+ gradient), its BN + ReLU (+ gradient)
    int64_t stat1 = -1, facc1 = -1, bacc1 = -1, part = -1;
    Cbrd res;
};
struct UpConv {
    int conv = -1, level = 0;
    int N = 0, h = 0, w = 0, Ho = 0, Wo = 0, C = 0;   // input h x w, concat buffer Ho x Wo
    TRef in, out, dOut, dIn;
    ConvOp fwd[4], dgr; WgradOp wg[4];
    int64_t bias_acc = -1; bool bias_fused = false;
};

// ---- optional event instrumentation: one hipEvent pair around every launch of a kernel class (bench.py's live
//      roofline measurement; off by default, adds nothing to the normal path)
enum ProfClass { PC_CONV = 0, PC_WGRAD = 1, PC_BN_STATS = 2, PC_BN_ACT = 3, PC_BN_BWD_REDUCE = 4, PC_BN_BWD_APPLY = 5,
                 PC_POOL_FUSE = 6, PC_PACK = 7, PC_COUNT = 8 };
struct ProfRec { hipEvent_t a, b; int klass; int name_id; double flops, bytes; };
struct Prof {
    bool on = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    std::vector<std::string> names;      // kernel names as rocprofv3 prints them (substring), interned
    size_t used = 0;
    int intern(const std::string& n) {
        for (size_t i = 0; i < names.size(); ++i)
            if (names[i] == n) return (int)i;
        names.push_back(n);
        return (int)names.size() - 1;
    }
    hipEvent_t get() {
        if (used == pool.size()) { hipEvent_t ev; (void)hipEventCreate(&ev); pool.push_back(ev); }
        return pool[used++];
    }
    ~Prof() { for (auto ev : pool) (void)hipEventDestroy(ev); }
};

// ---- SegCD plan: one generic layer = conv (1x1 / 3x3 / strided / the 7x7 stem) -> BatchNorm [-> + residual] [-> ReLU]
struct ViewRef { int64_t off = 0; int ld = 0; int64_t goff = 0; int gmask = 1; };    // workspace-relative SliceViews entry
struct GLayer {
    std::string name;
    int conv = -1, bn = -1, kind = K_CONV3;
    int N = 0, Hi = 0, Wi = 0, Ho = 0, Wo = 0, K = 0, C = 0, groups = 2, npg = 0;
    bool relu = true;
    TRef in, Y, A, res;                   // res.off < 0: no residual
    TRef dA, dIn, dRes;                   // dA: summed gradient of A (dY is formed in place); dIn: this layer's contribution to
    bool has_dIn = true;                  // d(in); dRes: gated gradient of the residual branch (dZout), off < 0: none
    std::vector<ViewRef> extra_dst;       // decoder concat slices that also receive A
    std::vector<ViewRef> grad_src;        // gradient contributions gathered into dA by the BatchNorm-backward reduction
    bool grad_base = true;                // dA already holds a contribution when the backward of this layer starts
    ConvOp fwd, dgr[4]; int ndgr = 0; WgradOp wg[8]; int nwg = 0;
    int64_t stat = -1, coef = -1, facc = -1, bacc = -1;
    int g_first = 0;                      // BatchNorm group the reference normalises first (running-statistic update order)
};
enum GStepKind { GS_LAYER = 0, GS_MAXPOOL = 1, GS_UPSAMPLE = 2, GS_ABSDIFF = 3 };
// GS_ABSDIFF (FFCTLCD): images [2B, 3B) of `src` (C channels of a [3B, h, w, ld] tensor) = |date 0 - date 1|; backward: the
// gradient of that third group (dsrc) becomes a [2B, h, w, C] contribution buffer (ddst) the producer of the two dates gathers
struct GStep { int kind = GS_LAYER; int layer = -1; TRef src, dst, dsrc, ddst; int N = 0, h = 0, w = 0, C = 0; };

// ---- SNUNet-ECAM plan (SNUNet.py:63-152)
struct SrcSlice { TRef src; TRef dsrc; int C = 0; int prod = -1; int grp = -1; };   // a producer's output inside a consumer's concat
struct NBlock {                                              // conv_block_nested (SNUNet.py:8-26)
    std::string name;
    int c1 = -1, bn1 = -1, c2 = -1, bn2 = -1;
    int N = 0, H = 0, W = 0, groups = 1, npg = 0, Cin = 0, C = 0;
    TRef in, dIn;                                            // (concat) input and its gradient; dIn.off < 0: no data gradient
    std::vector<SrcSlice> srcs;                              // prefix slices copied in; the up-sampled tail is written in place
    int up = -1;
    TRef Y1, A1, Y2, Out, P, dOut, dP, dZ2, dA1;
    bool pool = false;
    int64_t stat1 = -1, stat2 = -1;
    int64_t facc1 = -1, bacc1 = -1, facc2 = -1, bacc2 = -1;
    int64_t d1_sum_acc = -1; int d1_sum_c0 = 0, d1_sum_C = 0;
    std::vector<ViewRef> extra_dst;                          // consumers' concat slices that receive Out (forward)
    std::vector<ViewRef> grad_src;                           // consumers' concat-gradient slices (+ up-sampling paths) summed into dOut
    bool grad_base = false;                                  // dOut already holds a gradient (max-pool / ECAM) before the sum
    ConvOp f1, f2, d1, d2; WgradOp w1, w2;
};
struct SnUp {                                                // up: ConvTranspose2d(C, C, 2, stride=2) (SNUNet.py:29-43)
    int conv = -1, N = 0, h = 0, w = 0, C = 0;
    TRef src, dsrc, out, dOut, tmp;                          // out/dOut: tail slice of the consumer's concat buffers
    ConvOp fwd[4], dgr; WgradOp wg[4];
    int64_t bias_acc = -1; bool bias_fused = false; int coff = 0;
};

struct CfPlan;                                                  // ChangeFormer plan (engine_cf.inl)
struct stcd_engine_impl {
    Prof prof;
    std::shared_ptr<CfPlan> cf;
    std::vector<NBlock> sn_blocks;
    std::vector<SnUp> sn_ups;
    std::vector<int> sn_order;                               // forward order of blocks (index into sn_blocks)
    std::vector<XConc> xc;                                   // SiamUnet_cross_conc: one block per level (index = level 0..3)
    int sn_final = -1;
    int64_t sn_w[4] = {0, 0, 0, 0};                          // ca.fc1, ca.fc2, ca1.fc1, ca1.fc2 offsets in the flat params
    TRef snE, sndE, snZ, sndZ;
    int64_t sn_pool = -1, sn_argm = -1, sn_att = -1, sn_hid = -1, sn_sums = -1, sn_dpool = -1, sn_part = -1;
    int64_t sn_dout_begin = -1, sn_dout_end = -1;
    ConvOp sn_final_fwd, sn_final_dgr; WgradOp sn_final_wg;
    // ---- SegCD (ResNet-50 UNet) plan
    std::vector<GLayer> g_layers;                            // every conv + BatchNorm (+ residual) (+ ReLU) layer, forward order
    std::vector<GStep> g_fwd;                                // forward program (the backward walks it in reverse)
    int g_stem = -1, g_head_conv = -1;
    int debug_flags = 0;
    std::vector<stcd_ws_tensor> ws_tensors;
    std::vector<std::array<int, 4>> g_blocks;                // per residual block: (L1, L2, L3, Ld or -1) bottleneck; (L1, -1, L2, Ld or -1) basic
    int seg_x = 4, seg_layers[4] = {3, 4, 6, 3};             // block expansion (4: Bottleneck, 1: BasicBlock) and blocks per stage
    int seg_dates = 2;                                       // 2: SegCD (both dates batched, per-date BatchNorm groups); 1: UnetSeg
    bool seg_ffc = false;                                    // FFCTLCD: the decoder also runs on |f1 - f2| (a third group)
    TRef g_pool_idc;                                         // identity-branch contribution of layer1.0 to d(max-pool output)
    std::vector<std::array<int, 2>> g_dec;                   // (conv1, conv2) per decoder block
    TRef gP0, gdP0, gX3, gdX3, gFuseTmp;
    int64_t g_pool_idx = -1;                                  // winners of the stem's 3x3 max-pool (bytes)
    int64_t g_stem_part = -1;                                 // k_stem_wgrad's per-chunk partial filters (fp32 / non-MFMA path: fixed-order finish)
    int64_t g_raw3 = -1, g_draw3 = -1;
    ConvOp g_head_fwd, g_head_dgr; WgradOp g_head_wg;
    int arch = 0, in_ch = 3, label = 2, dt = F32;
    int64_t min_bn_count = 0;                                // fewest values per channel any BatchNorm layer of the current plan sees
    float drop_p = 0.2f;
    std::vector<stcd_tensor_info> params;
    int64_t param_floats = 0, enc_param_end = 0;
    std::vector<ConvW> convs;
    std::vector<BnP> bns;
    int64_t bn_floats = 0;
    // shape-bound plan
    bool configured = false, fwd_training = false;
    int B = 0, H = 0, W = 0;
    std::vector<DropP> drops;
    int64_t drop_floats = 0;
    int64_t ws_bytes = 0;
    std::vector<Cbrd> enc, dec;
    std::vector<UpConv> ups;
    int final_conv = -1;
    ConvOp final_fwd, final_dgr; WgradOp final_wg;
    const Cbrd* final_dgr_bwd = nullptr;                // the last decoder layer, when final_dgr also forms its BatchNorm-backward sums
    const Cbrd* final_xsrc = nullptr;   // conv11d reads conv12d's raw output (virtual activation)
    int use_virt = 0;                   // STCD_VIRT_ACT=1: virtual activations (round 4: built, bit-identical, MEASURED 2-3 % SLOWER on the
                                        // headline step -- DESIGN.md section 4 -- so every activation is materialised by default)
    int fc_virt_layers = 0;             // layers of the current FC-Siam plan whose activation is virtual
    int xf_mode = 0;                    // STCD_XF_MODE bit 0: the virtual layers' tables come from a k_bn_finalize launch instead of every
                                        // consumer block's prologue; bit 1: timing experiment (no transform, wrong results)
    std::vector<ConvOp*> conv_ops;      // every ConvOp of the plan (weight-image packing walks this)
    int64_t slab = -1, slab_floats = 0;
    std::vector<WgradOp*> wgrad_ops;                    // every WgradOp of the plan
    std::vector<WgradGroup> wgroups[2];                 // per backward stage: grouped weight-gradient launches
    std::vector<ReduceJob> rjobs[2];                    // per backward stage
    int64_t rjobs_total[2] = {0, 0}, rjobs_off[2] = {-1, -1};
    // batched filter repacking: [0] = forward-only job list (eval), [1] = forward + data-gradient filters (training)
    std::vector<PackJob> jobs[2];
    int64_t jobs_total[2] = {0, 0}, jobs_off[2] = {-1, -1};
    const void* jobs_uploaded_ws = nullptr;
    uint64_t weights_tag = 0, packed_tag = 0;      // stcd_set_weights_tag: skip the repack while the caller vouches for the weights
    const void* packed_ws = nullptr; const void* packed_params = nullptr; int packed_level = -1;
    TRef X0, G, finalIn, dFinalIn;
    int Hs[5] = {0}, Ws[5] = {0};
    TRef D[4], dD[4], P[4], dP[4];
    int64_t zero_begin = -1, zero_end = -1;          // everything a step needs zeroed (accumulators, ...): ONE memset per forward
    int64_t final_bias_acc = -1;
    std::vector<BiasJob> bias_jobs; int64_t bias_jobs_off = -1;
    int64_t masks = -1, dwe_begin = -1, dwe_end = -1, scratch8 = -1;
    int use_mfma = 1, use_small = 1, use_wgroup = 1, wgroup_min_tiles = 8, wgroup_rounds = 1, use_res = 1, use_skip_fused = 1, use_act_fuse = 1, use_gemm = 1, use_skip_recompute = 1, wg_tail_split = 0, use_bwdsum = 1, use_bwdsum_res = 0;
    // FC-Siam backward: the decoder's grouped weight gradients (+ slab reduce, bias finish) run on a low-priority side stream beside
    // the encoder's backward chain on the caller's stream; their grids get 1 / wg_side_div of the planner's block budget so that the
    // chain's blocks find free slots (wgrad_side_stream)
    int wg_side_on = 1, wg_side_div = 4;
    bool wg_early = false; hipStream_t wg_cur_side = nullptr;      // set by backward_fcsiam for the duration of a call (exec_wgrad)
    hipStream_t wg_side = nullptr; hipEvent_t wg_fork = nullptr, wg_join = nullptr; int wg_side_dev = -1;
};

}  // namespace stcd

using namespace stcd;
struct stcd_engine : stcd_engine_impl {};

namespace stcd {

// ------------------------------------------------------------------------------------------ model tables
static const int ENC_STAGE_CONVS[4] = {2, 2, 3, 3};
static const int ENC_C[4] = {16, 32, 64, 128};

static void add_param(stcd_engine& e, const std::string& name, std::initializer_list<int64_t> shape, int64_t* off_out) {
    stcd_tensor_info ti;
    memset(&ti, 0, sizeof(ti));
    snprintf(ti.name, sizeof(ti.name), "%s", name.c_str());
    ti.ndim = (int)shape.size();
    int64_t n = 1;
    int i = 0;
    for (auto s : shape) { ti.shape[i++] = s; n *= s; }
    ti.numel = n;
    ti.offset = e.param_floats;
    *off_out = ti.offset;
    e.param_floats += (n + 3) & ~(int64_t)3;   // keep every tensor 16-B aligned in the flat buffer
    e.params.push_back(ti);
}

static void make_specs(ConvW& c, bool need_dgrad) {
    if (c.kind == K_STEM7) {      // dedicated kernels read / write the reference layout directly: nothing to pack
        c.fwd = PackSpec{}; c.dgrad = PackSpec{};
        c.fwd.ks = 7; c.fwd.K = c.cin; c.fwd.N = c.cout; c.fwd.kpad = c.kin_p; c.fwd.wld = c.nout_p;
        return;
    }
    if (c.kind == K_CONV3_S2) {   // Conv2d(k3, s2, p1): forward = stride-2 gather; data gradient = 4 sub-pixel phases (as K_UPCONV's forward)
        PackSpec& f = c.fwd;
        f.ks = 3; f.K = c.cin; f.N = c.cout; f.kpad = c.kin_p; f.wld = c.nout_p; f.ntaps = 9; f.kn_major = 0;
        {   // taps grouped by the PARITY PLANE of the input pixel they read (row 2m + ky - 1: ky == 1 -> even rows, else odd), planes
            // (0,0),(0,1),(1,0),(1,1) with 1+2+2+4 taps: the weight gradient runs as four stride-1 launches over plane views
            int t = 0;
            for (int py = 0; py < 2; ++py)
                for (int px = 0; px < 2; ++px)
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx)
                            if ((ky != 1) == (py == 1) && (kx != 1) == (px == 1)) { f.ky[t] = ky; f.kx[t] = kx; ++t; }
        }
        PackSpec& d = c.dgrad;
        d.ks = 3; d.K = c.cout; d.N = c.cin; d.kpad = c.nout_p; d.wld = round8(c.cin); d.ntaps = need_dgrad ? 9 : 0; d.kn_major = 1;
        int t = 0;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px)
                for (int dy = 0; dy <= py; ++dy)
                    for (int dx = 0; dx <= px; ++dx) { d.ky[t] = py + 1 - 2 * dy; d.kx[t] = px + 1 - 2 * dx; ++t; }
        return;
    }
    if (c.kind == K_CONV1 || c.kind == K_CONVT2 || c.kind == K_CONV1_S2) {
        const int ks = c.kind == K_CONVT2 ? 2 : 1;
        PackSpec& f = c.fwd;
        f.ks = ks; f.K = c.cin; f.N = c.cout; f.kpad = c.kin_p; f.wld = c.nout_p;
        f.ntaps = ks * ks;
        f.kn_major = c.kind == K_CONVT2;              // ConvTranspose2d weights are [Cin][Cout][k][k]
        for (int t = 0; t < f.ntaps; ++t) { f.ky[t] = t / ks; f.kx[t] = t % ks; }   // K_CONVT2: tap = output phase (py,px)
        PackSpec& d = c.dgrad;
        d.ks = ks; d.K = c.cout; d.N = c.cin; d.kpad = c.nout_p; d.wld = round8(c.cin);
        d.ntaps = need_dgrad ? ks * ks : 0;
        d.kn_major = c.kind != K_CONVT2;
        for (int t = 0; t < ks * ks; ++t) { d.ky[t] = t / ks; d.kx[t] = t % ks; }   // K_CONVT2: tap (dy,dx) of the stride-2 gather
        return;
    }
    PackSpec& f = c.fwd;
    f.ks = 3;
    f.K = c.cin; f.N = c.cout; f.kpad = c.kin_p; f.wld = c.nout_p;
    PackSpec& d = c.dgrad;
    d.ks = 3;
    d.K = c.cout; d.N = c.cin; d.kpad = c.nout_p; d.wld = round8(c.cin);
    d.ntaps = 0;
    if (c.kind == K_CONV3 || c.kind == K_CONVT3_S1) {
        f.ntaps = 9;
        f.kn_major = c.kind == K_CONVT3_S1;       // ConvTranspose2d weights are [Cin][Cout][3][3]
        for (int t = 0; t < 9; ++t) {
            int dy = t / 3 - 1, dx = t % 3 - 1;
            f.ky[t] = c.kind == K_CONV3 ? dy + 1 : 1 - dy;
            f.kx[t] = c.kind == K_CONV3 ? dx + 1 : 1 - dx;
        }
        if (need_dgrad) {
            d.ntaps = 9;
            d.kn_major = c.kind == K_CONV3;
            for (int t = 0; t < 9; ++t) {
                int dy = t / 3 - 1, dx = t % 3 - 1;
                d.ky[t] = c.kind == K_CONV3 ? 1 - dy : 1 + dy;
                d.kx[t] = c.kind == K_CONV3 ? 1 - dx : 1 + dx;
            }
        }
    } else {  // K_UPCONV: 4 sub-pixel phases, 1+2+2+4 taps, stored in phase order (py,px) = (0,0),(0,1),(1,0),(1,1)
        f.ntaps = 9;
        f.kn_major = 1;
        int t = 0;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px)
                for (int dy = 0; dy <= py; ++dy)
                    for (int dx = 0; dx <= px; ++dx) {
                        f.ky[t] = py + 1 - 2 * dy;
                        f.kx[t] = px + 1 - 2 * dx;
                        ++t;
                    }
        if (need_dgrad) {   // stride-2 3x3 conv over dOut
            d.ntaps = 9;
            d.kn_major = 0;
            for (int k = 0; k < 9; ++k) { d.ky[k] = k / 3; d.kx[k] = k % 3; }
        }
    }
}

static int add_conv(stcd_engine& e, const std::string& name, int kind, int cin, int cout, bool need_dgrad, bool bias = true) {
    ConvW c;
    c.name = name; c.kind = kind; c.cin = cin; c.cout = cout;
    c.kin_p = round8(cin); c.nout_p = round8(cout);
    if (kind == K_CONV3 || kind == K_CONV3_S2) add_param(e, name + ".weight", {cout, cin, 3, 3}, &c.w_off);
    else if (kind == K_CONV1 || kind == K_CONV1_S2) add_param(e, name + ".weight", {cout, cin, 1, 1}, &c.w_off);
    else if (kind == K_STEM7) add_param(e, name + ".weight", {cout, cin, 7, 7}, &c.w_off);
    else if (kind == K_CONVT2) add_param(e, name + ".weight", {cin, cout, 2, 2}, &c.w_off);
    else add_param(e, name + ".weight", {cin, cout, 3, 3}, &c.w_off);
    c.b_off = -1;
    if (bias) add_param(e, name + ".bias", {cout}, &c.b_off);
    make_specs(c, need_dgrad);
    e.convs.push_back(c);
    return (int)e.convs.size() - 1;
}
static int add_bn(stcd_engine& e, const std::string& name, int C, int calls) {
    BnP b;
    b.name = name; b.C = C; b.calls = calls;
    add_param(e, name + ".weight", {C}, &b.g_off);
    add_param(e, name + ".bias", {C}, &b.b_off);
    b.run_off = e.bn_floats;
    e.bn_floats += 2 * C;
    e.bns.push_back(b);
    return (int)e.bns.size() - 1;
}

struct DecSpec { const char* up; int C; int n; const char* sfx[3]; int cout[3]; };
static const DecSpec DEC[4] = {
    {"upconv4", 128, 3, {"43d", "42d", "41d"}, {128, 128, 64}},
    {"upconv3", 64, 3, {"33d", "32d", "31d"}, {64, 64, 32}},
    {"upconv2", 32, 2, {"22d", "21d", nullptr}, {32, 16, 0}},
    {"upconv1", 16, 2, {"12d", "11d", nullptr}, {16, -1, 0}},   // 11d -> label_nbr, no BN
};

// FC-EF (`Unet`, models/Unet.py:93-154): the FC-Siam-conc plan with ONE encoder stream over cat(x1, x2) -- every encoder tensor holds
// B images in one BatchNorm group, the skips are the stream's own activations, written straight into the concat buffers
static inline int fc_dates(const stcd_engine& e) { return e.arch == STCD_ARCH_FCEF ? 1 : 2; }
static inline bool fc_concat_skips(const stcd_engine& e) { return e.arch == STCD_ARCH_CONC || e.arch == STCD_ARCH_FCEF; }
static inline bool fc_cross(const stcd_engine& e) { return e.arch == STCD_ARCH_XCONC; }
// skip layers of diff / sub whose activations are never stored: the forward writes only the pooled map and the fused skip,
// the backward (k_skip_bwd_pair) recomputes them from Y
static inline bool skip_recomputed(const stcd_engine& e, const Cbrd& L) {
    return e.use_skip_recompute && e.use_act_fuse && e.use_skip_fused && L.pool && L.fuse_dst.off >= 0 && L.groups == 2 &&
           skip_pair_supported(L.npg, L.H, L.W, e.convs[L.conv].cout);
}
static inline bool fc_family(const stcd_engine& e) {      // the engines backward_fcsiam runs
    return e.arch == STCD_ARCH_DIFF || e.arch == STCD_ARCH_CONC || e.arch == STCD_ARCH_SUB || e.arch == STCD_ARCH_FCEF || e.arch == STCD_ARCH_XCONC;
}

static void build_fcsiam_tables(stcd_engine& e) {
    // registration order of SiamUnet_*.__init__ (SiamUnet_diff.py:18-90): conv, bn per layer; upconv before its stage
    for (int s = 0; s < 4; ++s)
        for (int j = 0; j < ENC_STAGE_CONVS[s]; ++j) {
            std::string sfx = std::to_string(s + 1) + std::to_string(j + 1);
            int cin = j == 0 ? (s == 0 ? (e.arch == STCD_ARCH_FCEF ? 2 * e.in_ch : e.in_ch) : ENC_C[s - 1]) : ENC_C[s];
            add_conv(e, "conv" + sfx, K_CONV3, cin, ENC_C[s], !(s == 0 && j == 0));
            add_bn(e, "bn" + sfx, ENC_C[s], fc_dates(e));
        }
    e.enc_param_end = e.param_floats;
    for (int k = 0; k < 4; ++k) {
        const DecSpec& d = DEC[k];
        add_conv(e, d.up, K_UPCONV, d.C, d.C, true);
        int skipc = e.arch == STCD_ARCH_CONC ? 2 * d.C : d.C;
        int cin = d.C + skipc;
        for (int j = 0; j < d.n; ++j) {
            int cout = d.cout[j] < 0 ? e.label : d.cout[j];
            add_conv(e, std::string("conv") + d.sfx[j], K_CONVT3_S1, cin, cout, true);
            if (d.cout[j] >= 0) add_bn(e, std::string("bn") + d.sfx[j], cout, 1);
            cin = cout;
        }
    }
    if (fc_cross(e)) {      // SiamUnet_crossconc.py:119-122: cross_conc1..4 registered after the decoder, each diff.0 / diff.1 / conv_res.0 / conv_res.1
        e.xc.assign(4, XConc());
        for (int l = 0; l < 4; ++l) {
            XConc& X = e.xc[l];
            const std::string n = "cross_conc" + std::to_string(l + 1);
            X.level = l; X.C = ENC_C[l];
            add_param(e, n + ".diff.0.weight", {X.C, 2, 3, 3}, &X.w_off);
            add_param(e, n + ".diff.0.bias", {X.C}, &X.b_off);
            X.bn1 = add_bn(e, n + ".diff.1", X.C, 1);
            X.res.conv = add_conv(e, n + ".conv_res.0", K_CONV3, X.C, X.C, true);
            X.res.bn = add_bn(e, n + ".conv_res.1", X.C, 1);
        }
    }
}

// ------------------------------------------------------------------------------------------ workspace plan
struct Bump {
    int64_t cur = 0;
    int64_t take(int64_t bytes) {
        int64_t o = cur;
        cur += (bytes + 255) & ~(int64_t)255;
        return o;
    }
};

static int conv_index(const stcd_engine& e, const std::string& name) {
    for (size_t i = 0; i < e.convs.size(); ++i)
        if (e.convs[i].name == name) return (int)i;
    return -1;
}
static int bn_index(const stcd_engine& e, const std::string& name) {
    for (size_t i = 0; i < e.bns.size(); ++i)
        if (e.bns[i].name == name) return (int)i;
    return -1;
}

static stcd_conv_geom geom3(int N, int H, int W, int K, int ldi, int co, int ldo) {
    stcd_conv_geom g;
    memset(&g, 0, sizeof(g));
    g.n = N; g.hi = H; g.wi = W; g.ci = K; g.ldi = ldi;
    g.hm = H; g.wm = W; g.in_stride = 1;
    g.ho = H; g.wo = W; g.out_stride = 1; g.oy0 = 0; g.ox0 = 0;
    g.co = co; g.ldo = ldo;
    g.ntaps = 9;
    for (int t = 0; t < 9; ++t) { g.dy[t] = (int8_t)(t / 3 - 1); g.dx[t] = (int8_t)(t % 3 - 1); }
    return g;
}

// 4 sub-pixel phases of ConvTranspose2d(k3,s2,p1,op1): out(2m+py, 2n+px) = sum_{dy<=py, dx<=px} in(m+dy, n+dx) W[py+1-2dy][px+1-2dx]
static stcd_conv_geom geom_up_phase(const UpConv& U, int py, int px, int ldi, int ldo, int* tap0) {
    stcd_conv_geom g;
    memset(&g, 0, sizeof(g));
    g.n = U.N; g.hi = U.h; g.wi = U.w; g.ci = U.C; g.ldi = ldi;
    g.hm = U.h; g.wm = U.w; g.in_stride = 1;
    g.ho = U.Ho; g.wo = U.Wo; g.out_stride = 2; g.oy0 = py; g.ox0 = px;
    g.co = U.C; g.ldo = ldo;
    int t = 0;
    for (int dy = 0; dy <= py; ++dy)
        for (int dx = 0; dx <= px; ++dx) { g.dy[t] = (int8_t)dy; g.dx[t] = (int8_t)dx; ++t; }
    g.ntaps = t;
    static const int start[4] = {0, 1, 3, 5};
    *tap0 = start[py * 2 + px];
    return g;
}

static void conv_work(const stcd_engine& e, const stcd_conv_geom& g, int kreal, int nreal, double* flops, double* bytes);

// the tap-list GEMM kernel takes what the resident-filter kernel cannot, and (STCD_GEMM_OVER_RES: 0 never, 1 [default] when the
// resident kernel would run 16-channel output slices, i.e. re-read X once per 16 output channels, 2 always)
static void pick_gemm_or_res(const stcd_engine& e, ConvOp& op, const stcd_conv_geom& g, int groups) {
    op.gemm = ConvGemmPlan();
    if (!e.use_gemm || op.small) return;
    static const int env_mode = [] { const char* v = getenv("STCD_GEMM_OVER_RES"); return v ? atoi(v) : -1; }();
    // (ChangeFormer's 256-channel decoder convolutions get 16-channel output slices from the resident-filter kernel, i.e. X is
    //  re-read 16 times through L2: the GEMM kernel's 128 x 128 tile measured 8.3 vs 9.5 ms per step there; a tie elsewhere)
    const int mode = env_mode >= 0 ? env_mode : (e.cf ? 1 : 0);
    if (op.res.ok && !(mode == 2 || (mode == 1 && op.res.NT == 1))) return;
    op.gemm = conv_gemm_plan(g, op.plan, groups);
    if (op.gemm.ok) op.res = ConvResPlan();
    // opt-in (measured 0.93 - 1.03 of k_conv_gemm on ChangeFormer's 256 -> 256 layers): layers that take the GEMM kernel,
    // carry no fused BatchNorm statistics (groups == 1 callers that pass none) and fill the chip with 16 x 16 tiles
    static const bool halo_on = [] { const char* v = getenv("STCD_HALO_KERNEL"); return v && v[0] == '1'; }();
    op.halo = ConvHaloPlan();
    // the LDS-DMA kernel takes the layers it fits once they fill the chip with 256-position tiles (ChangeFormer's decoder head)
    op.dma = ConvDmaPlan();
    if (op.gemm.ok && (int64_t)g.n * g.hm * g.wm >= 256 * 192) op.dma = conv_dma_plan(g, op.plan);
    if (halo_on && op.gemm.ok && (int64_t)g.n * ((g.hm + 15) / 16) * ((g.wm + 15) / 16) >= 512) op.halo = conv_halo_plan(g, op.plan);
}

// weight-gradient plan of one launch: the GEMM kernel for one-tap launches with >= 64 channels on both sides, else the tile kernel
static WgradMfmaPlan pick_wgrad_plan(const stcd_engine& e, const stcd_conv_geom& g, int kpad, int wld) {
    // (a GEMM group is one more launch of the stage: only layers with >= 0.8 GFLOP go there -- FC-Siam's two one-tap up-conv
    //  phases, 0.54 GFLOP each, stay in the tile kernel's grid: measured +2 % on the diff step otherwise)
    if (e.use_gemm && e.use_wgroup && 2.0 * g.n * g.hm * g.wm * (double)g.ci * g.co >= 0.8e9) {
        WgradMfmaPlan p = wgrad_gemm_plan(g, kpad, wld);
        if (p.ok) return p;
    }
    // (64 x 32 tile: SNUNet everywhere; ChangeFormer for its 3x3 layers only -- measured 5.0 vs 5.4 ms there, but 2.7 vs 2.1 ms on the
    //  4-tap phases of its transposed convs)
    // ChangeFormer's 256 -> 256 3x3 layers on 256^2 / 512^2 maps: the LDS-DMA kernel (measured: DESIGN.md section 4)
    if (e.use_wgroup && (g.ntaps == 9 || (e.cf && g.ntaps == 4)) && (int64_t)g.n * g.hm * g.wm >= 65536) {
        WgradMfmaPlan p = wgrad_dma_plan(g, kpad, wld);
        if (p.ok) return p;
    }
    return wgrad_mfma_plan(g, kpad, wld, e.arch == STCD_ARCH_SNUNET || (e.cf && g.ntaps == 9));
}

static void build_pack_jobs(stcd_engine& e, Bump& ws) {
    // ---- per-launch slab regions + the batched reduce job tables (MFMA path only)
    for (int st = 0; st < 2; ++st) { e.rjobs[st].clear(); e.rjobs_total[st] = 0; }
    for (int st = 0; st < 2; ++st) e.wgroups[st].clear();
    if (e.dt == BF16 && e.use_mfma) {
        // ---- grouped weight-gradient launches: one grid per (stage, kernel variant).  The K split (gx) of every layer is
        //      chosen for the GROUP: about two resident rounds of blocks over all its layers, each block owning at least
        //      8 tiles, so deep layers get few slabs and the prologue / slab epilogue is amortised.
        if (e.use_wgroup) {
            for (WgradOp* op : e.wgrad_ops) {
                if (!op->plan.ok) continue;
                const bool t9 = !op->plan.dma && op->g.ntaps == 9;      // (one LDS-DMA group per stage, whatever the tap count)
                std::vector<WgradGroup>& gs = e.wgroups[op->stage];
                size_t gi = 0;
                for (; gi < gs.size(); ++gi)
                    if (gs[gi].gemm == op->plan.gemm && gs[gi].dma == op->plan.dma && gs[gi].WCI == op->plan.WCI && gs[gi].NTW == op->plan.NTW && gs[gi].t9 == t9 && gs[gi].xf == (op->xf_C > 0) && gs[gi].tail == op->tail) break;
                if (gi == gs.size()) { WgradGroup g; g.WCI = op->plan.WCI; g.NTW = op->plan.NTW; g.t9 = t9; g.gemm = op->plan.gemm; g.dma = op->plan.dma; g.xf = op->xf_C > 0; g.tail = op->tail; gs.push_back(g); }
                op->grouped = true;
            }
            for (int st = 0; st < 2; ++st)
                for (WgradGroup& G : e.wgroups[st]) {
                    if (G.gemm) { G.lds_bytes = 2 * 2 * G.gemm * (64 * 64 + 8 * 32); continue; }     // gx fixed by wgrad_gemm_plan
                    if (G.dma) {
                        // one block per CU and ONE round: the smallest split length (in 64-position K-tiles) whose blocks -- (split, tap,
                        // channel tile) over all layers of the group -- fit the 256 CUs; every block then walks about the same K
                        G.lds_bytes = 128 * 1024;
                        std::vector<WgradOp*> dops;
                        for (WgradOp* op : e.wgrad_ops)
                            if (op->grouped && op->stage == st && op->plan.dma) dops.push_back(op);
                        auto blocks_at = [&](int64_t kt) {
                            int64_t b = 0;
                            for (WgradOp* op : dops) {
                                const int64_t M = (int64_t)op->g.n * op->g.hi * op->g.wi;
                                b += ((M + kt * 64 - 1) / (kt * 64)) * op->plan.gy * op->plan.gz;
                            }
                            return b;
                        };
                        int64_t lo = 1, hi = 1;
                        // (STCD_WGRAD_DMA_BLOCKS: fewer than one block per CU leaves whole CUs to the kernels of the backward chain the
                        //  group runs beside -- its blocks fill the register file of the CUs they sit on)
                        static const int dma_env = [] { const char* v = getenv("STCD_WGRAD_DMA_BLOCKS"); return v && atoi(v) > 0 ? atoi(v) : 0; }();
                        static const bool cf_side_env = [] { const char* v = getenv("STCD_CF_SIDE"); return !(v && v[0] == '0'); }();
                        // 192 when the group runs on the side stream (single-call backwards): measured 26.9 -> 26.3 ms; 256 gains nothing there
                        const int dma_blocks = dma_env ? dma_env : (e.wg_side_on && cf_side_env ? 192 : 256);
                        while (blocks_at(hi) > dma_blocks) hi *= 2;
                        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (blocks_at(mid) <= dma_blocks) hi = mid; else lo = mid + 1; }
                        for (WgradOp* op : dops) {
                            const ConvW& cv = e.convs[op->conv];
                            const int64_t M = (int64_t)op->g.n * op->g.hi * op->g.wi;
                            int64_t S = (M + lo * 64 - 1) / (lo * 64);
                            const int64_t slab_bytes = (int64_t)op->g.ntaps * cv.fwd.kpad * cv.fwd.wld * 4;
                            S = std::min<int64_t>(S, std::max<int64_t>(1, ((int64_t)64 << 20) / slab_bytes));
                            wgrad_dma_set_split(op->plan, op->g, (int)S, cv.fwd.kpad, cv.fwd.wld);
                        }
                        continue;
                    }
                    std::vector<WgradOp*> ops;
                    int64_t W = 0;
                    for (WgradOp* op : e.wgrad_ops)
                        if (op->grouped && op->stage == st && !op->plan.gemm && !op->plan.dma && op->plan.WCI == G.WCI && op->plan.NTW == G.NTW && (op->g.ntaps == 9) == G.t9 && (op->xf_C > 0) == G.xf && op->tail == G.tail) {
                            ops.push_back(op);
                            const int64_t ntiles = (int64_t)op->g.n * ((op->g.wm + 15) / 16) * ((op->g.hm + 7) / 8);
                            W += ntiles * op->plan.gy * op->plan.gz;
                        }
                    for (WgradOp* op : ops) {   // LDS need of the group = max over its jobs
                        const ConvW& cv = e.convs[op->conv];
                        WgradJob j = wgrad_make_job(op->g, op->plan, 0, 0, 0, cv.fwd.kpad, cv.fwd.wld);
                        if (op->xf_C > 0) wgrad_job_set_xf(j, op->xf_stat_off, op->xf_mask_off, op->xf_C, op->xf_groups, op->xf_npg);
                        G.lds_bytes = std::max(G.lds_bytes, j.lds_bytes);
                    }
                    int slots = wgrad_variant_slots(G.WCI, G.NTW, G.t9, G.lds_bytes);
                    // stage 0 of the FC-Siam family runs beside stage 1's chain on a side stream: its grids leave room for the chain's blocks
                    // (stage 1's groups keep the full budget: half cost 2 %, a quarter 7 %)
                    if (st == 0 && e.wg_side_on && e.wg_side_div > 1 && fc_family(e)) slots = std::max(64, slots / e.wg_side_div);
                    const int64_t rounds = e.wgroup_rounds;
                    const int64_t tpb = std::max<int64_t>(e.wgroup_min_tiles, (W + rounds * slots - 1) / (rounds * slots));
                    for (WgradOp* op : ops) {
                        const ConvW& cv = e.convs[op->conv];
                        const int64_t ntiles = (int64_t)op->g.n * ((op->g.wm + 15) / 16) * ((op->g.hm + 7) / 8);
                        const int64_t slab_bytes = (int64_t)op->g.ntaps * cv.fwd.kpad * cv.fwd.wld * 4;
                        int64_t gx = std::max<int64_t>(1, (ntiles + tpb - 1) / tpb);
                        gx = std::min<int64_t>(gx, std::max<int64_t>(1, ((int64_t)32 << 20) / slab_bytes));
                        op->plan.gx = (int)gx;
                        op->plan.slab_floats = gx * op->g.ntaps * cv.fwd.kpad * cv.fwd.wld;
                    }
                }
        }
        int64_t cur[2] = {0, 0};
        for (WgradOp* op : e.wgrad_ops) {
            if (!op->plan.ok) continue;
            const ConvW& cv = e.convs[op->conv];
            op->slab = ws.take(op->plan.slab_floats * 4);
            if (op->grouped) {
                const bool t9 = !op->plan.dma && op->g.ntaps == 9;      // (one LDS-DMA group per stage, whatever the tap count)
                for (WgradGroup& G : e.wgroups[op->stage])
                    if (G.gemm == op->plan.gemm && G.dma == op->plan.dma && G.WCI == op->plan.WCI && G.NTW == op->plan.NTW && G.t9 == t9 && G.xf == (op->xf_C > 0) && G.tail == op->tail) {
                        op->group = (int)(&G - e.wgroups[op->stage].data());
                        WgradJob j = op->plan.dma ? wgrad_dma_make_job(op->g, op->plan, op->in_off, op->dout_off, op->slab, cv.fwd.kpad, cv.fwd.wld)
                                   : op->plan.gemm ? wgrad_gemm_make_job(op->g, op->plan, op->in_off, op->dout_off, op->slab, cv.fwd.kpad, cv.fwd.wld)
                                                   : wgrad_make_job(op->g, op->plan, op->in_off, op->dout_off, op->slab, cv.fwd.kpad, cv.fwd.wld);
                        if (op->xf_C > 0 && !op->plan.dma && !op->plan.gemm) wgrad_job_set_xf(j, op->xf_stat_off, op->xf_mask_off, op->xf_C, op->xf_groups, op->xf_npg);
                        j.start = G.total_blocks;
                        G.total_blocks += j.gx * j.gy * j.gz;
                        G.jobs.push_back(j);
                        double fl, by;
                        conv_work(e, op->g, op->kreal, op->nreal, &fl, &by);
                        G.flops += fl; G.bytes += by;
                    }
            }
            ReduceJob j{};
            j.slab_off = op->slab; j.out_off = cv.w_off;
            j.slab_stride = (int64_t)op->g.ntaps * cv.fwd.kpad * cv.fwd.wld;
            j.gx = op->plan.gx; j.ntaps = op->g.ntaps; j.K = cv.cin; j.N = cv.cout; j.kpad = cv.fwd.kpad; j.wld = cv.fwd.wld;
            j.ks = cv.fwd.ks; j.kn_major = cv.fwd.kn_major;
            for (int t = 0; t < op->g.ntaps; ++t) {
                j.ky[t] = op->own_taps ? op->oky[t] : cv.fwd.ky[op->tap0 + t];
                j.kx[t] = op->own_taps ? op->okx[t] : cv.fwd.kx[op->tap0 + t];
            }
            j.tiled = (int64_t)j.ntaps * j.K * j.N >= 512 * 1024;     // measured: the tiled form pays from ~0.5 M weights per launch
            if (!j.tiled) {   // index space: blocks of (256 / parts) outputs x parts threads; <= 32 slabs per part where 64 parts allow
                const int64_t outs = (int64_t)j.ntaps * j.K * j.N;
                int parts = 1;
                while (parts < 64 && parts * 32 < j.gx) parts *= 2;
                j.parts = (int8_t)parts;
                const int64_t OUTS = 256 / parts;
                j.count = ((outs + OUTS - 1) / OUTS) * 256;
            } else {   // index space: one block (256 threads) per tile of 32 co x TK ci x taps, all slabs (>= 0.5 M weights: the 32 / 64 MB
                const int TK = j.ntaps == 1 ? 32 : 16;                 // slab caps leave <= 32 of them): see k_reduce_jobs
                const int64_t ntiles = (int64_t)((j.N + 31) / 32) * ((j.K + TK - 1) / TK);
                j.parts = 1;
                j.count = ntiles * 256;
            }
            j.start = cur[op->stage];
            cur[op->stage] += (j.count + 255) & ~(int64_t)255;     // block-aligned: one job per block
            e.rjobs[op->stage].push_back(j);
        }
        for (int st = 0; st < 2; ++st) {
            e.rjobs_total[st] = cur[st];
            e.rjobs_off[st] = ws.take((int64_t)e.rjobs[st].size() * sizeof(ReduceJob) + 16);
            for (WgradGroup& G : e.wgroups[st]) G.table_off = ws.take((int64_t)G.jobs.size() * sizeof(WgradJob) + 16);
        }
    }
    // ---- one repack launch per forward: job tables (uploaded to the workspace on first use)
    for (int with_dgrad = 0; with_dgrad < 2; ++with_dgrad) {
        std::vector<PackJob>& jobs = e.jobs[with_dgrad];
        jobs.clear();
        int64_t cur = 0;
        // every job starts on a 256-element boundary: a block then belongs to ONE job and finds it with scalar loads
        auto push = [&](PackJob j) { j.start = cur; cur += (j.count + 255) & ~(int64_t)255; jobs.push_back(j); };
        const bool mfma = e.dt == BF16 && e.use_mfma;
        if (!mfma) {
            for (auto& cv : e.convs)
                for (int d = 0; d <= with_dgrad; ++d) {
                    const PackSpec& ps = d ? cv.dgrad : cv.fwd;
                    if (!ps.ntaps) continue;
                    PackJob j{};
                    j.kind = 0; j.ps = ps; j.src_off = cv.w_off; j.dst_off = d ? cv.wpk_dgrad : cv.wpk_fwd;
                    j.count = (int64_t)ps.ntaps * ps.kpad * ps.wld;
                    push(j);
                }
        } else {
            for (const ConvOp* op : e.conv_ops) {
                if (op->dgrad && !with_dgrad) continue;
                const ConvW& cv = e.convs[op->conv];
                const PackSpec& full = op->dgrad ? cv.dgrad : cv.fwd;
                PackJob j{};
                j.ps = full;
                j.ps.ntaps = op->g.ntaps;
                for (int t = 0; t < op->g.ntaps; ++t) { j.ps.ky[t] = full.ky[op->tap0 + t]; j.ps.kx[t] = full.kx[op->tap0 + t]; }
                j.src_off = cv.w_off;
                if (cv.kind == K_STEM7) {          // the stem's forward as a 7-tap GEMM over 8-pixel rows (configure_segcd)
                    j.kind = 2; j.dst_off = op->wf; j.count = op->plan.wf_elems;
                    j.Co = cv.cout; j.NTtot = op->plan.NTtot; j.aux = cv.cin;
                    push(j);
                    continue;
                }
                if (op->plan.ok && op->wf >= 0) {
                    j.kind = 1; j.dst_off = op->wf;
                    j.count = op->plan.modeB ? op->plan.wf_elems : op->plan.wf_elems / op->g.ntaps;      // mode A: one thread per (ci, co), all taps
                    j.Ci = op->g.ci; j.Co = op->g.co; j.CiB = op->plan.CiB; j.nchunks = op->plan.nchunks; j.KS = op->plan.KS;
                    j.NTtot = op->plan.NTtot; j.modeB = op->plan.modeB;
                } else {   // no MFMA plan for this launch: it runs on the reference kernel and needs the fp32 image
                    j.kind = 0;
                    j.dst_off = (op->dgrad ? cv.wpk_dgrad : cv.wpk_fwd) + (int64_t)op->tap0 * full.kpad * full.wld * 4;
                    j.count = (int64_t)j.ps.ntaps * full.kpad * full.wld;
                }
                push(j);
            }
        }
        e.jobs_total[with_dgrad] = cur;
        e.jobs_off[with_dgrad] = ws.take((int64_t)jobs.size() * sizeof(PackJob) + 16);
    }
}

static int configure_fcsiam(stcd_engine& e, int B, int H, int W) {
    const int64_t T = (int64_t)dsize(e.dt);
    const int ND = fc_dates(e);            // encoder streams (dates) stacked in the batch dimension
    e.enc.clear(); e.dec.clear(); e.ups.clear(); e.drops.clear();
    e.drop_floats = 0;
    e.Hs[0] = H; e.Ws[0] = W;
    for (int s = 0; s < 4; ++s) { e.Hs[s + 1] = e.Hs[s] / 2; e.Ws[s + 1] = e.Ws[s] / 2; }
    Bump ws;
    auto add_drop = [&](const std::string& name, int rows, int C) {
        DropP d; d.name = name; d.rows = rows; d.C = C; d.off = e.drop_floats;
        e.drop_floats += (int64_t)rows * C;
        e.drops.push_back(d);
        return (int)e.drops.size() - 1;
    };
    auto plain = [&](int N, int h, int w, int C) { TRef t; t.off = ws.take((int64_t)N * h * w * C * T); t.ld = C; return t; };

    e.X0 = plain(ND * B, H, W, 8);
    // concat buffers first (conc keeps its skips inside them)
    for (int s = 0; s < 4; ++s) {
        int Cd = ENC_C[s] + (e.arch == STCD_ARCH_CONC ? 2 : 1) * ENC_C[s];
        e.D[s] = plain(B, e.Hs[s], e.Ws[s], Cd);
        e.dD[s] = plain(B, e.Hs[s], e.Ws[s], Cd);
        e.P[s] = plain(ND * B, e.Hs[s + 1], e.Ws[s + 1], ENC_C[s]);
        e.dP[s] = plain(ND * B, e.Hs[s + 1], e.Ws[s + 1], ENC_C[s]);
    }
    // ---- zero arena, directly behind dP[3] (whose date-0 half must read as zero: the reference's T1 bottleneck pool is
    //      dead work, SiamUnet_diff.py:119 overwritten at :143; the date-1 half is rewritten by upconv4's data gradient):
    //      BatchNorm accumulators of every layer (forward + backward) and the bias-gradient accumulators
    auto r256 = [](int64_t b) { return (b + 255) & ~(int64_t)255; };
    int64_t arena_bytes = 0;
    for (int s = 0; s < 4; ++s) arena_bytes += 2 * ENC_STAGE_CONVS[s] * r256(bn_acc_bytes(2, ENC_C[s]));
    for (int k = 0; k < 4; ++k) {
        for (int j = 0; j < DEC[k].n; ++j)
            if (DEC[k].cout[j] >= 0) arena_bytes += 2 * r256(bn_acc_bytes(1, DEC[k].cout[j]));
        arena_bytes += r256(bn_acc_bytes(1, DEC[k].C));
    }
    arena_bytes += r256(bn_acc_bytes(1, 8));
    if (fc_cross(e)) for (int s = 0; s < 4; ++s) arena_bytes += 4 * r256(bn_acc_bytes(1, ENC_C[s]));      // two BatchNorms per cross_conc block
    e.zero_begin = e.dP[3].off;
    Bump za;
    za.cur = ws.take(arena_bytes);
    e.zero_end = za.cur + arena_bytes;
    // ---- encoder
    for (int s = 0; s < 4; ++s) {
        const int C = ENC_C[s], h = e.Hs[s], w = e.Ws[s];
        for (int j = 0; j < ENC_STAGE_CONVS[s]; ++j) {
            std::string sfx = std::to_string(s + 1) + std::to_string(j + 1);
            Cbrd L;
            L.conv = conv_index(e, "conv" + sfx);
            L.bn = bn_index(e, "bn" + sfx);
            L.drop = add_drop("do" + sfx, ND * B, C);
            L.N = ND * B; L.H = h; L.W = w; L.groups = ND; L.npg = B;
            const bool first = j == 0, last = j == ENC_STAGE_CONVS[s] - 1;
            if (first && s == 0) { L.in = e.X0; L.K = 8; }
            else if (first) { L.in = e.P[s - 1]; L.K = ENC_C[s - 1]; }
            else { L.in.off = e.enc.back().A.off; L.in.ld = C; L.K = C; }
            L.Y = plain(ND * B, h, w, C);
            if (last && fc_concat_skips(e)) {
                L.A.off = e.D[s].off + C * T; L.A.ld = e.D[s].ld; L.A.goff = C;
                L.dA.off = e.dD[s].off + C * T; L.dA.ld = e.dD[s].ld; L.dA.goff = C;
                L.dY = plain(ND * B, h, w, C);
            } else {
                TRef a = plain(ND * B, h, w, C), da = plain(ND * B, h, w, C);
                L.A.off = a.off; L.A.ld = C; L.A.goff = (int64_t)B * h * w * C;
                L.dA.off = da.off; L.dA.ld = C; L.dA.goff = L.A.goff;
                L.dY = da;   // in place
            }
            if (last) { L.pool = true; L.P = e.P[s]; L.dPool = e.dP[s]; }
            if (last && !fc_concat_skips(e) && !fc_cross(e)) { L.fuse_dst.off = e.D[s].off + C * T; L.fuse_dst.ld = e.D[s].ld; }
            if (first && s == 0) L.has_dIn = false;
            else if (first) { L.has_dIn = true; L.dIn = e.dP[s - 1]; }
            else { L.has_dIn = true; L.dIn.off = e.enc.back().dA.off; L.dIn.ld = C; }
            L.stat = ws.take((int64_t)2 * 4 * C * 4);
            e.enc.push_back(L);
        }
    }
    // ---- decoder
    // T2 half of the last pool (FC-EF: the one stream's)
    TRef prevA = {e.P[3].off + (int64_t)(fc_dates(e) - 1) * B * e.Hs[4] * e.Ws[4] * ENC_C[3] * T, ENC_C[3]};
    TRef prevdA = {e.dP[3].off + (int64_t)(fc_dates(e) - 1) * B * e.Hs[4] * e.Ws[4] * ENC_C[3] * T, ENC_C[3]};
    for (int k = 0; k < 4; ++k) {
        const DecSpec& d = DEC[k];
        const int s = 3 - k, h = e.Hs[s], w = e.Ws[s];
        UpConv U;
        U.conv = conv_index(e, d.up); U.level = s; U.N = B; U.h = e.Hs[s + 1]; U.w = e.Ws[s + 1]; U.Ho = h; U.Wo = w; U.C = d.C;
        U.in = prevA; U.dIn = prevdA;
        U.out = e.D[s]; U.dOut = e.dD[s];
        e.ups.push_back(U);
        TRef in = e.D[s], dIn = e.dD[s];
        int K = e.D[s].ld;
        for (int j = 0; j < d.n; ++j) {
            if (d.cout[j] < 0) {   // conv11d: no BN, writes the logits
                e.final_conv = conv_index(e, "conv11d");
                e.finalIn = in; e.dFinalIn = dIn;
                break;
            }
            Cbrd L;
            L.conv = conv_index(e, std::string("conv") + d.sfx[j]);
            L.bn = bn_index(e, std::string("bn") + d.sfx[j]);
            L.drop = add_drop(std::string("do") + d.sfx[j], B, d.cout[j]);
            const int C = d.cout[j];
            L.N = B; L.H = h; L.W = w; L.groups = 1; L.npg = B;
            L.in = in; L.K = K;
            L.Y = plain(B, h, w, C);
            TRef a = plain(B, h, w, C), da = plain(B, h, w, C);
            L.A.off = a.off; L.A.ld = C; L.A.goff = 0;
            L.dA.off = da.off; L.dA.ld = C; L.dA.goff = 0;
            L.dY = da;
            L.has_dIn = true; L.dIn = dIn;
            L.stat = ws.take((int64_t)4 * C * 4);
            e.dec.push_back(L);
            in = a; dIn = da; K = C;
            prevA = a; prevdA = da;
        }
    }
    if (fc_cross(e)) {      // the cross_conc block of every level: pairwise depthwise conv + BN + ReLU, then a dropout-free Cbrd whose
        for (int s = 0; s < 4; ++s) {       // activation IS the skip slice of the level's concat buffer
            XConc& X = e.xc[s];
            const int C = ENC_C[s], h = e.Hs[s], w = e.Ws[s];
            X.G = plain(B, h, w, C); X.dG = plain(B, h, w, C); X.R = plain(B, h, w, C); X.dR = plain(B, h, w, C);
            X.stat1 = ws.take((int64_t)4 * C * 4);
            X.part = ws.take(pairdw_partial_floats(B, h, w, C) * 4);
            Cbrd& L = X.res;
            L.drop = -1; L.N = B; L.H = h; L.W = w; L.groups = 1; L.npg = B;
            L.in = X.R; L.K = C;
            L.Y = plain(B, h, w, C);
            L.A.off = e.D[s].off + C * T; L.A.ld = e.D[s].ld; L.A.goff = 0;
            L.dA.off = e.dD[s].off + C * T; L.dA.ld = e.dD[s].ld; L.dA.goff = 0;
            L.dY = plain(B, h, w, C);
            L.has_dIn = true; L.dIn = X.dR;
            L.stat = ws.take((int64_t)4 * C * 4);
        }
    }
    e.G = plain(B, H, W, 8);
    for (auto& X : e.xc) {
        X.facc1 = za.take(bn_acc_bytes(1, X.C)); X.bacc1 = za.take(bn_acc_bytes(1, X.C));
        X.res.facc = za.take(bn_acc_bytes(1, X.C)); X.res.bacc = za.take(bn_acc_bytes(1, X.C));
    }
    for (auto& L : e.enc) { L.facc = za.take(bn_acc_bytes(L.groups, e.convs[L.conv].cout)); L.bacc = za.take(bn_acc_bytes(L.groups, e.convs[L.conv].cout)); }
    for (auto& L : e.dec) { L.facc = za.take(bn_acc_bytes(L.groups, e.convs[L.conv].cout)); L.bacc = za.take(bn_acc_bytes(L.groups, e.convs[L.conv].cout)); }
    for (auto& U : e.ups) U.bias_acc = za.take(bn_acc_bytes(1, U.C));
    e.final_bias_acc = za.take(bn_acc_bytes(1, 8));
    if (za.cur > e.zero_end) { set_error("internal: zero arena overflow"); return 1; }
    e.scratch8 = ws.take(256);
    e.masks = ws.take(e.drop_floats * 4);
    for (auto& c : e.convs) {
        c.wpk_fwd = ws.take((int64_t)c.fwd.ntaps * c.fwd.kpad * c.fwd.wld * 4);
        if (c.dgrad.ntaps) c.wpk_dgrad = ws.take((int64_t)c.dgrad.ntaps * c.dgrad.kpad * c.dgrad.wld * 4);
    }
    e.dwe_begin = ws.cur;
    for (auto& c : e.convs) {
        c.dwe_floats = (int64_t)c.fwd.ntaps * c.fwd.kpad * c.fwd.wld;
        c.dwe = ws.take(c.dwe_floats * 8);     // fp64 accumulators of the reference weight-gradient path
    }
    e.dwe_end = ws.cur;

    // ---- bind every conv-shaped launch (geometry, filter, MFMA plan); the layer vectors are final from here on
    e.conv_ops.clear();
    e.wgrad_ops.clear();
    e.slab_floats = 0;
    auto bind_conv = [&](ConvOp& op, const stcd_conv_geom& g, int conv, bool dgrad, int tap0, int kreal, int nreal, int groups = 1) {
        op.g = g; op.conv = conv; op.dgrad = dgrad; op.tap0 = tap0; op.kreal = kreal; op.nreal = nreal;
        op.plan = ConvMfmaPlan(); op.wf = -1; op.res = ConvResPlan(); op.res_groups = groups;
        if (e.dt == BF16) {
            op.plan = conv_mfma_plan(g);
            if (op.plan.ok) op.wf = ws.take(op.plan.wf_elems * 2);
            op.small = conv_small_ok(g, op.plan);
            if (e.use_res && !op.small) op.res = conv_res_plan(g, op.plan, groups);
            pick_gemm_or_res(e, op, g, groups);
        }
        e.conv_ops.push_back(&op);
    };
    auto bind_wgrad = [&](WgradOp& op, const stcd_conv_geom& g, int conv, int tap0, int kreal, int nreal, int64_t in_off,
                          int64_t dout_off) {
        op.g = g; op.conv = conv; op.tap0 = tap0; op.kreal = kreal; op.nreal = nreal;
        op.in_off = in_off; op.dout_off = dout_off; op.grouped = false;
        op.xf_stat_off = op.xf_mask_off = -1; op.xf_C = 0; op.xf_groups = op.xf_npg = 1;
        op.plan = WgradMfmaPlan(); op.slab = -1;
        op.stage = e.convs[conv].w_off < e.enc_param_end ? 1 : 0;     // encoder filters are finalised by stage 1
        if (e.dt == BF16) op.plan = pick_wgrad_plan(e, g, e.convs[conv].fwd.kpad, e.convs[conv].fwd.wld);
        e.wgrad_ops.push_back(&op);
    };
    auto bind_cbrd = [&](Cbrd& L) {
        const ConvW& cv = e.convs[L.conv];
        bind_conv(L.fwd, geom3(L.N, L.H, L.W, L.K, L.in.ld, cv.cout, L.Y.ld), L.conv, false, 0, cv.cin, cv.cout, L.groups);
        bind_wgrad(L.wg, geom3(L.N, L.H, L.W, L.K, L.in.ld, cv.cout, L.dY.ld), L.conv, 0, cv.cin, cv.cout, L.in.off, L.dY.off);
        if (L.has_dIn)      // (tiles partitioned by the layer's BatchNorm groups: its data gradient may carry the previous layer's backward sums)
            bind_conv(L.dgr, geom3(L.N, L.H, L.W, cv.dgrad.kpad, L.dY.ld, cv.cin, L.dIn.ld), L.conv, true, 0, cv.cout, cv.cin,
                      (e.use_bwdsum && e.use_bwdsum_res) ? L.groups : 1);
    };
    for (auto& L : e.enc) bind_cbrd(L);
    for (auto& L : e.dec) bind_cbrd(L);
    for (auto& X : e.xc) bind_cbrd(X.res);
    for (auto& L : e.enc) L.wg.tail = false;
    if (e.wg_tail_split && !e.enc.empty()) e.enc[0].wg.tail = true;
    for (auto& U : e.ups) {
        const ConvW& cv = e.convs[U.conv];
        for (int ph = 0; ph < 4; ++ph) {
            int tap0;
            stcd_conv_geom g = geom_up_phase(U, ph >> 1, ph & 1, U.in.ld, U.out.ld, &tap0);
            bind_conv(U.fwd[ph], g, U.conv, false, tap0, U.C, U.C);
            bind_wgrad(U.wg[ph], g, U.conv, tap0, U.C, U.C, U.in.off, U.dOut.off);
        }
        // data gradient: 3x3 stride-2 conv over dOut.  (hi,wi) are the BUFFER dims (row pitch); taps reach at most
        // row 2h-1 / col 2w-1, so replication-padded rows/cols are never read.
        stcd_conv_geom gd;
        memset(&gd, 0, sizeof(gd));
        gd.n = U.N; gd.hi = U.Ho; gd.wi = U.Wo; gd.ci = cv.dgrad.kpad; gd.ldi = U.dOut.ld;
        gd.hm = U.h; gd.wm = U.w; gd.in_stride = 2;
        gd.ho = U.h; gd.wo = U.w; gd.out_stride = 1;
        gd.co = U.C; gd.ldo = U.dIn.ld;
        gd.ntaps = 9;
        for (int t = 0; t < 9; ++t) { gd.dy[t] = (int8_t)(t / 3 - 1); gd.dx[t] = (int8_t)(t % 3 - 1); }
        bind_conv(U.dgr, gd, U.conv, true, 0, U.C, U.C);
    }
    {
        const ConvW& cv = e.convs[e.final_conv];
        bind_conv(e.final_fwd, geom3(B, H, W, cv.kin_p, e.finalIn.ld, e.label, e.label), e.final_conv, false, 0, cv.cin, e.label);
        bind_wgrad(e.final_wg, geom3(B, H, W, cv.kin_p, e.finalIn.ld, e.label, 8), e.final_conv, 0, cv.cin, e.label, e.finalIn.off, e.G.off);
        bind_conv(e.final_dgr, geom3(B, H, W, cv.dgrad.kpad, 8, cv.cin, e.dFinalIn.ld), e.final_conv, true, 0, e.label, cv.cin);
    }
    e.slab = ws.take(e.slab_floats * 4);

    // ---- BatchNorm-backward sums inside the data gradient that produces dA (k_conv_small<.., BWD>): layer N's data gradient writes
    //      dA of layer P (plain buffer, same map, 16 channels); P then skips its k_bn_reduce launch.  Level-1 layers only (the
    //      small-channel kernel): conv12 -> conv11, conv11d -> conv12d, the final conv -> conv11d.
    for (auto& L : e.enc) { L.dgr_bwd = nullptr; L.bwd_sums_fused = false; }
    for (auto& L : e.dec) { L.dgr_bwd = nullptr; L.bwd_sums_fused = false; }
    e.final_dgr_bwd = nullptr;
    if (e.use_bwdsum && e.dt == BF16 && e.use_mfma && e.use_small && !e.use_virt) {
        // STCD_BWDSUM_RES=1: also the layers behind a k_conv_res data gradient (k_conv_res<.., BWD>, one or two n-tiles).  Measured: eight
        // more launches gone (114 -> 106), bn_bwd_reduce 0.150 -> 0.055 ms, conv 0.970 -> 1.052 ms: the step 2.094 -> 2.114 ms (diff), conc
        // 2.433 -> 2.475 -- the resident-filter kernel pays more for 56 extra registers and the Y loads than the small launches cost.  Opt-in.
        const int res_too = e.use_bwdsum_res;
        auto dest_ok = [&](const Cbrd& P, const ConvOp& dgr, int64_t dIn_off, int dIn_ld, int N) {
            const bool kern = (dgr.small && conv_small_bwdsum_ok(dgr.g)) ||
                              (res_too && !dgr.small && !dgr.gemm.ok && e.use_res && dgr.res.ok && dgr.res_groups == P.groups &&
                               conv_res_bwdsum_ok(dgr.g, dgr.res));
            return !P.pool && P.fuse_dst.off < 0 && !P.virt && kern && dgr.wf >= 0 && dIn_off == P.dA.off && dIn_ld == P.dA.ld &&
                   e.convs[P.conv].cout == dgr.g.co && P.N == N && N % P.groups == 0 &&
                   (P.groups == 1 || P.dA.goff == (int64_t)P.npg * P.H * P.W * P.dA.ld);
        };
        auto scan = [&](std::vector<Cbrd>& v) {
            for (size_t i = 1; i < v.size(); ++i) {
                Cbrd& N = v[i]; Cbrd& P = v[i - 1];
                if (N.has_dIn && N.dgr_sum_acc < 0 && dest_ok(P, N.dgr, N.dIn.off, N.dIn.ld, N.N)) { N.dgr_bwd = &P; P.bwd_sums_fused = true; }
            }
        };
        scan(e.enc); scan(e.dec);
        if (!e.dec.empty() && dest_ok(e.dec.back(), e.final_dgr, e.dFinalIn.off, e.dFinalIn.ld, e.dec.back().N)) {
            e.final_dgr_bwd = &e.dec.back(); e.dec.back().bwd_sums_fused = true;
        }
    }

    // ---- virtual activations: a non-skip layer whose ONLY reader is the next conv of its stage hands that conv its raw output Y;
    //      the conv's forward launch (k_conv_small / k_conv_res, XF variants) and its weight gradient (k_wgrad_group, job.xf_*)
    //      apply BN-affine + ReLU + Dropout2d while staging.  Eligibility mirrors exec_conv / exec_wgrad's kernel choice, which is
    //      fixed at configure time (the switches are read at stcd_create).
    e.final_xsrc = nullptr; e.fc_virt_layers = 0;
    for (auto& L : e.enc) { L.virt = false; L.xsrc = nullptr; }
    for (auto& L : e.dec) { L.virt = false; L.xsrc = nullptr; }
    if (e.use_virt && e.dt == BF16 && e.use_mfma && e.use_wgroup && !fc_cross(e)) {
        auto fwd_ok = [&](const ConvOp& op, bool nchw) {
            if (op.wf < 0 || (op.gemm.ok && !nchw)) return false;
            if (op.small && e.use_small) return true;
            return op.res.ok && !nchw && !op.res.single_halo;
        };
        auto wg_ok = [&](const WgradOp& op) { return op.plan.ok && !op.plan.gemm && !op.plan.dma && op.plan.WCI < 4 && op.plan.NTW < 4; };
        auto link = [&](Cbrd& P, ConvOp& cf, WgradOp& cw, bool nchw) {
            if (P.pool || P.fuse_dst.off >= 0) return false;
            if (!fwd_ok(cf, nchw) || !wg_ok(cw)) return false;
            P.virt = true; ++e.fc_virt_layers;
            cw.in_off = P.Y.off;
            cw.xf_stat_off = P.stat;
            cw.xf_mask_off = (P.drop >= 0) ? e.masks + e.drops[P.drop].off * 4 : -1;
            cw.xf_C = e.convs[P.conv].cout; cw.xf_groups = P.groups; cw.xf_npg = P.npg;
            return true;
        };
        for (size_t i = 0; i + 1 < e.enc.size(); ++i) {
            Cbrd& P = e.enc[i]; Cbrd& Cn = e.enc[i + 1];
            if (Cn.in.off != P.A.off) continue;                       // (the next layer reads the pooled map: P is a skip layer)
            if (link(P, Cn.fwd, Cn.wg, false)) { Cn.xsrc = &P; Cn.in.off = P.Y.off; Cn.in.ld = P.Y.ld; }
        }
        for (size_t i = 0; i < e.dec.size(); ++i) {
            Cbrd& P = e.dec[i];
            if (i + 1 < e.dec.size() && e.dec[i + 1].in.off == P.A.off) {
                Cbrd& Cn = e.dec[i + 1];
                if (link(P, Cn.fwd, Cn.wg, false)) { Cn.xsrc = &P; Cn.in.off = P.Y.off; Cn.in.ld = P.Y.ld; }
            } else if (e.finalIn.off == P.A.off) {
                if (link(P, e.final_fwd, e.final_wg, true)) { e.final_xsrc = &P; e.finalIn.off = P.Y.off; e.finalIn.ld = P.Y.ld; }
            }
        }
    }

    // ---- bias gradients without a pass of their own: the transposed conv of stage k wrote channels [0, C) of the concat
    //      buffer, so its bias gradient is the per-channel sum of that slice of d(concat) -- which the data-gradient launch
    //      of the stage's first conv produces.  When that launch runs on the resident-filter kernel it sums the slice in its
    //      epilogue (integer accumulators, as the BatchNorm statistics); one k_bias_finish launch per backward converts.
    e.bias_jobs.clear();
    {
        size_t di = 0;
        for (int k = 0; k < 4; ++k) {
            UpConv& U = e.ups[k];
            Cbrd& L = e.dec[di];
            int nb = 0;
            for (int j = 0; j < DEC[k].n; ++j) nb += DEC[k].cout[j] >= 0;
            di += nb;
            U.bias_fused = false;
            const bool even = 2 * U.h == U.Ho && 2 * U.w == U.Wo;
            if (e.dt == BF16 && e.use_mfma && nb > 0 && L.has_dIn && L.dIn.off == U.dOut.off && L.dgr.res.ok && L.dgr.wf >= 0 && even) {
                U.bias_fused = true;
                L.dgr_sum_acc = U.bias_acc; L.dgr_sum_c0 = 0; L.dgr_sum_C = U.C;
            }
            if (e.dt == BF16) {     // fused or not, the bf16 path sums the bias gradient in the integer accumulator (deterministic)
                BiasJob jb{}; jb.acc_off = U.bias_acc; jb.out_off = e.convs[U.conv].b_off; jb.C = U.C; jb.valid = U.C; jb.scale = BN_BS;
                e.bias_jobs.push_back(jb);
            }
        }
        BiasJob jb{}; jb.acc_off = e.final_bias_acc; jb.out_off = e.convs[e.final_conv].b_off; jb.C = 8; jb.valid = e.label; jb.scale = BN_BS;
        e.bias_jobs.push_back(jb);
        e.bias_jobs_off = ws.take((int64_t)e.bias_jobs.size() * sizeof(BiasJob) + 16);
    }

    // test introspection (stcd_ws_tensor_*): input, conv output, activation (per date: the two halves may sit in different
    // buffers) and output gradient of every conv + BN layer, as they stand after a forward / backward
    e.ws_tensors.clear();
    auto rec_layer = [&](const Cbrd& L) {
        const ConvW& cv = e.convs[L.conv];
        auto rec = [&](const char* sfx, int64_t off, int ld, int n, int ch) {
            if (off < 0) return;
            stcd_ws_tensor r;
            memset(&r, 0, sizeof(r));
            snprintf(r.name, sizeof(r.name), "%s.%s", cv.name.c_str(), sfx);
            r.offset_bytes = off; r.n = n; r.h = L.H; r.w = L.W; r.c = ch; r.ld = ld; r.dtype = e.dt;
            e.ws_tensors.push_back(r);
        };
        // "in.virt": the input is the producer's RAW conv output (virtual activation); the producer then records no A
        rec(L.xsrc ? "in.virt" : "in", L.in.off, L.in.ld, L.N, L.K);
        rec("Y", L.Y.off, L.Y.ld, L.N, cv.cout);
        for (int g = 0; g < L.groups && !L.virt && !skip_recomputed(e, L); ++g) {
            char sfx[8];
            snprintf(sfx, sizeof(sfx), "A.g%d", g);
            rec(sfx, L.A.off + g * L.A.goff * T, L.A.ld, L.npg, cv.cout);
        }
        rec("dY", L.dY.off, L.dY.ld, L.N, cv.cout);
    };
    for (const Cbrd& L : e.enc) rec_layer(L);
    for (const Cbrd& L : e.dec) rec_layer(L);

    build_pack_jobs(e, ws);
    e.jobs_uploaded_ws = nullptr;
    e.ws_bytes = ws.cur;
    return 0;
}

// ------------------------------------------------------------------------------------------ execution helpers
struct Ctx {
    stcd_engine& e;
    char* ws;
    const float* params;
    float* grads;
    hipStream_t s;
    template <typename P = void> P* at(int64_t off) const { return (P*)(ws + off); }
};

struct ProfScope {
    Prof& p; hipStream_t s; bool on; hipEvent_t b;
    ProfScope(const Ctx& c, int klass, double flops, double bytes, const char* kernel = nullptr);
    ~ProfScope() { if (on) (void)hipEventRecord(b, s); }
};

ProfScope::ProfScope(const Ctx& c, int klass, double flops, double bytes, const char* kernel)
    : p(c.e.prof), s(c.s), on(c.e.prof.on), b(nullptr) {
    if (!on) return;
    static const char* DEF[PC_COUNT] = {"k_conv", "k_wgrad", "k_bn_reduce<T, 0>", "k_bn_act", "k_bn_reduce<T, 1>", "k_bn_bwd_apply",
                                        "k_pool_bwd|k_fuse|k_slice", "k_pack_jobs|k_reduce_dw"};
    ProfRec r;
    r.a = p.get(); r.b = p.get(); r.klass = klass; r.flops = flops; r.bytes = bytes;
    r.name_id = p.intern(kernel ? kernel : DEF[klass]);
    b = r.b;
    p.recs.push_back(r);
    (void)hipEventRecord(r.a, s);
}

// algorithmic work of one conv-like launch: flops = 2 MACs over real channels,
// bytes = (input + output + weights) * sizeof(T), each tensor counted once.
static void conv_work(const stcd_engine& e, const stcd_conv_geom& g, int kreal, int nreal, double* flops, double* bytes) {
    double pos = (double)g.n * g.hm * g.wm;
    *flops = 2.0 * pos * g.ntaps * kreal * nreal;
    double in_px = g.in_stride == 1 ? pos : (double)g.n * (2.0 * g.hm) * (2.0 * g.wm);
    *bytes = (in_px * kreal + pos * nreal + (double)g.ntaps * kreal * nreal) * (double)dsize(e.dt);
}

static bool mfma_on(const stcd_engine& e) { return e.dt == BF16 && e.use_mfma; }

// A request for per-channel sums of the launch's output, fused into the conv's epilogue: BatchNorm statistics of a forward
// conv (all channels, sum and sum of squares) or the sum of a channel slice of a data gradient (a bias gradient).
struct StatReq { long long* acc = nullptr; int groups = 1; int c0 = 0; int C = 0; float s1 = BN_FS1, s2 = BN_FS2; };
// *fused receives 1 when the kernel delivered the sums, 0 when the caller must run the separate pass.
static void exec_conv(const Ctx& c, const ConvOp& op, const void* in, const float* bias, void* out, bool nchw,
                      const StatReq* sr = nullptr, int* stat_chunks = nullptr, const ConvEpi* epi = nullptr, int* epi_fused = nullptr,
                      const XfSrc* xf = nullptr, const BwdSum* bs = nullptr) {
    const int stat_groups = (sr && sr->acc) ? sr->groups : 0;
    long long* stat_acc = sr ? sr->acc : nullptr;
    const bool bn_form = sr && sr->c0 == 0 && sr->C == op.g.co && sr->s1 == BN_FS1 && sr->s2 == BN_FS2;
    const ConvW& cv = c.e.convs[op.conv];
    const PackSpec& ps = op.dgrad ? cv.dgrad : cv.fwd;
    double fl, by;
    conv_work(c.e, op.g, op.kreal, op.nreal, &fl, &by);
    char kname[64];
    const bool gemm_path = mfma_on(c.e) && op.gemm.ok && op.wf >= 0 && !nchw;
    const bool small_path = !gemm_path && mfma_on(c.e) && op.small && op.wf >= 0 && c.e.use_small;
    const bool res_path = !gemm_path && !small_path && mfma_on(c.e) && op.res.ok && op.wf >= 0 && !nchw;
    const bool mfma_path = !small_path && !res_path && mfma_on(c.e) && op.plan.ok && op.wf >= 0;
    const bool dma_path = gemm_path && op.dma.ok && op.res_groups == 1 && !(stat_groups > 0 && stat_acc);
    if (dma_path) snprintf(kname, sizeof(kname), "k_conv_dma");
    else if (gemm_path && op.halo.ok && !(stat_groups > 0 && stat_acc)) snprintf(kname, sizeof(kname), "k_conv_halo");
    else if (gemm_path) snprintf(kname, sizeof(kname), "k_conv_gemm<%d>", op.gemm.W);
    else if (res_path) snprintf(kname, sizeof(kname), "k_conv_res<%d, %d, %s>", op.res.NT, op.res.CW, op.res.single_halo ? "true" : "false");
    else if (small_path) snprintf(kname, sizeof(kname), "k_conv_small<1, %d>", (op.g.ntaps * op.g.ci + 31) / 32 <= 5 ? 5 : 9);
    else if (mfma_path) snprintf(kname, sizeof(kname), "k_conv_mfma<%d>", op.plan.NT);
    else snprintf(kname, sizeof(kname), "k_conv_ref");
    if (bs) {      // planned at configure time for exactly this kernel (dest_ok): anything else is a plan error
        ProfScope ps2(c, PC_CONV, fl, by + (double)op.g.n * op.g.ho * op.g.wo * op.g.co * 2.0, small_path ? "k_conv_small<bwd>" : "k_conv_res<bwd>");
        const int rc = small_path ? launch_conv_small(op.g, in, c.at(op.wf), bias, out, nchw, bs->groups, nullptr, op.g.co, c.s, nullptr, bs)
                                  : launch_conv_res(op.g, op.plan, op.res, in, c.at(op.wf), bias, out, op.res_groups, nullptr, op.g.co, c.s, 0,
                                                    BN_BS, BN_BS, nullptr, bs);
        if (rc != 0) set_error("fused BatchNorm-backward sums: the data-gradient launch does not fit its BWD kernel");
        return;
    }
    ProfScope prof(c, PC_CONV, fl, by, kname);
    if (stat_chunks) *stat_chunks = 0;
    if (epi_fused) *epi_fused = 0;
    if (dma_path && launch_conv_dma(op.g, op.plan, op.dma, in, c.at(op.wf), bias, out, c.s, epi) == 0) {
        if (epi_fused && epi) *epi_fused = 1;
        return;
    }
    if (gemm_path && op.halo.ok && !(stat_groups > 0 && stat_acc)) {
        if (launch_conv_halo(op.g, op.plan, op.halo, in, c.at(op.wf), bias, out, c.s, epi) == 0) {
            if (epi_fused && epi) *epi_fused = 1;
            return;
        }
    }
    if (gemm_path) {
        const bool want = stat_groups > 0 && stat_groups == op.res_groups && stat_acc;
        long long* sp = want ? stat_acc : nullptr;
        if (launch_conv_gemm(op.g, op.plan, op.gemm, in, c.at(op.wf), bias, out, op.res_groups, sp, want ? sr->C : op.g.co, c.s,
                           want ? sr->c0 : 0, want ? sr->s1 : BN_FS1, want ? sr->s2 : BN_FS2, epi) == 0) {
            if (stat_chunks && want) *stat_chunks = 1;
            if (epi_fused && epi) *epi_fused = 1;      // (the other kernels have no fused epilogue: the caller runs it element-wise)
            return;
        }
    }
    const bool xf_on = xf && xf->on;
    if (mfma_on(c.e) && op.small && op.wf >= 0 && c.e.use_small) {
        // (a virtual input partitions the tiles by the PRODUCER's BatchNorm groups, statistics or not: eval mode has none)
        const int groups = xf_on ? xf->groups : (stat_groups > 0 && bn_form) ? stat_groups : 1;
        long long* sp = (stat_groups > 0 && bn_form && stat_groups == groups) ? stat_acc : nullptr;
        if (launch_conv_small(op.g, in, c.at(op.wf), bias, out, nchw, groups, sp, op.g.co, c.s, xf) == 0) {
            if (stat_chunks && sp) *stat_chunks = 1;
            return;
        }
    }
    if (res_path) {
        const bool want = stat_groups > 0 && stat_groups == op.res_groups && stat_acc;
        long long* sp = want ? stat_acc : nullptr;
        if (launch_conv_res(op.g, op.plan, op.res, in, c.at(op.wf), bias, out, op.res_groups, sp, want ? sr->C : op.g.co, c.s,
                            want ? sr->c0 : 0, want ? sr->s1 : BN_FS1, want ? sr->s2 : BN_FS2, xf) == 0) {
            if (stat_chunks && want) *stat_chunks = 1;
            return;
        }
    }
    if (xf_on) {     // planned at configure time for a kernel that applies the transform: nothing below can (and must not run on raw Y)
        set_error("internal: a virtual-activation input reached a convolution kernel without the staging transform");
        return;
    }
    if (mfma_on(c.e) && op.plan.ok && op.wf >= 0 &&
        launch_conv_mfma(op.g, op.plan, in, c.at(op.wf), bias, out, nchw, c.s) == 0)
        return;
    const float* w = c.at<float>(op.dgrad ? cv.wpk_dgrad : cv.wpk_fwd) + (int64_t)op.tap0 * ps.kpad * ps.wld;
    launch_conv_ref(c.e.dt, op.g, in, w, ps.kpad, ps.wld, bias, out, nchw, c.s);
}

// the four sub-pixel phases of a stride-2 transposed convolution as one launch; false: not applicable (caller loops)
static bool exec_conv_x4(const Ctx& c, const ConvOp ops[4], const void* in, const float* bias, void* out) {
    if (!mfma_on(c.e)) return false;
    stcd_conv_geom g[4]; ConvMfmaPlan p[4]; const void* wf[4];
    double fl = 0.0, by = 0.0;
    for (int k = 0; k < 4; ++k) {
        if (!ops[k].plan.ok || ops[k].wf < 0) return false;
        g[k] = ops[k].g; p[k] = ops[k].plan; wf[k] = c.at(ops[k].wf);
        double f1, b1;
        conv_work(c.e, ops[k].g, ops[k].kreal, ops[k].nreal, &f1, &b1);
        fl += f1; by += b1;
    }
    char kname[64];
    snprintf(kname, sizeof(kname), "k_conv_mfma_x4<%d>", p[0].NT);
    ProfScope prof(c, PC_CONV, fl, by, kname);
    return launch_conv_mfma_x4(g, p, in, wf, bias, out, c.s) == 0;
}

// weight gradient of one launch, delivered straight into the reference-layout gradient tensor
static void wgroup_member_ready(const Ctx& c, const WgradOp& op);
static void exec_wgrad(const Ctx& c, const WgradOp& op, const void* in, const void* dout) {
    const ConvW& cv = c.e.convs[op.conv];
    PackSpec sub = cv.fwd;                 // this launch's taps of the filter
    sub.ntaps = op.g.ntaps;
    for (int t = 0; t < op.g.ntaps; ++t) {
        sub.ky[t] = op.own_taps ? op.oky[t] : cv.fwd.ky[op.tap0 + t];
        sub.kx[t] = op.own_taps ? op.okx[t] : cv.fwd.kx[op.tap0 + t];
    }
    double fl, by;
    conv_work(c.e, op.g, op.kreal, op.nreal, &fl, &by);
    if (mfma_on(c.e) && op.plan.ok && op.grouped) {          // runs in the stage's grouped launch (reduce_stage, or early: wgroup_member_ready)
        if (c.e.wg_early) wgroup_member_ready(c, op);
        return;
    }
    if (mfma_on(c.e) && op.plan.ok) {
        int rc;
        {
            char kname[64];
            snprintf(kname, sizeof(kname), "k_wgrad_mfma<%d, %d>", op.plan.WCI, op.plan.NTW);
            ProfScope prof(c, PC_WGRAD, fl, by, kname);
            rc = launch_wgrad_mfma(op.g, op.plan, in, dout, c.at<float>(op.slab), sub.kpad, sub.wld, c.s);
        }
        if (rc == 0) return;       // the slabs are summed by the stage's batched reduce launch (reduce_stage)
    }
    if (op.own_taps) { set_error("internal: no reference fallback for a launch with its own tap table"); return; }
    double* dwe = c.at<double>(cv.dwe) + (int64_t)op.tap0 * sub.kpad * sub.wld;
    if (mfma_on(c.e))   // the bulk memset of the reference path's accumulators is skipped in MFMA mode
        (void)hipMemsetAsync(dwe, 0, (size_t)sub.ntaps * sub.kpad * sub.wld * 8, c.s);
    {
        ProfScope prof(c, PC_WGRAD, fl, by, "k_wgrad_ref");
        launch_wgrad_ref_f64(c.e.dt, op.g, in, dout, dwe, sub.kpad, sub.wld, c.s);
    }
    launch_unpack_dw(sub, dwe, c.grads + cv.w_off, c.s);
}

static int pack_all_weights(const Ctx& c, bool with_dgrad) {
    stcd_engine& e = c.e;
    if (e.jobs_uploaded_ws != (const void*)c.ws) {      // first use of this workspace: upload both job tables
        for (int k = 0; k < 2; ++k) {
            if (!e.jobs[k].empty())
                STCD_HIP(hipMemcpyAsync(c.at(e.jobs_off[k]), e.jobs[k].data(), e.jobs[k].size() * sizeof(PackJob),
                                        hipMemcpyHostToDevice, c.s));
            if (!e.rjobs[k].empty())
                STCD_HIP(hipMemcpyAsync(c.at(e.rjobs_off[k]), e.rjobs[k].data(), e.rjobs[k].size() * sizeof(ReduceJob),
                                        hipMemcpyHostToDevice, c.s));
            for (WgradGroup& G : e.wgroups[k])
                STCD_HIP(hipMemcpyAsync(c.at(G.table_off), G.jobs.data(), G.jobs.size() * sizeof(WgradJob), hipMemcpyHostToDevice, c.s));
        }
        if (!e.bias_jobs.empty())
            STCD_HIP(hipMemcpyAsync(c.at(e.bias_jobs_off), e.bias_jobs.data(), e.bias_jobs.size() * sizeof(BiasJob), hipMemcpyHostToDevice, c.s));
        e.jobs_uploaded_ws = c.ws;
    }
    const int k = with_dgrad ? 1 : 0;
    // the caller declared the weights constant under this tag (stcd_set_weights_tag): the images packed into THIS workspace from
    // THIS parameter buffer under the same tag are still valid (the forward-only image set is a subset of the training one)
    // Training forwards always repack and never vouch for the next call (an optimizer step follows them).
    if (k == 0 && e.weights_tag != 0 && e.weights_tag == e.packed_tag && e.packed_ws == (const void*)c.ws && e.packed_params == (const void*)c.params)
        return 0;
    ProfScope prof(c, PC_PACK, 0.0, 0.0);
    launch_pack_jobs(c.at<PackJob>(e.jobs_off[k]), (int)e.jobs[k].size(), e.jobs_total[k], c.params, c.ws, c.s);
    e.packed_tag = k == 0 ? e.weights_tag : 0; e.packed_ws = c.ws; e.packed_params = c.params; e.packed_level = k;
    return 0;
}

static void launch_wgroup(const Ctx& c, WgradGroup& G) {
    {
        char kname[64];
        if (G.dma) snprintf(kname, sizeof(kname), "k_wgrad_dma");
        else if (G.gemm) snprintf(kname, sizeof(kname), "k_wgrad_gemm<%d>", G.gemm);
        else snprintf(kname, sizeof(kname), "k_wgrad_group<%d, %d, %s>", G.WCI, G.NTW, G.t9 ? "true" : "false");
        ProfScope prof(c, PC_WGRAD, G.flops, G.bytes, kname);
        G.launched = true;
        if (G.dma) {
            launch_wgrad_dma_group(c.at<WgradJob>(G.table_off), (int)G.jobs.size(), G.total_blocks, c.ws, c.s);
            return;
        }
        if (G.gemm) {
            launch_wgrad_gemm_group(G.gemm, c.at<WgradJob>(G.table_off), (int)G.jobs.size(), G.total_blocks, c.ws, c.s);
            return;
        }
        if (launch_wgrad_group(G.WCI, G.NTW, G.t9, c.at<WgradJob>(G.table_off), (int)G.jobs.size(), G.total_blocks, G.lds_bytes,
                               c.ws, c.s, G.xf) != 0)
            set_error("grouped weight-gradient launch exceeds the LDS budget");
    }
}

// FC-Siam backward with early launches (e.wg_early): a member's operands (X stored by the forward, dY final) exist when the chain
// calls exec_wgrad for it; the group's grid goes out with its LAST member -- on the side stream, behind an event of the chain's
// stream, when there is one -- instead of at the end of the stage
static void wgroup_member_ready(const Ctx& c, const WgradOp& op) {
    stcd_engine& e = c.e;
    if (op.group < 0 || op.group >= (int)e.wgroups[op.stage].size()) return;
    WgradGroup& G = e.wgroups[op.stage][op.group];
    if (++G.seen < (int)G.jobs.size() || G.launched) return;
    if (e.wg_cur_side) {
        if (hipEventRecord(e.wg_fork, c.s) != hipSuccess || hipStreamWaitEvent(e.wg_cur_side, e.wg_fork, 0) != hipSuccess) { set_error("side-stream fork failed"); return; }
        Ctx cs{e, c.ws, c.params, c.grads, e.wg_cur_side};
        launch_wgroup(cs, G);
    } else {
        launch_wgroup(c, G);
    }
}

// the stage's groups that have not gone out yet, then the batched slab reduction; resets the groups' run-time state
static void reduce_stage(const Ctx& c, int stage) {
    stcd_engine& e = c.e;
    for (WgradGroup& G : e.wgroups[stage]) {
        if (!G.launched) launch_wgroup(c, G);
        G.launched = false; G.seen = 0;
    }
    if (e.rjobs[stage].empty()) return;
    ProfScope prof(c, PC_PACK, 0.0, 0.0, "k_reduce_jobs");
    launch_reduce_jobs(c.at<ReduceJob>(e.rjobs_off[stage]), (int)e.rjobs[stage].size(), e.rjobs_total[stage], c.ws, c.grads, c.s);
}

// the producer P as its consumer's staging transform sees it (XfSrc, common.h): training -> batch statistics from P's accumulators,
// the consumer's block 0 publishes P.stat and updates the running statistics; eval -> the running statistics, nothing written
static XfSrc xf_source(const Ctx& c, const Cbrd& P, float* bn_running, bool training) {
    stcd_engine& e = c.e;
    const BnP& bn = e.bns[P.bn];
    XfSrc x;
    x.C = e.convs[P.conv].cout; x.groups = P.groups; x.npg = P.npg; x.ppg = (long long)P.npg * P.H * P.W;
    x.facc = training ? c.at<long long>(P.facc) : nullptr;
    x.gamma = c.params + bn.g_off; x.beta = c.params + bn.b_off;
    x.rmean = bn_running + bn.run_off; x.rvar = bn_running + bn.run_off + x.C;
    x.stat = c.at<float>(P.stat);
    x.mask = (training && e.drop_p > 0.f && P.drop >= 0) ? c.at<float>(e.masks) + e.drops[P.drop].off : nullptr;
    x.publish = training ? 1 : 0;
    x.on = 1;
    if (training && (e.xf_mode & 1)) {       // the table was published by k_bn_finalize right behind the producer (cbrd_forward)
        x.facc = nullptr; x.gamma = nullptr; x.beta = nullptr; x.rmean = nullptr; x.rvar = nullptr; x.publish = 0;
    }
    if (e.xf_mode & 2) x.on = 2;             // TIMING EXPERIMENT ONLY (wrong results): stage the raw tensor, no transform
    return x;
}

static const float* a_gamma(const Ctx& c, const Cbrd& L) { return c.params + c.e.bns[L.bn].g_off; }

static void cbrd_forward(const Ctx& c, const Cbrd& L, float* bn_running, bool training) {
    stcd_engine& e = c.e;
    const ConvW& cv = e.convs[L.conv];
    const BnP& bn = e.bns[L.bn];
    const int C = cv.cout;
    int fused_chunks = 0;
    StatReq sr; sr.acc = training ? c.at<long long>(L.facc) : nullptr; sr.groups = L.groups; sr.C = C;
    XfSrc xf;
    if (L.xsrc) xf = xf_source(c, *L.xsrc, bn_running, training);
    exec_conv(c, L.fwd, c.at(L.in.off), c.params + cv.b_off, c.at(L.Y.off), false, &sr, &fused_chunks, nullptr, nullptr, L.xsrc ? &xf : nullptr);
    const int64_t ppg = (int64_t)L.npg * L.H * L.W;
    const double act_bytes = (double)L.N * L.H * L.W * C * (double)dsize(e.dt);
    float* stat = c.at<float>(L.stat);
    if (training) {
        if (!fused_chunks) {
            ProfScope ps(c, PC_BN_STATS, 0.0, act_bytes);
            launch_bn_stats(e.dt, c.at(L.Y.off), L.Y.ld, C, L.groups, ppg, c.at<long long>(L.facc), c.s);
        }
    }
    if (L.virt) {            // the consumer applies scale / shift / ReLU / Dropout2d while staging Y (and publishes the statistics)
        if (training && (e.xf_mode & 1)) {
            XfSrc x = xf_source(c, L, bn_running, training);
            x.facc = c.at<long long>(L.facc); x.gamma = a_gamma(c, L); x.beta = x.gamma ? c.params + e.bns[L.bn].b_off : nullptr;
            x.rmean = bn_running + e.bns[L.bn].run_off; x.rvar = x.rmean + C;
            ProfScope ps(c, PC_BN_STATS, 0.0, 0.0, "k_bn_finalize");
            launch_bn_finalize(x, c.s);
        }
        return;
    }
    BnActArgs a;
    a.Y = c.at(L.Y.off); a.ldy = L.Y.ld;
    a.A = skip_recomputed(e, L) ? nullptr : c.at(L.A.off); a.lda = L.A.ld; a.a_group_off = L.A.goff;
    a.P = L.pool ? c.at(L.P.off) : nullptr; a.ldp = L.P.ld;
    a.stat = stat;
    a.mask = (training && e.drop_p > 0.f && L.drop >= 0) ? c.at<float>(e.masks) + e.drops[L.drop].off : nullptr;
    a.C = C; a.groups = L.groups; a.npg = L.npg; a.H = L.H; a.W = L.W; a.relu = 1;
    // the activation kernel derives scale / shift itself: from the accumulators in training mode, from the running statistics in
    // eval mode (no finalize / prepare launch either way)
    a.facc = training ? c.at<long long>(L.facc) : nullptr;
    a.gamma = c.params + bn.g_off; a.beta = c.params + bn.b_off;
    a.running_mean = bn_running + bn.run_off; a.running_var = bn_running + bn.run_off + C;
    if (L.fuse_dst.off >= 0 && e.use_act_fuse && L.groups == 2) {
        ProfScope ps(c, PC_BN_ACT, 0.0, act_bytes * (a.A ? 2.75 : 1.75), "k_bn_act_pair");
        launch_bn_act_pair(e.dt, a, c.at(L.fuse_dst.off), L.fuse_dst.ld, e.arch == STCD_ARCH_DIFF ? 0 : 1, c.s);
        return;
    }
    ProfScope ps(c, PC_BN_ACT, 0.0, act_bytes * (L.pool ? 2.25 : 2.0));
    launch_bn_act(e.dt, a, c.s);
}

// the BatchNorm-backward sums of layer P as a data-gradient launch forms them (BwdSum, common.h)
static BwdSum bwd_sum_of(const Ctx& c, const Cbrd& P) {
    stcd_engine& e = c.e;
    BwdSum b;
    b.Y = c.at(P.Y.off); b.ldy = P.Y.ld; b.stat = c.at<float>(P.stat);
    b.mask = (e.drop_p > 0.f && P.drop >= 0) ? c.at<float>(e.masks) + e.drops[P.drop].off : nullptr;
    b.acc = c.at<long long>(P.bacc); b.groups = P.groups;
    return b;
}

// skip_chunks > 0: dA and the BN partial sums were already produced by launch_skip_bwd (that many rows per date)
static void cbrd_backward(const Ctx& c, const Cbrd& L, int skip_chunks = 0) {
    stcd_engine& e = c.e;
    const ConvW& cv = e.convs[L.conv];
    const BnP& bn = e.bns[L.bn];
    const int C = cv.cout;
    const int64_t HW = (int64_t)L.H * L.W, ppg = (int64_t)L.npg * HW;
    const float* stat = c.at<float>(L.stat);
    const float* mask = (e.drop_p > 0.f && L.drop >= 0) ? c.at<float>(e.masks) + e.drops[L.drop].off : nullptr;
    long long* bacc = c.at<long long>(L.bacc);
    const double act_bytes = (double)L.N * HW * C * (double)dsize(e.dt);
    (void)ppg;
    if (!skip_chunks && !L.bwd_sums_fused) {
        ProfScope ps(c, PC_BN_BWD_REDUCE, 0.0, 2.0 * act_bytes);
        launch_bn_bwd_reduce(e.dt, c.at(L.dA.off), L.dA.ld, L.dA.goff, c.at(L.Y.off), L.Y.ld, stat, mask, C, L.groups, L.npg, HW, 1,
                             bacc, c.s);
    }
    {
        ProfScope ps(c, PC_BN_BWD_APPLY, 0.0, 3.0 * act_bytes);
        launch_bn_bwd_apply(e.dt, c.at(L.dA.off), L.dA.ld, L.dA.goff, c.at(L.dY.off), L.dY.ld, c.at(L.Y.off), L.Y.ld, stat, bacc,
                            c.grads + bn.g_off, c.grads + bn.b_off, mask, C, L.groups, L.npg, HW, 1, c.s);
    }
    // weight gradient (the conv bias feeds only a train-mode BN: its gradient is exactly zero and stays zero)
    exec_wgrad(c, L.wg, c.at(L.in.off), c.at(L.dY.off));
    if (L.has_dIn) {
        StatReq sr;
        if (L.dgr_sum_acc >= 0) { sr.acc = c.at<long long>(L.dgr_sum_acc); sr.groups = 1; sr.c0 = L.dgr_sum_c0; sr.C = L.dgr_sum_C; sr.s1 = sr.s2 = BN_BS; }
        BwdSum bs;
        if (L.dgr_bwd) bs = bwd_sum_of(c, *L.dgr_bwd);
        exec_conv(c, L.dgr, c.at(L.dY.off), nullptr, c.at(L.dIn.off), false, L.dgr_sum_acc >= 0 ? &sr : nullptr, nullptr, nullptr, nullptr, nullptr,
                  L.dgr_bwd ? &bs : nullptr);
    }
}

static void upconv_forward(const Ctx& c, const UpConv& U) {
    stcd_engine& e = c.e;
    const ConvW& cv = e.convs[U.conv];
    if (!exec_conv_x4(c, U.fwd, c.at(U.in.off), c.params + cv.b_off, c.at(U.out.off)))
        for (int ph = 0; ph < 4; ++ph) exec_conv(c, U.fwd[ph], c.at(U.in.off), c.params + cv.b_off, c.at(U.out.off), false);
    launch_rep_pad(e.dt, c.at(U.out.off), U.out.ld, U.N, U.Ho, U.Wo, 2 * U.h, 2 * U.w, U.C, c.s);
}

static void upconv_backward(const Ctx& c, const UpConv& U) {
    stcd_engine& e = c.e;
    const ConvW& cv = e.convs[U.conv];
    launch_rep_pad_bwd(e.dt, c.at(U.dOut.off), U.dOut.ld, U.N, U.Ho, U.Wo, 2 * U.h, 2 * U.w, U.C, c.s);
    // bias gradient over the un-padded 2h x 2w region == all phases' positions
    if (U.bias_fused && mfma_on(e)) {
        // summed by the data-gradient launch that produced dOut; converted by k_bias_finish at the end of the stage
    } else if (2 * U.h == U.Ho && 2 * U.w == U.Wo) {
        launch_bias_grad(e.dt, c.at(U.dOut.off), U.dOut.ld, (int64_t)U.N * U.Ho * U.Wo, U.C, c.grads + cv.b_off, c.s,
                         e.dt == BF16 ? c.at<long long>(U.bias_acc) : nullptr);
    } else {
        const int64_t T = (int64_t)dsize(e.dt);
        for (int n = 0; n < U.N; ++n)
            for (int y = 0; y < 2 * U.h; ++y)
                launch_bias_grad(e.dt, c.at<char>(U.dOut.off) + ((int64_t)(n * U.Ho + y) * U.Wo) * U.dOut.ld * T, U.dOut.ld,
                                 2 * U.w, U.C, c.grads + cv.b_off, c.s, e.dt == BF16 ? c.at<long long>(U.bias_acc) : nullptr);
    }
    for (int ph = 0; ph < 4; ++ph) exec_wgrad(c, U.wg[ph], c.at(U.in.off), c.at(U.dOut.off));
    exec_conv(c, U.dgr, c.at(U.dOut.off), nullptr, c.at(U.dIn.off), false);
}

// cross_conc block (SiamUnet_crossconc.py:24-33): skip slice of the level's concat buffer = relu(bn(conv(relu(bn(pairdw(x1, x2))))))
static void xconc_forward(const Ctx& c, const XConc& X, const Cbrd& skip, float* bn_running, bool training) {
    stcd_engine& e = c.e;
    const BnP& bn = e.bns[X.bn1];
    const int C = X.C, B = X.res.N, h = X.res.H, w = X.res.W;
    {
        ProfScope ps(c, PC_POOL_FUSE, 0.0, 3.0 * B * h * w * C * (double)dsize(e.dt), "k_pairdw_fwd");
        launch_pairdw_fwd(e.dt, c.at(skip.A.off), skip.A.ld, skip.A.goff, c.at(X.G.off), X.G.ld, c.params + X.w_off, c.params + X.b_off, B, h, w, C, c.s);
    }
    if (training) launch_bn_stats(e.dt, c.at(X.G.off), X.G.ld, C, 1, (int64_t)B * h * w, c.at<long long>(X.facc1), c.s);
    BnActArgs a;
    a.Y = c.at(X.G.off); a.ldy = X.G.ld;
    a.A = c.at(X.R.off); a.lda = X.R.ld; a.a_group_off = 0;
    a.P = nullptr; a.ldp = 0;
    a.stat = c.at<float>(X.stat1);
    a.mask = nullptr;
    a.C = C; a.groups = 1; a.npg = B; a.H = h; a.W = w; a.relu = 1;
    a.facc = training ? c.at<long long>(X.facc1) : nullptr;
    a.gamma = c.params + bn.g_off; a.beta = c.params + bn.b_off;
    a.running_mean = bn_running + bn.run_off; a.running_var = bn_running + bn.run_off + C;
    launch_bn_act(e.dt, a, c.s);
    cbrd_forward(c, X.res, bn_running, training);
}
// its backward: writes the gradient of BOTH dates' skip activations (the pool gradient is accumulated onto them in the encoder stage)
static void xconc_backward(const Ctx& c, const XConc& X, const Cbrd& skip) {
    stcd_engine& e = c.e;
    const BnP& bn = e.bns[X.bn1];
    const int C = X.C, B = X.res.N, h = X.res.H, w = X.res.W;
    const int64_t HW = (int64_t)h * w;
    cbrd_backward(c, X.res);                       // BN-backward of conv_res, its weight gradient, data gradient -> dR
    launch_bn_bwd_reduce(e.dt, c.at(X.dR.off), X.dR.ld, 0, c.at(X.G.off), X.G.ld, c.at<float>(X.stat1), nullptr, C, 1, B, HW, 1,
                         c.at<long long>(X.bacc1), c.s);
    launch_bn_bwd_apply(e.dt, c.at(X.dR.off), X.dR.ld, 0, c.at(X.dG.off), X.dG.ld, c.at(X.G.off), X.G.ld, c.at<float>(X.stat1),
                        c.at<long long>(X.bacc1), c.grads + bn.g_off, c.grads + bn.b_off, nullptr, C, 1, B, HW, 1, c.s);
    ProfScope ps(c, PC_POOL_FUSE, 0.0, 6.0 * B * h * w * C * (double)dsize(e.dt), "k_pairdw_bwd");
    launch_pairdw_bwd_data(e.dt, c.at(X.dG.off), X.dG.ld, c.at(skip.dA.off), skip.dA.ld, skip.dA.goff, c.params + X.w_off, B, h, w, C, c.s);
    launch_pairdw_bwd_filter(e.dt, c.at(skip.A.off), skip.A.ld, skip.A.goff, c.at(X.dG.off), X.dG.ld, c.grads + X.w_off, c.at<float>(X.part),
                             B, h, w, C, c.s);
    // (diff.0's bias sits in front of a train-mode BatchNorm: its gradient is exactly zero and stays zero)
}

static int forward_fcsiam(stcd_engine& e, const float* x1, const float* x2, const float* params, float* bn_running,
                          const float* masks, uint64_t seed, int training, float* logits, void* workspace, hipStream_t s) {
    Ctx c{e, (char*)workspace, params, nullptr, s};
    const int B = e.B, dt = e.dt;
    const int64_t T = (int64_t)dsize(dt);
    static const int SKIP_IDX[4] = {1, 3, 6, 9};
    if (training && e.drop_p > 0.f) {
        if (masks) STCD_HIP(hipMemcpyAsync(c.at(e.masks), masks, e.drop_floats * 4, hipMemcpyDeviceToDevice, s));
        else launch_dropout_gen(c.at<float>(e.masks), e.drop_floats, seed, e.drop_p, s);
    } else if (training && e.fc_virt_layers > 0) {
        // p == 0 with virtual activations: the weight-gradient job table addresses the mask buffer unconditionally (it is built at
        // configure time), so the buffer must read as all ones (k_dropout_gen with p = 0 writes 1 / (1 - 0) everywhere)
        launch_dropout_gen(c.at<float>(e.masks), e.drop_floats, seed, 0.f, s);
    }
    if (pack_all_weights(c, training != 0)) return 1;
    if (training) STCD_HIP(hipMemsetAsync(c.at(e.zero_begin), 0, e.zero_end - e.zero_begin, s));   // the step's ONE workspace memset
    launch_in_pack(dt, x1, x2, c.at(e.X0.off), B, e.in_ch, e.H, e.W, s, e.arch == STCD_ARCH_FCEF ? 0 : 2);      // 0: cat(x1, x2) along the channels
    for (auto& L : e.enc) cbrd_forward(c, L, bn_running, training != 0);
    size_t di = 0;
    for (int k = 0; k < 4; ++k) {
        const UpConv& U = e.ups[k];
        const int s_ = U.level, C = ENC_C[s_];
        upconv_forward(c, U);
        const Cbrd& skip = e.enc[SKIP_IDX[s_]];
        if (fc_cross(e)) xconc_forward(c, e.xc[s_], skip, bn_running, training != 0);
        else if (!fc_concat_skips(e) && !(e.use_act_fuse && skip.fuse_dst.off >= 0)) {
            ProfScope ps(c, PC_POOL_FUSE, 0.0, 3.0 * B * e.Hs[s_] * e.Ws[s_] * C * (double)T);
            launch_fuse(dt, e.arch == STCD_ARCH_DIFF ? 0 : 1, c.at(skip.A.off), skip.A.ld, skip.A.goff,
                        c.at<char>(e.D[s_].off) + C * T, e.D[s_].ld, B, (int64_t)e.Hs[s_] * e.Ws[s_], C, s);
        }
        for (int j = 0; j < DEC[k].n; ++j) {
            if (DEC[k].cout[j] < 0) break;
            cbrd_forward(c, e.dec[di++], bn_running, training != 0);
        }
    }
    {
        XfSrc xf;
        if (e.final_xsrc) xf = xf_source(c, *e.final_xsrc, bn_running, training != 0);
        exec_conv(c, e.final_fwd, c.at(e.finalIn.off), params + e.convs[e.final_conv].b_off, logits, true, nullptr, nullptr, nullptr, nullptr,
                  e.final_xsrc ? &xf : nullptr);
    }
    STCD_HIP(hipGetLastError());
    return 0;
}

// The side stream of the decoder's weight gradients (FC-Siam family, both backward stages in one call): created on first use, lowest
// priority.  nullptr: switched off (STCD_WGRAD_SIDE=0), or the caller's stream is being captured into a graph (a captured step stays on
// one stream).  MEASURED (MI355X, SiamUnet_diff 16 x 256^2): serial 2.281-2.289 ms; side stream with the planner's full grids 2.265
// (the grouped launches fill every CU: the chain's kernels queue behind them); with a quarter of the block budget for stage 0's
// groups (fewer, longer blocks: fewer slabs too) 2.223-2.235 ms.  A CU-masked stream (hipExtStreamCreateWithCUMask, any mask
// including all CUs) cost 0.8-1.7 ms per step on this stack and was dropped.
static hipStream_t wgrad_side_stream(stcd_engine& e, hipStream_t s) {
    if (!e.wg_side_on || !e.use_wgroup || !mfma_on(e)) return nullptr;
    if (e.prof.on) return nullptr;                      // instrumented steps (stcd_profile_enable) time every kernel ALONE: same plan, one stream
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (cap != hipStreamCaptureStatusNone) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (e.wg_side && e.wg_side_dev == dev) return e.wg_side;
    if (e.wg_side) return nullptr;                      // created on another device: stay serial there
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);      // (least, greatest)
    hipStream_t st = nullptr;
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_lo) != hipSuccess) { (void)hipGetLastError(); e.wg_side_on = 0; return nullptr; }
    if (hipEventCreateWithFlags(&e.wg_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e.wg_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError(); (void)hipStreamDestroy(st); e.wg_side_on = 0; return nullptr;
    }
    e.wg_side = st; e.wg_side_dev = dev;
    return st;
}

struct EarlyScope {      // exec_wgrad sends a group out with its last member for the duration of a backward call
    stcd_engine& e;
    EarlyScope(stcd_engine& e_, bool on, hipStream_t sd) : e(e_) { e.wg_early = on; e.wg_cur_side = sd; }
    ~EarlyScope() { e.wg_early = false; e.wg_cur_side = nullptr; }
};

static int backward_fcsiam(stcd_engine& e, const float* grad_logits, const float* params, float* grads, void* workspace,
                           int stage, hipStream_t s) {
    Ctx c{e, (char*)workspace, params, grads, s};
    // both stages in one call (no gradient all-reduce between them): stage 0's weight gradients only read stored tensors and write
    // their own slabs / gradient entries, so they can run beside stage 1's chain
    hipStream_t side = stage < 0 ? wgrad_side_stream(e, s) : nullptr;
    static const bool early_on = [] { const char* v = getenv("STCD_WGRAD_EARLY"); return !(v && v[0] == '0'); }();
    EarlyScope early_scope(e, early_on && mfma_on(e) && e.use_wgroup, side);
    for (int st = 0; st < 2; ++st)
        if (stage < 0 || stage == st)
            for (WgradGroup& G : e.wgroups[st]) { G.seen = 0; G.launched = false; }      // (a failed call may have left them set)
    const int B = e.B, dt = e.dt;
    const int64_t T = (int64_t)dsize(dt);
    static const int SKIP_IDX[4] = {1, 3, 6, 9};
    if (stage <= 0) {
        STCD_HIP(hipMemsetAsync(grads, 0, e.param_floats * 4, s));
        if (!mfma_on(e)) STCD_HIP(hipMemsetAsync(c.at(e.dwe_begin), 0, e.dwe_end - e.dwe_begin, s));
        // (the date-0 half of dP[3] and every accumulator were zeroed by the forward's arena memset)
        // conv11d: its bias gradient = per-channel sum of d(logits), formed while the gradient is packed
        launch_gout_pack(dt, grad_logits, c.at(e.G.off), B, e.label, e.H, e.W, s, c.at<long long>(e.final_bias_acc));
        exec_wgrad(c, e.final_wg, c.at(e.finalIn.off), c.at(e.G.off));
        {
            BwdSum bs;
            if (e.final_dgr_bwd) bs = bwd_sum_of(c, *e.final_dgr_bwd);
            exec_conv(c, e.final_dgr, c.at(e.G.off), nullptr, c.at(e.dFinalIn.off), false, nullptr, nullptr, nullptr, nullptr, nullptr,
                      e.final_dgr_bwd ? &bs : nullptr);
        }
        int di = (int)e.dec.size() - 1;
        for (int k = 3; k >= 0; --k) {
            int nb = 0;
            for (int j = 0; j < DEC[k].n; ++j) nb += DEC[k].cout[j] >= 0;
            for (int j = 0; j < nb; ++j) cbrd_backward(c, e.dec[di--]);
            const UpConv& U = e.ups[k];
            const int s_ = U.level, C = ENC_C[s_];
            upconv_backward(c, U);
            const Cbrd& skip = e.enc[SKIP_IDX[s_]];
            if (fc_cross(e)) xconc_backward(c, e.xc[s_], skip);
            else if (!fc_concat_skips(e) && !e.use_skip_fused) {
                ProfScope ps(c, PC_POOL_FUSE, 0.0, (e.arch == STCD_ARCH_DIFF ? 5.0 : 3.0) * B * e.Hs[s_] * e.Ws[s_] * C * (double)T);
                launch_fuse_bwd(dt, e.arch == STCD_ARCH_DIFF ? 0 : 1, c.at(skip.A.off), skip.A.ld, skip.A.goff,
                                c.at<char>(e.dD[s_].off) + C * T, e.dD[s_].ld, c.at(skip.dA.off), skip.dA.ld, skip.dA.goff, B,
                                (int64_t)e.Hs[s_] * e.Ws[s_], C, s);
            }
        }
        if (side) {
            STCD_HIP(hipEventRecord(e.wg_fork, s));
            STCD_HIP(hipStreamWaitEvent(side, e.wg_fork, 0));
            Ctx cs{e, (char*)workspace, params, grads, side};
            reduce_stage(cs, 0);
            launch_bias_finish(c.at<BiasJob>(e.bias_jobs_off), (int)e.bias_jobs.size(), c.ws, c.grads, side);
            STCD_HIP(hipEventRecord(e.wg_join, side));
        } else {
            reduce_stage(c, 0);
            launch_bias_finish(c.at<BiasJob>(e.bias_jobs_off), (int)e.bias_jobs.size(), c.ws, c.grads, s);
        }
    }
    if (stage < 0 || stage == 1) {
        for (int li = (int)e.enc.size() - 1; li >= 0; --li) {
            const Cbrd& L = e.enc[li];
            int skip_chunks = 0;
            if (L.pool && !fc_concat_skips(e) && !fc_cross(e) && e.use_skip_fused) {
                // pool gradient + skip-fusion gradient + BN partial sums of the level's last conv in one pass
                const int C = e.convs[L.conv].cout;
                int lvl = 0;
                while (lvl < 4 && SKIP_IDX[lvl] != li) ++lvl;
                const float* mk = e.drop_p > 0.f ? c.at<float>(e.masks) + e.drops[L.drop].off : nullptr;
                if (skip_recomputed(e, L)) {
                    ProfScope ps(c, PC_POOL_FUSE, 0.0, 2.75 * L.N * L.H * L.W * C * (double)T, "k_skip_bwd_pair");
                    launch_skip_bwd_pair(dt, e.arch == STCD_ARCH_DIFF ? 0 : 1, c.at(L.Y.off), L.Y.ld, c.at<char>(e.dD[lvl].off) + C * T,
                                         e.dD[lvl].ld, c.at(L.dPool.off), L.dPool.ld, c.at(L.dA.off), L.dA.ld, L.dA.goff,
                                         c.at<float>(L.stat), mk, L.npg, L.H, L.W, C, c.at<long long>(L.bacc), s);
                } else {
                ProfScope ps(c, PC_POOL_FUSE, 0.0, 4.75 * L.N * L.H * L.W * C * (double)T, "k_skip_bwd");
                launch_skip_bwd(dt, e.arch == STCD_ARCH_DIFF ? 0 : 1, c.at(L.A.off), L.A.ld, L.A.goff, c.at(L.Y.off), L.Y.ld,
                                c.at<char>(e.dD[lvl].off) + C * T, e.dD[lvl].ld, c.at(L.dPool.off), L.dPool.ld, c.at(L.dA.off), L.dA.ld,
                                L.dA.goff, c.at<float>(L.stat), e.drop_p > 0.f ? c.at<float>(e.masks) + e.drops[L.drop].off : nullptr,
                                L.npg, L.H, L.W, C, c.at<long long>(L.bacc), s);
                }
                skip_chunks = 1;
            } else if (L.pool) {  // dA_skip += gradient routed back through the 2x2 max-pool
                const int C = e.convs[L.conv].cout;
                ProfScope ps(c, PC_POOL_FUSE, 0.0, 3.25 * L.N * L.H * L.W * C * (double)T);
                launch_pool_bwd(dt, c.at(L.A.off), L.A.ld, L.A.goff, c.at(L.dPool.off), L.dPool.ld, c.at(L.dA.off), L.dA.ld,
                                L.dA.goff, L.groups, L.npg, L.H, L.W, C, 1, s);
            }
            cbrd_backward(c, L, skip_chunks);
        }
        if (side) {
            STCD_HIP(hipEventRecord(e.wg_fork, s));
            STCD_HIP(hipStreamWaitEvent(side, e.wg_fork, 0));
            Ctx cs{e, (char*)workspace, params, grads, side};
            reduce_stage(cs, 1);
            STCD_HIP(hipEventRecord(e.wg_join, side));
            STCD_HIP(hipStreamWaitEvent(s, e.wg_join, 0));
        } else {
            reduce_stage(c, 1);
        }
    }
    STCD_HIP(hipGetLastError());
    return 0;
}

// ================================================================================================ SNUNet-ECAM
static const int SN_F[5] = {32, 64, 128, 256, 512};

static int add_block_params(stcd_engine& e, const std::string& name, int cin, int c, int calls, bool need_dgrad1, NBlock* b) {
    b->name = name; b->Cin = cin; b->C = c;
    b->c1 = add_conv(e, name + ".conv1", K_CONV3, cin, c, need_dgrad1);
    b->bn1 = add_bn(e, name + ".bn1", c, calls);
    b->c2 = add_conv(e, name + ".conv2", K_CONV3, c, c, true);
    b->bn2 = add_bn(e, name + ".bn2", c, calls);
    return 0;
}

// registration order of SNUNet_ECAM.__init__ (SNUNet.py:73-106)
static void build_snunet_tables(stcd_engine& e) {
    const int* f = SN_F;
    e.sn_blocks.clear(); e.sn_ups.clear();
    auto blk = [&](const std::string& n, int cin, int c, int calls, bool dg) {
        NBlock b; add_block_params(e, n, cin, c, calls, dg, &b); e.sn_blocks.push_back(b);
    };
    auto up = [&](const std::string& n, int c) {
        SnUp u; u.C = c; u.conv = add_conv(e, n + ".up", K_CONVT2, c, c, true); e.sn_ups.push_back(u);
    };
    blk("conv0_0", e.in_ch, f[0], 2, false);
    blk("conv1_0", f[0], f[1], 2, true); up("Up1_0", f[1]);
    blk("conv2_0", f[1], f[2], 2, true); up("Up2_0", f[2]);
    blk("conv3_0", f[2], f[3], 2, true); up("Up3_0", f[3]);
    blk("conv4_0", f[3], f[4], 1, true); up("Up4_0", f[4]);
    blk("conv0_1", f[0] * 2 + f[1], f[0], 1, true);
    blk("conv1_1", f[1] * 2 + f[2], f[1], 1, true); up("Up1_1", f[1]);
    blk("conv2_1", f[2] * 2 + f[3], f[2], 1, true); up("Up2_1", f[2]);
    blk("conv3_1", f[3] * 2 + f[4], f[3], 1, true); up("Up3_1", f[3]);
    blk("conv0_2", f[0] * 3 + f[1], f[0], 1, true);
    blk("conv1_2", f[1] * 3 + f[2], f[1], 1, true); up("Up1_2", f[1]);
    blk("conv2_2", f[2] * 3 + f[3], f[2], 1, true); up("Up2_2", f[2]);
    blk("conv0_3", f[0] * 4 + f[1], f[0], 1, true);
    blk("conv1_3", f[1] * 4 + f[2], f[1], 1, true); up("Up1_3", f[1]);
    blk("conv0_4", f[0] * 5 + f[1], f[0], 1, true);
    const int c4 = f[0] * 4, c1 = f[0];
    add_param(e, "ca.fc1.weight", {c4 / 16, c4, 1, 1}, &e.sn_w[0]);
    add_param(e, "ca.fc2.weight", {c4, c4 / 16, 1, 1}, &e.sn_w[1]);
    add_param(e, "ca1.fc1.weight", {c1 / 4, c1, 1, 1}, &e.sn_w[2]);
    add_param(e, "ca1.fc2.weight", {c1, c1 / 4, 1, 1}, &e.sn_w[3]);
    e.sn_final = add_conv(e, "conv_final", K_CONV1, c4, e.label, true);
    e.enc_param_end = 0;      // one backward stage: every gradient is final when stage 0 returns
}

static int sn_block_index(const stcd_engine& e, const std::string& name) {
    for (size_t i = 0; i < e.sn_blocks.size(); ++i)
        if (e.sn_blocks[i].name == name) return (int)i;
    return -1;
}
static int sn_up_index(const stcd_engine& e, const std::string& name) {
    for (size_t i = 0; i < e.sn_ups.size(); ++i)
        if (e.convs[e.sn_ups[i].conv].name == name + ".up") return (int)i;
    return -1;
}

static stcd_conv_geom geom1(int N, int H, int W, int K, int ldi, int co, int ldo) {   // 1x1
    stcd_conv_geom g = geom3(N, H, W, K, ldi, co, ldo);
    g.ntaps = 1; g.dy[0] = 0; g.dx[0] = 0;
    return g;
}

static int configure_snunet(stcd_engine& e, int B, int H, int W) {
    const int64_t T = (int64_t)dsize(e.dt);
    const int* f = SN_F;
    e.drops.clear(); e.drop_floats = 0;
    e.conv_ops.clear(); e.wgrad_ops.clear(); e.slab_floats = 0;
    Bump ws;
    int hs[5], wsz[5];
    hs[0] = H; wsz[0] = W;
    for (int l = 0; l < 4; ++l) { hs[l + 1] = hs[l] / 2; wsz[l + 1] = wsz[l] / 2; }
    auto plain = [&](int N, int h, int w, int C) { TRef t; t.off = ws.take((int64_t)N * h * w * C * T); t.ld = C; return t; };
    auto B_ = [&](const char* n) -> NBlock& { return e.sn_blocks[sn_block_index(e, n)]; };

    e.X0 = plain(2 * B, H, W, 8);
    e.G = plain(B, H, W, 8);
    const int c4 = f[0] * 4;
    e.snE = plain(B, H, W, c4); e.snZ = plain(B, H, W, c4); e.sndZ = plain(B, H, W, c4);

    // ---- activation buffers of every block.  dOut buffers are allocated back to back: one memset per backward.
    e.sn_dout_begin = ws.cur;
    e.sndE = plain(B, H, W, c4);
    for (auto& b : e.sn_blocks) {
        const int lvl = b.name[4] - '0', j = b.name[6] - '0';
        const bool enc = j == 0;
        b.H = hs[lvl]; b.W = wsz[lvl];
        b.N = (enc && lvl < 4) ? 2 * B : B;
        b.groups = (enc && lvl < 4) ? 2 : 1;
        b.npg = B;
        if (lvl == 0 && j >= 1) {            // x0_1..x0_4 live directly in the ECAM input (one copy less)
            b.dOut.off = e.sndE.off + (int64_t)(j - 1) * f[0] * T; b.dOut.ld = c4;
        } else {
            b.dOut = plain(b.N, b.H, b.W, b.C);
        }
    }
    e.sn_dout_end = ws.cur;
    for (auto& b : e.sn_blocks) {
        const int lvl = b.name[4] - '0', j = b.name[6] - '0';
        const bool enc = j == 0;
        if (lvl == 0 && j >= 1) { b.Out.off = e.snE.off + (int64_t)(j - 1) * f[0] * T; b.Out.ld = c4; }
        else b.Out = plain(b.N, b.H, b.W, b.C);
        b.Y1 = plain(b.N, b.H, b.W, b.C); b.A1 = plain(b.N, b.H, b.W, b.C); b.Y2 = plain(b.N, b.H, b.W, b.C);
        b.dZ2 = plain(b.N, b.H, b.W, b.C); b.dA1 = plain(b.N, b.H, b.W, b.C);
        b.pool = enc && lvl < 4;
        if (b.pool) { b.P = plain(b.N, b.H / 2, b.W / 2, b.C); b.dP = plain(b.N, b.H / 2, b.W / 2, b.C); }
        b.stat1 = ws.take((int64_t)b.groups * 4 * b.C * 4);
        b.stat2 = ws.take((int64_t)b.groups * 4 * b.C * 4);
    }
    e.zero_begin = ws.cur;
    for (auto& b : e.sn_blocks) {
        b.facc1 = ws.take(bn_acc_bytes(b.groups, b.C)); b.facc2 = ws.take(bn_acc_bytes(b.groups, b.C));
        b.bacc1 = ws.take(bn_acc_bytes(b.groups, b.C)); b.bacc2 = ws.take(bn_acc_bytes(b.groups, b.C));
    }
    for (auto& u : e.sn_ups) u.bias_acc = ws.take(bn_acc_bytes(1, u.C));
    e.final_bias_acc = ws.take(bn_acc_bytes(1, 8));
    e.zero_end = ws.cur;
    // ---- inputs
    for (auto& b : e.sn_blocks) {
        const int lvl = b.name[4] - '0', j = b.name[6] - '0';
        b.srcs.clear(); b.up = -1;
        if (j == 0) {
            if (lvl == 0) { b.in = e.X0; b.Cin = 8; b.dIn.off = -1; }
            else {
                NBlock& prev = e.sn_blocks[sn_block_index(e, std::string("conv") + char('0' + lvl - 1) + "_0")];
                b.in = prev.P; b.dIn = prev.dP; b.Cin = prev.C;
                if (lvl == 4) {             // only the B date goes one level deeper (SNUNet.py:123 is commented out)
                    const int64_t half = (int64_t)B * (prev.H / 2) * (prev.W / 2) * prev.C * T;
                    b.in.off += half; b.dIn.off += half;
                }
            }
            continue;
        }
        b.in = plain(b.N, b.H, b.W, b.Cin); b.dIn = plain(b.N, b.H, b.W, b.Cin);
        NBlock& enc = e.sn_blocks[sn_block_index(e, std::string("conv") + char('0' + lvl) + "_0")];
        const int64_t half = (int64_t)B * enc.H * enc.W * enc.C * T;
        int coff = 0;
        auto add_src = [&](TRef src, TRef dsrc, int C, int prod, int grp) {
            SrcSlice s_; s_.src = src; s_.dsrc = dsrc; s_.C = C; s_.prod = prod; s_.grp = grp; b.srcs.push_back(s_); coff += C;
        };
        const int enc_i = sn_block_index(e, std::string("conv") + char('0' + lvl) + "_0");
        add_src(enc.Out, enc.dOut, enc.C, enc_i, 0);                                                      // x{lvl}_0A
        add_src(TRef{enc.Out.off + half, enc.Out.ld}, TRef{enc.dOut.off + half, enc.dOut.ld}, enc.C, enc_i, 1);   // x{lvl}_0B
        for (int k = 1; k < j; ++k) {
            const int di = sn_block_index(e, std::string("conv") + char('0' + lvl) + "_" + char('0' + k));
            NBlock& d = e.sn_blocks[di];
            add_src(d.Out, d.dOut, d.C, di, -1);
        }
        // up-sampled tail: Up{lvl+1}_{j-1}( x{lvl+1}_{j-1} ), the B date for j == 1
        const std::string upname = std::string("Up") + char('0' + lvl + 1) + "_" + char('0' + j - 1);
        b.up = sn_up_index(e, upname);
        SnUp& u = e.sn_ups[b.up];
        NBlock& lower = e.sn_blocks[sn_block_index(e, std::string("conv") + char('0' + lvl + 1) + "_" + char('0' + j - 1))];
        u.N = B; u.h = lower.H; u.w = lower.W;
        u.src = lower.Out; u.dsrc = lower.dOut;
        if (j == 1 && lvl + 1 < 4) {        // encoder output of both dates: take the B half
            const int64_t lh = (int64_t)B * lower.H * lower.W * lower.C * T;
            u.src.off += lh; u.dsrc.off += lh;
        }
        u.out = TRef{b.in.off + (int64_t)coff * T, b.Cin};
        u.dOut = TRef{b.dIn.off + (int64_t)coff * T, b.Cin};
        u.coff = coff;
        u.tmp = plain(B, u.h, u.w, u.C);
    }
    // ---- dense concatenation without copy kernels: every producer writes its output straight into its consumers' concat
    //      slices (extra destinations of its last activation kernel), and gathers its gradient from their concat-gradient
    //      slices (extra sources of its BatchNorm-backward reduction).  The two dates of an encoder level sit side by side
    //      in a consumer (A at coff, B at coff + C): one view with a per-date channel offset.
    for (auto& b : e.sn_blocks) { b.extra_dst.clear(); b.grad_src.clear(); }
    for (auto& b : e.sn_blocks) {
        int coff = 0;
        for (auto& s_ : b.srcs) {
            if (s_.grp != 1) {       // (the date-1 slice of an encoder level rides on its date-0 view)
                NBlock& P = e.sn_blocks[s_.prod];
                ViewRef v; v.ld = b.in.ld; v.off = b.in.off + (int64_t)coff * T;
                if (s_.grp == 0) { v.gmask = 3; v.goff = s_.C; } else { v.gmask = 1; v.goff = 0; }
                P.extra_dst.push_back(v);
                ViewRef gv = v; gv.off = b.dIn.off + (int64_t)coff * T; gv.ld = b.dIn.ld;
                P.grad_src.push_back(gv);
            }
            coff += s_.C;
        }
    }
    for (auto& b : e.sn_blocks) {
        const int lvl = b.name[4] - '0', j = b.name[6] - '0';
        b.grad_base = (b.pool) || (lvl == 0 && j >= 1);          // max-pool gradient / ECAM gradient is written first
        if (b.up >= 0) {        // gradient arriving through the transposed conv that up-samples the LOWER block's output
            SnUp& u = e.sn_ups[b.up];
            NBlock& lower = e.sn_blocks[sn_block_index(e, std::string("conv") + char('0' + lvl + 1) + "_" + char('0' + j - 1))];
            ViewRef v; v.off = u.tmp.off; v.ld = u.tmp.ld; v.goff = 0;
            v.gmask = (lower.groups == 2) ? 2 : 1;               // an encoder level feeds its date-1 half only (SNUNet.py:127-142)
            lower.grad_src.push_back(v);
        }
    }
    for (auto& b : e.sn_blocks)
        if ((int)b.extra_dst.size() > MAX_VIEWS || (int)b.grad_src.size() > MAX_VIEWS) { set_error("internal: too many concat views"); return 1; }

    // ---- forward order (SNUNet.py:119-142)
    static const char* ORDER[] = {"conv0_0", "conv1_0", "conv2_0", "conv3_0", "conv4_0", "conv0_1", "conv1_1", "conv0_2", "conv2_1",
                                  "conv1_2", "conv0_3", "conv3_1", "conv2_2", "conv1_3", "conv0_4"};
    e.sn_order.clear();
    for (auto n : ORDER) e.sn_order.push_back(sn_block_index(e, n));

    e.scratch8 = ws.take(256);
    e.masks = ws.take(256);
    e.sn_pool = ws.take((int64_t)B * 4 * c4 * 4); e.sn_argm = ws.take((int64_t)B * 2 * c4 * 8);
    e.sn_att = ws.take((int64_t)B * 2 * c4 * 4); e.sn_hid = ws.take((int64_t)B * 2 * 2 * 16 * 4);
    e.sn_sums = ws.take((int64_t)B * 2 * c4 * 4); e.sn_dpool = ws.take((int64_t)B * 4 * c4 * 4);
    e.sn_part = ws.take(ecam_part_floats(B, c4) * 4);
    for (auto& c : e.convs) {
        c.wpk_fwd = ws.take((int64_t)c.fwd.ntaps * c.fwd.kpad * c.fwd.wld * 4);
        if (c.dgrad.ntaps) c.wpk_dgrad = ws.take((int64_t)c.dgrad.ntaps * c.dgrad.kpad * c.dgrad.wld * 4);
    }
    e.dwe_begin = ws.cur;
    for (auto& c : e.convs) {
        c.dwe_floats = (int64_t)c.fwd.ntaps * c.fwd.kpad * c.fwd.wld;
        c.dwe = ws.take(c.dwe_floats * 8);     // fp64 accumulators of the reference weight-gradient path
    }
    e.dwe_end = ws.cur;

    auto bind_conv = [&](ConvOp& op, const stcd_conv_geom& g, int conv, bool dgrad, int tap0, int kreal, int nreal, int groups = 1) {
        op.g = g; op.conv = conv; op.dgrad = dgrad; op.tap0 = tap0; op.kreal = kreal; op.nreal = nreal;
        op.plan = ConvMfmaPlan(); op.wf = -1; op.small = false; op.res = ConvResPlan(); op.res_groups = groups;
        if (e.dt == BF16) {
            op.plan = conv_mfma_plan(g);
            if (op.plan.ok) op.wf = ws.take(op.plan.wf_elems * 2);
            op.small = conv_small_ok(g, op.plan);
            if (e.use_res && !op.small) op.res = conv_res_plan(g, op.plan, groups);
            pick_gemm_or_res(e, op, g, groups);
        }
        e.conv_ops.push_back(&op);
    };
    auto bind_wgrad = [&](WgradOp& op, const stcd_conv_geom& g, int conv, int tap0, int kreal, int nreal, int64_t in_off,
                          int64_t dout_off) {
        op.g = g; op.conv = conv; op.tap0 = tap0; op.kreal = kreal; op.nreal = nreal;
        op.in_off = in_off; op.dout_off = dout_off; op.grouped = false;
        op.plan = WgradMfmaPlan(); op.slab = -1; op.stage = 0;
        if (e.dt == BF16) op.plan = pick_wgrad_plan(e, g, e.convs[conv].fwd.kpad, e.convs[conv].fwd.wld);
        e.wgrad_ops.push_back(&op);
    };
    for (auto& b : e.sn_blocks) {
        const ConvW& c1 = e.convs[b.c1];
        const ConvW& c2 = e.convs[b.c2];
        bind_conv(b.f1, geom3(b.N, b.H, b.W, b.Cin, b.in.ld, b.C, b.Y1.ld), b.c1, false, 0, c1.cin, b.C, b.groups);
        bind_wgrad(b.w1, geom3(b.N, b.H, b.W, b.Cin, b.in.ld, b.C, b.dA1.ld), b.c1, 0, c1.cin, b.C, b.in.off, b.dA1.off);
        if (b.dIn.off >= 0) bind_conv(b.d1, geom3(b.N, b.H, b.W, c1.dgrad.kpad, b.dA1.ld, c1.cin, b.dIn.ld), b.c1, true, 0, b.C, c1.cin);
        bind_conv(b.f2, geom3(b.N, b.H, b.W, b.C, b.A1.ld, b.C, b.Y2.ld), b.c2, false, 0, b.C, b.C, b.groups);
        bind_wgrad(b.w2, geom3(b.N, b.H, b.W, b.C, b.A1.ld, b.C, b.dOut.ld), b.c2, 0, b.C, b.C, b.A1.off, b.dOut.off);
        bind_conv(b.d2, geom3(b.N, b.H, b.W, c2.dgrad.kpad, b.dOut.ld, b.C, b.dA1.ld), b.c2, true, 0, b.C, b.C);
    }
    for (auto& u : e.sn_ups) {
        if (u.N == 0) continue;
        const ConvW& cv = e.convs[u.conv];
        for (int ph = 0; ph < 4; ++ph) {
            stcd_conv_geom g;
            memset(&g, 0, sizeof(g));
            g.n = u.N; g.hi = u.h; g.wi = u.w; g.ci = u.C; g.ldi = u.src.ld;
            g.hm = u.h; g.wm = u.w; g.in_stride = 1;
            g.ho = 2 * u.h; g.wo = 2 * u.w; g.out_stride = 2; g.oy0 = ph >> 1; g.ox0 = ph & 1;
            g.co = u.C; g.ldo = u.out.ld;
            g.ntaps = 1; g.dy[0] = 0; g.dx[0] = 0;
            bind_conv(u.fwd[ph], g, u.conv, false, ph, u.C, u.C);
            bind_wgrad(u.wg[ph], g, u.conv, ph, u.C, u.C, u.src.off, u.dOut.off);
        }
        stcd_conv_geom gd;
        memset(&gd, 0, sizeof(gd));
        gd.n = u.N; gd.hi = 2 * u.h; gd.wi = 2 * u.w; gd.ci = cv.dgrad.kpad; gd.ldi = u.dOut.ld;
        gd.hm = u.h; gd.wm = u.w; gd.in_stride = 2;
        gd.ho = u.h; gd.wo = u.w; gd.out_stride = 1;
        gd.co = u.C; gd.ldo = u.tmp.ld;
        gd.ntaps = 4;
        for (int t = 0; t < 4; ++t) { gd.dy[t] = (int8_t)(t >> 1); gd.dx[t] = (int8_t)(t & 1); }
        bind_conv(u.dgr, gd, u.conv, true, 0, u.C, u.C);
    }
    {
        const ConvW& cv = e.convs[e.sn_final];
        bind_conv(e.sn_final_fwd, geom1(B, H, W, c4, e.snZ.ld, e.label, e.label), e.sn_final, false, 0, c4, e.label);
        bind_wgrad(e.sn_final_wg, geom1(B, H, W, c4, e.snZ.ld, e.label, 8), e.sn_final, 0, c4, e.label, e.snZ.off, e.G.off);
        bind_conv(e.sn_final_dgr, geom1(B, H, W, cv.dgrad.kpad, 8, c4, e.sndZ.ld), e.sn_final, true, 0, e.label, c4);
    }
    e.slab = ws.take(e.slab_floats * 4);
    // bias gradients of the transposed convs: summed by the consumer block's conv1 data gradient (see configure_fcsiam)
    e.bias_jobs.clear();
    for (auto& b : e.sn_blocks) {
        b.d1_sum_acc = -1;
        if (b.up < 0) continue;
        SnUp& u = e.sn_ups[b.up];
        u.bias_fused = false;
        if (e.dt == BF16 && e.use_mfma && b.dIn.off >= 0 && b.d1.res.ok && b.d1.wf >= 0) {
            u.bias_fused = true;
            b.d1_sum_acc = u.bias_acc; b.d1_sum_c0 = u.coff; b.d1_sum_C = u.C;
        }
        if (e.dt == BF16) {
            BiasJob jb{}; jb.acc_off = u.bias_acc; jb.out_off = e.convs[u.conv].b_off; jb.C = u.C; jb.valid = u.C; jb.scale = BN_BS;
            e.bias_jobs.push_back(jb);
        }
    }
    {
        BiasJob jb{}; jb.acc_off = e.final_bias_acc; jb.out_off = e.convs[e.sn_final].b_off; jb.C = 8; jb.valid = e.label; jb.scale = BN_BS;
        e.bias_jobs.push_back(jb);
        e.bias_jobs_off = ws.take((int64_t)e.bias_jobs.size() * sizeof(BiasJob) + 16);
    }
    build_pack_jobs(e, ws);
    e.jobs_uploaded_ws = nullptr;
    e.ws_bytes = ws.cur;
    return 0;
}

static void sn_block_forward(const Ctx& c, const NBlock& b, float* bn_running, bool training) {
    stcd_engine& e = c.e;
    const int64_t T = (int64_t)dsize(e.dt), px = (int64_t)b.N * b.H * b.W, ppg = (int64_t)b.npg * b.H * b.W;
    const double act_bytes = (double)px * b.C * (double)T;
    (void)px; (void)T;      // (the concat prefix of b.in was written by the producers: NBlock::extra_dst)
    BnActArgs a;
    auto bn_stage = [&](const ConvOp& f, const void* in, int conv, int bni, const TRef& Y, int64_t stat_off, int64_t facc_off) {
        const ConvW& cv = e.convs[conv];
        const BnP& bn = e.bns[bni];
        int fused = 0;
        StatReq sr; sr.acc = training ? c.at<long long>(facc_off) : nullptr; sr.groups = b.groups; sr.C = b.C;
        exec_conv(c, f, in, c.params + cv.b_off, c.at(Y.off), false, &sr, &fused);
        float* stat = c.at<float>(stat_off);
        if (training) {
            if (!fused) {
                ProfScope ps(c, PC_BN_STATS, 0.0, act_bytes);
                launch_bn_stats(e.dt, c.at(Y.off), Y.ld, b.C, b.groups, ppg, c.at<long long>(facc_off), c.s);
            }
            a.facc = c.at<long long>(facc_off); a.gamma = c.params + bn.g_off; a.beta = c.params + bn.b_off;
            a.running_mean = bn_running + bn.run_off; a.running_var = bn_running + bn.run_off + b.C;
        } else {      // eval: the activation kernel reads the running statistics itself
            a.facc = nullptr; a.gamma = c.params + bn.g_off; a.beta = c.params + bn.b_off;
            a.running_mean = bn_running + bn.run_off; a.running_var = bn_running + bn.run_off + b.C;
        }
    };
    bn_stage(b.f1, c.at(b.in.off), b.c1, b.bn1, b.Y1, b.stat1, b.facc1);
    a.Y = c.at(b.Y1.off); a.ldy = b.Y1.ld; a.A = c.at(b.A1.off); a.lda = b.A1.ld; a.a_group_off = ppg * b.A1.ld;
    a.P = nullptr; a.ldp = 0; a.stat = c.at<float>(b.stat1); a.mask = nullptr;
    a.C = b.C; a.groups = b.groups; a.npg = b.npg; a.H = b.H; a.W = b.W; a.relu = 1;
    {
        ProfScope ps(c, PC_BN_ACT, 0.0, 2.0 * act_bytes);
        launch_bn_act(e.dt, a, c.s);
    }
    bn_stage(b.f2, c.at(b.A1.off), b.c2, b.bn2, b.Y2, b.stat2, b.facc2);
    a.Y = c.at(b.Y2.off); a.ldy = b.Y2.ld; a.A = c.at(b.Out.off); a.lda = b.Out.ld; a.a_group_off = ppg * b.Out.ld;
    a.P = b.pool ? c.at(b.P.off) : nullptr; a.ldp = b.P.ld; a.stat = c.at<float>(b.stat2);
    a.res = c.at(b.Y1.off); a.ldres = b.Y1.ld;        // identity = conv1's raw output (SNUNet.py:19,25)
    a.extra.n = (int)b.extra_dst.size();
    for (int k = 0; k < a.extra.n; ++k) {
        a.extra.p[k] = c.at(b.extra_dst[k].off); a.extra.ld[k] = b.extra_dst[k].ld; a.extra.goff[k] = b.extra_dst[k].goff;
        a.extra.gmask[k] = b.extra_dst[k].gmask;
    }
    {
        ProfScope ps(c, PC_BN_ACT, 0.0, (b.pool ? 3.25 : 3.0) * act_bytes);
        launch_bn_act(e.dt, a, c.s);
    }
}

static void sn_block_backward(const Ctx& c, const NBlock& b) {
    stcd_engine& e = c.e;
    const int64_t T = (int64_t)dsize(e.dt), HW = (int64_t)b.H * b.W, px = (int64_t)b.N * HW, ppg = (int64_t)b.npg * HW;
    const double act_bytes = (double)px * b.C * (double)T;
    const BnP& bn1 = e.bns[b.bn1];
    const BnP& bn2 = e.bns[b.bn2];
    (void)ppg;
    const int64_t goff_out = ppg * b.dOut.ld, goff_a1 = ppg * b.dA1.ld;
    // out = relu(bn2(y2) + y1): gate on z2 + y1; dZ2 (gated) is also the gradient of the identity branch
    {
        ProfScope ps(c, PC_BN_BWD_REDUCE, 0.0, 3.0 * act_bytes);
        SliceViews xs;
        xs.n = (int)b.grad_src.size();
        for (int k = 0; k < xs.n; ++k) {
            xs.p[k] = c.at(b.grad_src[k].off); xs.ld[k] = b.grad_src[k].ld; xs.goff[k] = b.grad_src[k].goff; xs.gmask[k] = b.grad_src[k].gmask;
        }
        launch_bn_bwd_reduce(e.dt, c.at(b.dOut.off), b.dOut.ld, goff_out, c.at(b.Y2.off), b.Y2.ld, c.at<float>(b.stat2), nullptr, b.C,
                             b.groups, b.npg, HW, 1, c.at<long long>(b.bacc2), c.s, c.at(b.Y1.off), b.Y1.ld, &xs, b.grad_base ? 1 : 0,
                             c.at(b.dOut.off));
    }
    {
        ProfScope ps(c, PC_BN_BWD_APPLY, 0.0, 5.0 * act_bytes);
        launch_bn_bwd_apply(e.dt, c.at(b.dOut.off), b.dOut.ld, goff_out, c.at(b.dOut.off), b.dOut.ld, c.at(b.Y2.off), b.Y2.ld,
                            c.at<float>(b.stat2), c.at<long long>(b.bacc2), c.grads + bn2.g_off, c.grads + bn2.b_off, nullptr, b.C,
                            b.groups, b.npg, HW, 1, c.s, c.at(b.Y1.off), b.Y1.ld, c.at(b.dZ2.off), b.dZ2.ld, nullptr, 0,
                            c.grads + e.convs[b.c1].b_off);
    }
    exec_wgrad(c, b.w2, c.at(b.A1.off), c.at(b.dOut.off));                   // dOut now holds dY2
    exec_conv(c, b.d2, c.at(b.dOut.off), nullptr, c.at(b.dA1.off), false);
    {
        ProfScope ps(c, PC_BN_BWD_REDUCE, 0.0, 2.0 * act_bytes);
        launch_bn_bwd_reduce(e.dt, c.at(b.dA1.off), b.dA1.ld, goff_a1, c.at(b.Y1.off), b.Y1.ld, c.at<float>(b.stat1), nullptr, b.C,
                             b.groups, b.npg, HW, 1, c.at<long long>(b.bacc1), c.s);
    }
    {
        ProfScope ps(c, PC_BN_BWD_APPLY, 0.0, 4.0 * act_bytes);
        launch_bn_bwd_apply(e.dt, c.at(b.dA1.off), b.dA1.ld, goff_a1, c.at(b.dA1.off), b.dA1.ld, c.at(b.Y1.off), b.Y1.ld,
                            c.at<float>(b.stat1), c.at<long long>(b.bacc1), c.grads + bn1.g_off, c.grads + bn1.b_off, nullptr, b.C,
                            b.groups, b.npg, HW, 1, c.s, nullptr, 0, nullptr, 0, c.at(b.dZ2.off), b.dZ2.ld);
    }
    // conv1's bias reaches the loss only through the identity branch (its BN path has zero gradient): db1 = sum dZ2,
    // and dZ2 is exactly the gated gradient whose per-channel sum bn2's backward already formed as d(beta2): bn2's apply
    // kernel wrote it to both places (a 128-byte hipMemcpyAsync cost three copy kernels per block, 44 launches per step)
    exec_wgrad(c, b.w1, c.at(b.in.off), c.at(b.dA1.off));                   // dA1 now holds dY1
    if (b.dIn.off >= 0) {
        StatReq sr;
        if (b.d1_sum_acc >= 0) { sr.acc = c.at<long long>(b.d1_sum_acc); sr.groups = 1; sr.c0 = b.d1_sum_c0; sr.C = b.d1_sum_C; sr.s1 = sr.s2 = BN_BS; }
        exec_conv(c, b.d1, c.at(b.dA1.off), nullptr, c.at(b.dIn.off), false, b.d1_sum_acc >= 0 ? &sr : nullptr);
    }
}

static void sn_up_forward(const Ctx& c, const SnUp& u) {
    const ConvW& cv = c.e.convs[u.conv];
    if (exec_conv_x4(c, u.fwd, c.at(u.src.off), c.params + cv.b_off, c.at(u.out.off))) return;
    for (int ph = 0; ph < 4; ++ph) exec_conv(c, u.fwd[ph], c.at(u.src.off), c.params + cv.b_off, c.at(u.out.off), false);
}
static void sn_up_backward(const Ctx& c, const SnUp& u) {
    stcd_engine& e = c.e;
    const ConvW& cv = e.convs[u.conv];
    const int64_t T = (int64_t)dsize(e.dt);
    if (!(u.bias_fused && mfma_on(e)))
        launch_bias_grad(e.dt, c.at(u.dOut.off), u.dOut.ld, (int64_t)u.N * 4 * u.h * u.w, u.C, c.grads + cv.b_off, c.s,
                         e.dt == BF16 ? c.at<long long>(u.bias_acc) : nullptr);
    for (int ph = 0; ph < 4; ++ph) exec_wgrad(c, u.wg[ph], c.at(u.src.off), c.at(u.dOut.off));
    exec_conv(c, u.dgr, c.at(u.dOut.off), nullptr, c.at(u.tmp.off), false);     // summed into the lower block's dOut by ITS reduction
    (void)T;
}

static int forward_snunet(stcd_engine& e, const float* x1, const float* x2, const float* params, float* bn_running, int training,
                          float* logits, void* workspace, hipStream_t s) {
    Ctx c{e, (char*)workspace, params, nullptr, s};
    if (pack_all_weights(c, training != 0)) return 1;
    if (training) STCD_HIP(hipMemsetAsync(c.at(e.zero_begin), 0, e.zero_end - e.zero_begin, s));
    launch_in_pack(e.dt, x1, x2, c.at(e.X0.off), e.B, e.in_ch, e.H, e.W, s);
    for (int bi : e.sn_order) {
        const NBlock& b = e.sn_blocks[bi];
        if (b.up >= 0) sn_up_forward(c, e.sn_ups[b.up]);
        sn_block_forward(c, b, bn_running, training != 0);
    }
    const int c4 = SN_F[0] * 4;
    launch_ecam_forward(e.dt, c.at(e.snE.off), e.snE.ld, c.at(e.snZ.off), e.snZ.ld, e.B, (int64_t)e.H * e.W, c4, params + e.sn_w[0],
                        params + e.sn_w[1], params + e.sn_w[2], params + e.sn_w[3], c.at<float>(e.sn_pool), c.at<int64_t>(e.sn_argm),
                        c.at<float>(e.sn_att), c.at<float>(e.sn_hid), c.at<float>(e.sn_part), s);
    exec_conv(c, e.sn_final_fwd, c.at(e.snZ.off), params + e.convs[e.sn_final].b_off, logits, true);
    STCD_HIP(hipGetLastError());
    return 0;
}

static int backward_snunet(stcd_engine& e, const float* grad_logits, const float* params, float* grads, void* workspace, int stage,
                           hipStream_t s) {
    if (stage == 1) return 0;            // single-stage plan: everything is final after stage 0 / -1
    Ctx c{e, (char*)workspace, params, grads, s};
    // single call, no gradient hook: the grouped weight gradients go out with their last member on the engine's side stream (the
    // groups whose members all sit deep in the backward order -- the GEMM groups, the 32 x 32 tile group -- run beside the rest of
    // the chain: 13.22 -> 12.80 ms; cutting the groups into buckets of the backward order, or smaller grids, added nothing)
    hipStream_t side = stage < 0 ? wgrad_side_stream(e, s) : nullptr;
    static const bool early_on = [] { const char* v = getenv("STCD_WGRAD_EARLY"); return !(v && v[0] == '0'); }();
    EarlyScope early_scope(e, early_on && mfma_on(e) && e.use_wgroup, side);
    for (WgradGroup& G : e.wgroups[0]) { G.seen = 0; G.launched = false; }
    const int dt = e.dt, B = e.B;
    const int64_t T = (int64_t)dsize(dt);
    const int c4 = SN_F[0] * 4;
    STCD_HIP(hipMemsetAsync(grads, 0, e.param_floats * 4, s));
    if (!mfma_on(e)) STCD_HIP(hipMemsetAsync(c.at(e.dwe_begin), 0, e.dwe_end - e.dwe_begin, s));
    // (no bulk zeroing of the dOut buffers: each is fully written -- max-pool / ECAM gradient, or the gathered sum)
    for (auto& b : e.sn_blocks)          // pooled gradients of the A half of conv3_0 never get written (x4_0A does not exist)
        if (b.pool && b.name == "conv3_0") STCD_HIP(hipMemsetAsync(c.at(b.dP.off), 0, (int64_t)B * (b.H / 2) * (b.W / 2) * b.C * T, s));
    launch_gout_pack(dt, grad_logits, c.at(e.G.off), B, e.label, e.H, e.W, s, c.at<long long>(e.final_bias_acc));
    exec_wgrad(c, e.sn_final_wg, c.at(e.snZ.off), c.at(e.G.off));
    exec_conv(c, e.sn_final_dgr, c.at(e.G.off), nullptr, c.at(e.sndZ.off), false);
    launch_ecam_backward(dt, c.at(e.snE.off), e.snE.ld, c.at(e.sndZ.off), e.sndZ.ld, c.at(e.sndE.off), e.sndE.ld, B, (int64_t)e.H * e.W, c4,
                         params + e.sn_w[0], params + e.sn_w[1], params + e.sn_w[2], params + e.sn_w[3], grads + e.sn_w[0],
                         grads + e.sn_w[1], grads + e.sn_w[2], grads + e.sn_w[3], c.at<float>(e.sn_pool), c.at<int64_t>(e.sn_argm),
                         c.at<float>(e.sn_att), c.at<float>(e.sn_hid), c.at<float>(e.sn_sums), c.at<float>(e.sn_dpool), c.at<float>(e.sn_part), s);
    for (int oi = (int)e.sn_order.size() - 1; oi >= 0; --oi) {
        const NBlock& b = e.sn_blocks[e.sn_order[oi]];
        if (b.pool) {    // gradient coming back through the 2x2 max-pool of this encoder output
            ProfScope ps(c, PC_POOL_FUSE, 0.0, 3.25 * b.N * b.H * b.W * b.C * (double)T);
            const int64_t goff = (int64_t)b.npg * b.H * b.W;
            launch_pool_bwd(dt, c.at(b.Out.off), b.Out.ld, goff * b.Out.ld, c.at(b.dP.off), b.dP.ld, c.at(b.dOut.off), b.dOut.ld,
                            goff * b.dOut.ld, b.groups, b.npg, b.H, b.W, b.C, 0, s);
        }
        sn_block_backward(c, b);       // (its concat gradient b.dIn is gathered by the producers' reductions later on)
        if (b.up >= 0) sn_up_backward(c, e.sn_ups[b.up]);
    }
    if (side) {
        STCD_HIP(hipEventRecord(e.wg_fork, s));
        STCD_HIP(hipStreamWaitEvent(side, e.wg_fork, 0));
        Ctx cs{e, (char*)workspace, params, grads, side};
        reduce_stage(cs, 0);
        launch_bias_finish(c.at<BiasJob>(e.bias_jobs_off), (int)e.bias_jobs.size(), c.ws, c.grads, side);
        STCD_HIP(hipEventRecord(e.wg_join, side));
        STCD_HIP(hipStreamWaitEvent(s, e.wg_join, 0));
    } else {
        reduce_stage(c, 0);
        launch_bias_finish(c.at<BiasJob>(e.bias_jobs_off), (int)e.bias_jobs.size(), c.ws, c.grads, s);
    }
    STCD_HIP(hipGetLastError());
    return 0;
}


// ================================================================================================ SegCD (ResNet-50 UNet)
// smp.SegCD(encoder_name="resnet50"): the model the reference's scripts train (train_pse_cd.py:419-427, train_stcd.py:631-638).
//   SegCD.forward            /root/reference/segmentation_models_pytorch/decoders/unet/model.py:316-332
//   ResNet-50 encoder        /root/reference/segmentation_models_pytorch/encoders/resnet.py:37-70, /root/reference/models/resnet.py:78-190
//   UnetDecoder              /root/reference/segmentation_models_pytorch/decoders/unet/decoder.py:8-123
// Both dates run as ONE batch of 2B images through the shared encoder + decoder (every BatchNorm keeps per-date statistics:
// the modules are called once per date); only the head combines them.  Gradients of tensors with several consumers
// (identity branches, decoder skips) are never accumulated by extra kernels: every consumer writes its own contribution
// buffer and the producer's BatchNorm-backward reduction gathers them (SliceViews).
static const int RS_PLANES[4] = {64, 128, 256, 512};
static bool is_unetseg(int arch) { return arch >= STCD_ARCH_UNETSEG && arch <= STCD_ARCH_UNETSEG + 4; }
static bool is_ffctlcd(int arch) { return arch >= STCD_ARCH_FFCTLCD && arch <= STCD_ARCH_FFCTLCD + 4; }
static bool is_segcd(int arch) { return (arch >= STCD_ARCH_SEGCD && arch <= STCD_ARCH_SEGCD_R152) || is_unetseg(arch) || is_ffctlcd(arch); }

static const int SEG_DEC[5] = {256, 128, 64, 32, 16};

static int add_glayer(stcd_engine& e, const std::string& conv_name, const std::string& bn_name, int kind, int cin, int cout, bool relu,
                      bool need_dgrad, int bn_calls = 0) {
    GLayer L;
    L.name = conv_name; L.kind = kind; L.relu = relu; L.has_dIn = need_dgrad;
    L.conv = add_conv(e, conv_name, kind, cin, cout, need_dgrad, false);
    L.bn = add_bn(e, bn_name, cout, bn_calls ? bn_calls : e.seg_dates);
    e.g_layers.push_back(L);
    return (int)e.g_layers.size() - 1;
}

// encoders/resnet.py:126-171: block type and depths of the plain ResNet registry entries
static void segcd_encoder_cfg(stcd_engine& e) {
    static const int L18[4] = {2, 2, 2, 2}, L34[4] = {3, 4, 6, 3}, L101[4] = {3, 4, 23, 3}, L152[4] = {3, 8, 36, 3};
    const int* l = L34;
    e.seg_x = 4;
    e.seg_dates = is_unetseg(e.arch) ? 1 : 2;
    e.seg_ffc = is_ffctlcd(e.arch);
    switch (is_unetseg(e.arch) ? e.arch - STCD_ARCH_UNETSEG + STCD_ARCH_SEGCD
            : is_ffctlcd(e.arch) ? e.arch - STCD_ARCH_FFCTLCD + STCD_ARCH_SEGCD : e.arch) {      // same encoder order in every id range
        case STCD_ARCH_SEGCD_R18: e.seg_x = 1; l = L18; break;
        case STCD_ARCH_SEGCD_R34: e.seg_x = 1; l = L34; break;
        case STCD_ARCH_SEGCD_R101: l = L101; break;
        case STCD_ARCH_SEGCD_R152: l = L152; break;
        default: break;                                                // resnet50: Bottleneck, [3, 4, 6, 3]
    }
    for (int i = 0; i < 4; ++i) e.seg_layers[i] = l[i];
}

static void build_segcd_tables(stcd_engine& e) {
    segcd_encoder_cfg(e);
    const int X = e.seg_x;
    e.g_layers.clear(); e.g_blocks.clear(); e.g_dec.clear();
    e.g_stem = add_glayer(e, "encoder.conv1", "encoder.bn1", K_STEM7, e.in_ch, 64, true, false);
    int inpl = 64;
    for (int li = 0; li < 4; ++li)
        for (int b = 0; b < e.seg_layers[li]; ++b) {
            const int w = RS_PLANES[li], stride = (b == 0 && li > 0) ? 2 : 1;
            const std::string pre = "encoder.layer" + std::to_string(li + 1) + "." + std::to_string(b);
            // ResNet._make_layer (models/resnet.py:165-187): a down-sample where the stride or the width changes
            const bool down = b == 0 && (stride != 1 || inpl != w * X);
            std::array<int, 4> blk;
            if (X == 4) {      // Bottleneck (models/resnet.py:78-124): 1x1, 3x3 (carries the stride), 1x1
                blk[0] = add_glayer(e, pre + ".conv1", pre + ".bn1", K_CONV1, inpl, w, true, true);
                blk[1] = add_glayer(e, pre + ".conv2", pre + ".bn2", stride == 2 ? K_CONV3_S2 : K_CONV3, w, w, true, true);
                blk[2] = add_glayer(e, pre + ".conv3", pre + ".bn3", K_CONV1, w, 4 * w, true, true);
            } else {           // BasicBlock (models/resnet.py:37-75): 3x3 (carries the stride), 3x3
                blk[0] = add_glayer(e, pre + ".conv1", pre + ".bn1", stride == 2 ? K_CONV3_S2 : K_CONV3, inpl, w, true, true);
                blk[1] = -1;
                blk[2] = add_glayer(e, pre + ".conv2", pre + ".bn2", K_CONV3, w, w, true, true);
            }
            blk[3] = down ? add_glayer(e, pre + ".downsample.0", pre + ".downsample.1", stride == 2 ? K_CONV1_S2 : K_CONV1, inpl, X * w, false, true) : -1;
            e.g_blocks.push_back(blk);
            inpl = X * w;
        }
    const int ENC_OUT[5] = {512 * X, 256 * X, 128 * X, 64 * X, 64};
    int cin = ENC_OUT[0];
    for (int i = 0; i < 5; ++i) {
        const int cskip = i < 4 ? ENC_OUT[i + 1] : 0, cout = SEG_DEC[i];
        const std::string pre = "decoder.blocks." + std::to_string(i);
        std::array<int, 2> d;
        // FFCTLCD (model.py:407-423) runs the decoder three times per forward: |f1 - f2| first, then date 1, then date 2
        d[0] = add_glayer(e, pre + ".conv1.0", pre + ".conv1.1", K_CONV3, cin + cskip, cout, true, true, e.seg_ffc ? 3 : 0);
        d[1] = add_glayer(e, pre + ".conv2.0", pre + ".conv2.1", K_CONV3, cout, cout, true, true, e.seg_ffc ? 3 : 0);
        e.g_dec.push_back(d);
        cin = cout;
    }
    e.g_head_conv = add_conv(e, "segmentation_head.0", K_CONV3, SEG_DEC[4], e.label, true, true);
    e.enc_param_end = 0;      // one backward stage
}

static int configure_segcd(stcd_engine& e, int B, int H, int W) {
    const int64_t T = (int64_t)dsize(e.dt);
    const int D = e.seg_dates, N = D * B;
    e.drops.clear(); e.drop_floats = 0;
    e.conv_ops.clear(); e.wgrad_ops.clear(); e.slab_floats = 0;
    e.g_fwd.clear();
    Bump ws;
    auto plain = [&](int n, int h, int w, int C) { TRef t; t.off = ws.take((int64_t)n * h * w * C * T); t.ld = C; return t; };
    auto view = [&](const TRef& t, int coff, int ld, int h, int w) {       // both dates of a plain [2B,h,w,ld] tensor, channel offset coff
        ViewRef v; v.off = t.off + (int64_t)coff * T; v.ld = ld; v.goff = D == 2 ? (int64_t)B * h * w * ld : 0; v.gmask = D == 2 ? 3 : 1; return v;
    };
    for (auto& L : e.g_layers) { L.extra_dst.clear(); L.grad_src.clear(); L.grad_base = true; L.res = TRef(); L.dRes = TRef(); L.ndgr = 0; }
    e.X0 = plain(N, H, W, 8);
    // ---- shapes + activation buffers, forward order
    auto shape = [&](GLayer& L, const TRef& in, int K, int hi, int wi, int stride, int ng = 0) {
        const ConvW& cv = e.convs[L.conv];
        if (ng == 0) ng = D;
        const int n = ng * B;
        L.N = n; L.groups = ng; L.npg = B; L.in = in; L.K = K; L.Hi = hi; L.Wi = wi; L.Ho = hi / stride; L.Wo = wi / stride; L.C = cv.cout;
        L.Y = plain(n, L.Ho, L.Wo, L.C); L.A = plain(n, L.Ho, L.Wo, L.C); L.dA = plain(n, L.Ho, L.Wo, L.C);
        L.stat = ws.take((int64_t)std::max(2, ng) * 4 * L.C * 4); L.coef = ws.take((int64_t)std::max(2, ng) * 5 * L.C * 4);
        L.g_first = 0;
    };
    GLayer& S = e.g_layers[e.g_stem];
    shape(S, e.X0, 8, H, W, 2);
    S.has_dIn = false;
    { GStep st; st.kind = GS_LAYER; st.layer = e.g_stem; e.g_fwd.push_back(st); }
    e.gP0 = plain(N, H / 4, W / 4, 64); e.gdP0 = plain(N, H / 4, W / 4, 64);
    e.g_pool_idx = ws.take((int64_t)N * (H / 4) * (W / 4) * 64);
    e.g_stem_part = ws.take(stem_wgrad_part_floats(N, H, W, e.in_ch, 64) * 4);
    { GStep st; st.kind = GS_MAXPOOL; st.src = S.A; st.dst = e.gP0; st.dsrc = S.dA; st.ddst = e.gdP0; st.N = N; st.h = H / 2; st.w = W / 2; st.C = 64; e.g_fwd.push_back(st); }
    TRef cur = e.gP0; int curC = 64, h = H / 4, w = W / 4, prev_out = -1;      // prev_out: layer whose A is `cur` (-1: the max-pool)
    std::vector<int> stage_out;                                               // L3 of the last block of layer1..4
    size_t bi = 0;
    for (int li = 0; li < 4; ++li)
        for (int b = 0; b < e.seg_layers[li]; ++b, ++bi) {
            const std::array<int, 4>& blk = e.g_blocks[bi];
            const int stride = (b == 0 && li > 0) ? 2 : 1;
            const int ho = h / stride, wo = w / stride;
            GLayer& L1 = e.g_layers[blk[0]]; GLayer& L3 = e.g_layers[blk[2]];
            GLayer* L2 = blk[1] >= 0 ? &e.g_layers[blk[1]] : nullptr;
            if (L2) {          // Bottleneck: the 3x3 in the middle carries the stride
                shape(L1, cur, curC, h, w, 1);
                shape(*L2, L1.A, L1.C, h, w, stride);
            } else {           // BasicBlock: the first 3x3 carries it
                shape(L1, cur, curC, h, w, stride);
            }
            if (blk[3] >= 0) { GLayer& Ld = e.g_layers[blk[3]]; shape(Ld, cur, curC, h, w, stride); }
            GLayer& Lm = L2 ? *L2 : L1;                                        // producer of the last conv's input
            shape(L3, Lm.A, Lm.C, ho, wo, 1);
            L3.res = blk[3] >= 0 ? e.g_layers[blk[3]].A : cur;
            // gradient wiring: single-consumer tensors are written in place, the block input gets contribution buffers
            if (e.debug_flags & 1) {
                if (L2) { L2->dIn = plain(N, h, w, L1.C); L1.grad_base = false; L1.grad_src.push_back(view(L2->dIn, 0, L1.C, h, w)); }
                L3.dIn = plain(N, ho, wo, Lm.C); Lm.grad_base = false; Lm.grad_src.push_back(view(L3.dIn, 0, Lm.C, ho, wo));
            } else { if (L2) L2->dIn = L1.dA; L3.dIn = Lm.dA; }
            L1.dIn = (prev_out < 0) ? e.gdP0 : plain(N, h, w, curC);
            TRef idc;                                                          // identity-branch contribution to d(cur)
            if (blk[3] >= 0) { GLayer& Ld = e.g_layers[blk[3]]; L3.dRes = Ld.dA; Ld.dIn = plain(N, h, w, curC); idc = Ld.dIn; }
            else { L3.dRes = plain(N, h, w, curC); idc = L3.dRes; }
            if (prev_out >= 0) {
                GLayer& Pv = e.g_layers[prev_out];
                Pv.grad_base = false;
                Pv.grad_src.push_back(view(L1.dIn, 0, curC, h, w));
                Pv.grad_src.push_back(view(idc, 0, curC, h, w));
            } else e.g_pool_idc = idc;
            { GStep st; st.layer = blk[0]; e.g_fwd.push_back(st); }
            if (L2) { GStep st; st.layer = blk[1]; e.g_fwd.push_back(st); }
            if (blk[3] >= 0) { GStep st; st.layer = blk[3]; e.g_fwd.push_back(st); }
            { GStep st; st.layer = blk[2]; e.g_fwd.push_back(st); }
            cur = L3.A; curC = L3.C; h = ho; w = wo; prev_out = blk[2];
            if (b == e.seg_layers[li] - 1) stage_out.push_back(blk[2]);
        }
    // ---- decoder: x = f5; block i: nearest x2 into cat_i[:, :Cx], skip (written by its encoder producer) in cat_i[:, Cx:]
    const int skip_layer[4] = {stage_out[2], stage_out[1], stage_out[0], e.g_stem};          // f4, f3, f2, f1
    int xl = stage_out[3];                                                                  // layer producing x
    e.g_layers[xl].grad_base = true;                                                        // f5: the up-sampling gradient alone
    const bool ffc = e.seg_ffc;
    const int DG = ffc ? 3 : D, ND = DG * B;                                                // decoder groups / images
    auto absdiff = [&](const TRef& val, const TRef& grad, int coff, int C, int h_, int w_, GLayer& producer) {
        // third group of `val` (channels [coff, coff + C)) = |date 0 - date 1|; its gradient returns to `producer` as a contribution
        TRef tmp = plain(N, h_, w_, C);
        GStep st; st.kind = GS_ABSDIFF; st.N = B; st.h = h_; st.w = w_; st.C = C;
        st.src = val; st.src.off += (int64_t)coff * T; st.dsrc = grad; st.dsrc.off += (int64_t)coff * T; st.ddst = tmp;
        e.g_fwd.push_back(st);
        producer.grad_src.push_back(view(tmp, 0, C, h_, w_));
    };
    if (ffc) {      // x = [f5(date 0); f5(date 1); |f5 - f5|]: the last encoder layer writes the two dates into a 3B-image tensor
        GLayer& X = e.g_layers[xl];
        X.A = plain(ND, X.Ho, X.Wo, X.C); X.dA = plain(ND, X.Ho, X.Wo, X.C);
        absdiff(X.A, X.dA, 0, X.C, X.Ho, X.Wo, X);
    }
    for (int i = 0; i < 5; ++i) {
        GLayer& X = e.g_layers[xl];
        const int Cx = X.C, hh = 2 * X.Ho, ww = 2 * X.Wo;
        const int Cs = i < 4 ? e.g_layers[skip_layer[i]].C : 0;
        TRef cat = plain(ND, hh, ww, Cx + Cs), dcat = plain(ND, hh, ww, Cx + Cs);
        { GStep st; st.kind = GS_UPSAMPLE; st.src = X.A; st.dst = cat; st.dsrc = X.dA; st.ddst = dcat; st.N = ND; st.h = X.Ho; st.w = X.Wo; st.C = Cx; e.g_fwd.push_back(st); }
        if (i < 4) {
            GLayer& Sk = e.g_layers[skip_layer[i]];
            Sk.extra_dst.push_back(view(cat, Cx, Cx + Cs, hh, ww));
            Sk.grad_src.push_back(view(dcat, Cx, Cx + Cs, hh, ww));
            if (ffc) absdiff(cat, dcat, Cx, Cs, hh, ww, Sk);
        }
        GLayer& D1 = e.g_layers[e.g_dec[i][0]]; GLayer& D2 = e.g_layers[e.g_dec[i][1]];
        shape(D1, cat, Cx + Cs, hh, ww, 1, DG);
        shape(D2, D1.A, D1.C, hh, ww, 1, DG);
        if (ffc) D1.g_first = D2.g_first = 2;          // model.py:413-419: decoder(|f1 - f2|) runs before decoder(f1), decoder(f2)
        D1.dIn = dcat;
        if (e.debug_flags & 1) {
            D2.dIn = plain(ND, hh, ww, D1.C); D1.grad_base = false;
            ViewRef v = view(D2.dIn, 0, D1.C, hh, ww);
            if (ffc) v.gmask = 7;
            D1.grad_src.push_back(v);
        } else D2.dIn = D1.dA;
        for (int k : {e.g_dec[i][0], e.g_dec[i][1]}) { GStep st; st.layer = k; e.g_fwd.push_back(st); }
        xl = e.g_dec[i][1];
    }
    // ---- head: X3 = [d1; d2; |d1 - d2|] (3B images), one conv launch; the last decoder layer writes d1, d2 straight into X3
    //      (UnetSeg: X3 = d, B images, the head's output is the network's)
    GLayer& DL = e.g_layers[xl];
    const int nhead = D == 2 ? 3 * B : B;
    e.gX3 = plain(nhead, H, W, 16); e.gdX3 = plain(nhead, H, W, 16);
    DL.A = e.gX3; DL.dA = e.gdX3;
    DL.grad_base = true;
    if (D == 2) {
        if (!ffc) {      // FFCTLCD's last decoder layer already holds [d(f1); d(f2); d(|f1 - f2|)]: the head reads it as it is
            e.gFuseTmp = plain(N, H, W, 16);
            DL.grad_src.push_back(view(e.gFuseTmp, 0, 16, H, W));
        }
        e.g_raw3 = ws.take((int64_t)3 * B * e.label * H * W * 4); e.g_draw3 = ws.take((int64_t)3 * B * e.label * H * W * 4);
    }
    e.G = plain(nhead, H, W, 8);
    for (auto& L : e.g_layers)
        if ((int)L.extra_dst.size() > MAX_VIEWS || (int)L.grad_src.size() > MAX_VIEWS) { set_error("internal: too many views"); return 1; }
    // ---- zero arena: accumulators, tickets-free (consumer-side tables), bias accumulator of the head
    e.zero_begin = ws.cur;
    for (auto& L : e.g_layers) { L.facc = ws.take(bn_acc_bytes(std::max(2, L.groups), L.C)); L.bacc = ws.take(bn_acc_bytes(std::max(2, L.groups), L.C)); }
    e.final_bias_acc = ws.take(bn_acc_bytes(1, 8));
    e.zero_end = ws.cur;
    e.scratch8 = ws.take(256);
    e.masks = ws.take(256);
    for (auto& c : e.convs) {
        c.wpk_fwd = ws.take((int64_t)c.fwd.ntaps * c.fwd.kpad * c.fwd.wld * 4);
        if (c.dgrad.ntaps) c.wpk_dgrad = ws.take((int64_t)c.dgrad.ntaps * c.dgrad.kpad * c.dgrad.wld * 4);
    }
    e.dwe_begin = ws.cur;
    for (auto& c : e.convs) {
        c.dwe_floats = (int64_t)c.fwd.ntaps * c.fwd.kpad * c.fwd.wld;
        c.dwe = ws.take(c.dwe_floats * 8);
    }
    e.dwe_end = ws.cur;
    // ---- bind every conv-shaped launch
    auto bind_conv = [&](ConvOp& op, const stcd_conv_geom& g, int conv, bool dgrad, int tap0, int kreal, int nreal, int groups = 1) {
        op.g = g; op.conv = conv; op.dgrad = dgrad; op.tap0 = tap0; op.kreal = kreal; op.nreal = nreal;
        op.plan = ConvMfmaPlan(); op.wf = -1; op.small = false; op.res = ConvResPlan(); op.res_groups = groups;
        if (e.dt == BF16) {
            op.plan = conv_mfma_plan(g);
            if (op.plan.ok) op.wf = ws.take(op.plan.wf_elems * 2);
            op.small = conv_small_ok(g, op.plan);
            if (e.use_res && !op.small) op.res = conv_res_plan(g, op.plan, groups);
            pick_gemm_or_res(e, op, g, groups);
        }
        e.conv_ops.push_back(&op);
    };
    auto bind_wgrad = [&](WgradOp& op, const stcd_conv_geom& g, int conv, int64_t in_off, int64_t dout_off, int tap0 = 0, int wi_valid = 0) {
        const ConvW& cv = e.convs[conv];
        op.g = g; op.conv = conv; op.tap0 = tap0; op.kreal = cv.cin; op.nreal = cv.cout;
        op.in_off = in_off; op.dout_off = dout_off; op.grouped = false;
        op.plan = WgradMfmaPlan(); op.slab = -1; op.stage = 0;
        if (e.dt == BF16) { op.plan = pick_wgrad_plan(e, g, cv.fwd.kpad, cv.fwd.wld); op.plan.wi_valid = wi_valid; }
        e.wgrad_ops.push_back(&op);
    };
    for (auto& L : e.g_layers) {
        if (L.kind == K_STEM7) {
            // 7x7 stride-2 weight gradient on the MFMA path: row 2y + ky - 3 lies in parity plane py = (ky + 1) & 1 at plane row
            // y + (ky - 3 - py) / 2; per plane 3x3 / 3x4 / 4x3 / 4x4 taps, run as stride-1 launches of <= 9 taps over plane views
            // (see the stride-2 layers below).  fp32 mode (and any plan that does not fit) keeps the dedicated fp32 kernel.
            L.nwg = 0;
            L.fwd = ConvOp();
            if (e.dt == BF16 && e.use_mfma && e.use_gemm) {
                // forward on the tap-list GEMM kernel: tap = filter row ky, and the 7 filter columns x 8 (padded) channels of a
                // row are 112 contiguous bytes of the NHWC8 input -- one 64-"channel" row of 8 pixels starting 3 pixels to the left
                stcd_conv_geom gs;
                memset(&gs, 0, sizeof(gs));
                gs.n = L.N; gs.hi = L.Hi; gs.wi = L.Wi; gs.ci = 64; gs.ldi = L.in.ld;
                gs.hm = L.Ho; gs.wm = L.Wo; gs.in_stride = 2; gs.ho = L.Ho; gs.wo = L.Wo; gs.out_stride = 1;
                gs.co = e.convs[L.conv].cout; gs.ldo = L.Y.ld; gs.ntaps = 7;
                for (int t = 0; t < 7; ++t) { gs.dy[t] = (int8_t)(t - 3); gs.dx[t] = -3; }
                ConvOp& op = L.fwd;
                op.g = gs; op.conv = L.conv; op.dgrad = false; op.tap0 = 0; op.kreal = e.convs[L.conv].cin; op.nreal = gs.co;
                op.plan = conv_mfma_plan(gs); op.res_groups = L.groups;
                op.gemm = (L.in.ld == 8 && gs.co % 64 == 0) ? conv_gemm_plan(gs, op.plan, L.groups) : ConvGemmPlan();
                op.gemm.pix_chunks = 1;
                if (op.gemm.ok) { op.wf = ws.take(op.plan.wf_elems * 2); e.conv_ops.push_back(&op); }
            }
            if (!(e.dt == BF16 && e.use_mfma && e.use_wgroup)) continue;
            const ConvW& cv = e.convs[L.conv];
            bool all_ok = true;
            int nops = 0;
            for (int ph = 0; ph < 4 && all_ok; ++ph) {
                const int py = ph >> 1, px = ph & 1;
                std::vector<std::pair<int, int>> taps;
                for (int ky = 0; ky < 7; ++ky)
                    for (int kx = 0; kx < 7; ++kx)
                        if (((ky + 1) & 1) == py && ((kx + 1) & 1) == px) taps.push_back({ky, kx});
                const int nchunk = ((int)taps.size() + 8) / 9, per = ((int)taps.size() + nchunk - 1) / nchunk;
                for (int c0 = 0; c0 < (int)taps.size(); c0 += per) {
                    const int nt = std::min(per, (int)taps.size() - c0);
                    stcd_conv_geom gw;
                    memset(&gw, 0, sizeof(gw));
                    gw.n = L.N; gw.hi = L.Hi / 2; gw.wi = L.Wi; gw.ci = 8; gw.ldi = 2 * L.in.ld;
                    gw.hm = L.Ho; gw.wm = L.Wo; gw.in_stride = 1; gw.ho = L.Ho; gw.wo = L.Wo; gw.out_stride = 1;
                    gw.co = cv.cout; gw.ldo = L.dA.ld; gw.ntaps = nt;
                    WgradOp& op = L.wg[nops];
                    for (int t = 0; t < nt; ++t) {
                        const int ky = taps[c0 + t].first, kx = taps[c0 + t].second;
                        gw.dy[t] = (int8_t)((ky - 3 - py) / 2); gw.dx[t] = (int8_t)((kx - 3 - px) / 2);
                        op.oky[t] = (int8_t)ky; op.okx[t] = (int8_t)kx;
                    }
                    bind_wgrad(op, gw, L.conv, L.in.off + ((int64_t)py * L.Wi + px) * L.in.ld * T, L.dA.off, 0, L.Wi / 2);
                    op.own_taps = true;
                    all_ok = all_ok && op.plan.ok;
                    ++nops;
                }
            }
            if (all_ok) L.nwg = nops;
            else for (int k = 0; k < nops; ++k) e.wgrad_ops.pop_back();
            continue;
        }
        const ConvW& cv = e.convs[L.conv];
        stcd_conv_geom g;
        const int stride = L.Hi / L.Ho;
        if (L.kind == K_CONV3 && stride == 1) g = geom3(L.N, L.Hi, L.Wi, L.K, L.in.ld, cv.cout, L.Y.ld);
        else if (L.kind == K_CONV1 && stride == 1) g = geom1(L.N, L.Hi, L.Wi, L.K, L.in.ld, cv.cout, L.Y.ld);
        else {      // stride-2 gathers (3x3 p1 or 1x1)
            memset(&g, 0, sizeof(g));
            g.n = L.N; g.hi = L.Hi; g.wi = L.Wi; g.ci = L.K; g.ldi = L.in.ld;
            g.hm = L.Ho; g.wm = L.Wo; g.in_stride = 2; g.ho = L.Ho; g.wo = L.Wo; g.out_stride = 1;
            g.co = cv.cout; g.ldo = L.Y.ld;
            if (L.kind == K_CONV3_S2) { g.ntaps = 9; for (int t = 0; t < 9; ++t) { g.dy[t] = (int8_t)(cv.fwd.ky[t] - 1); g.dx[t] = (int8_t)(cv.fwd.kx[t] - 1); } }
            else { g.ntaps = 1; g.dy[0] = 0; g.dx[0] = 0; }
        }
        bind_conv(L.fwd, g, L.conv, false, 0, cv.cin, cv.cout, L.groups);
        if (stride == 1) {
            stcd_conv_geom gw = g; gw.ldo = L.dA.ld;
            bind_wgrad(L.wg[0], gw, L.conv, L.in.off, L.dA.off);
            L.nwg = 1;
        } else {
            // stride 2: input pixel (2m + ky - 1, 2n + kx - 1) lies in parity plane (py, px) at plane coordinates (m + dy, n + dx),
            // dy = -1 for ky == 0 and 0 otherwise.  A plane is a VIEW of the NHWC tensor: pixel stride 2*ld, and a virtual row
            // of Wi pixels (= two real rows), of which the first Wi/2 exist -- one stride-1 weight-gradient launch per plane.
            static const int start[4] = {0, 1, 3, 5}, cnt[4] = {1, 2, 2, 4};
            const int nph = L.kind == K_CONV3_S2 ? 4 : 1;
            for (int ph = 0; ph < nph; ++ph) {
                const int py = ph >> 1, px = ph & 1;
                stcd_conv_geom gw;
                memset(&gw, 0, sizeof(gw));
                gw.n = L.N; gw.hi = L.Hi / 2; gw.wi = L.Wi; gw.ci = L.K; gw.ldi = 2 * L.in.ld;
                gw.hm = L.Ho; gw.wm = L.Wo; gw.in_stride = 1; gw.ho = L.Ho; gw.wo = L.Wo; gw.out_stride = 1;
                gw.co = cv.cout; gw.ldo = L.dA.ld;
                gw.ntaps = L.kind == K_CONV3_S2 ? cnt[ph] : 1;
                for (int t = 0; t < gw.ntaps; ++t) {
                    const int ky = L.kind == K_CONV3_S2 ? cv.fwd.ky[start[ph] + t] : 1, kx = L.kind == K_CONV3_S2 ? cv.fwd.kx[start[ph] + t] : 1;
                    gw.dy[t] = (int8_t)(ky == 0 ? -1 : 0); gw.dx[t] = (int8_t)(kx == 0 ? -1 : 0);
                }
                bind_wgrad(L.wg[ph], gw, L.conv, L.in.off + ((int64_t)py * L.Wi + px) * L.in.ld * T, L.dA.off, start[ph], L.Wi / 2);
            }
            L.nwg = nph;
        }
        if (!L.has_dIn) continue;
        if (stride == 1) {
            stcd_conv_geom gd = L.kind == K_CONV3 ? geom3(L.N, L.Ho, L.Wo, cv.dgrad.kpad, L.dA.ld, cv.cin, L.dIn.ld)
                                                  : geom1(L.N, L.Ho, L.Wo, cv.dgrad.kpad, L.dA.ld, cv.cin, L.dIn.ld);
            bind_conv(L.dgr[0], gd, L.conv, true, 0, cv.cout, cv.cin);
            L.ndgr = 1;
        } else if (L.kind == K_CONV3_S2) {      // d(in)(2m+py, 2n+px) = sum_{dy<=py, dx<=px} dY(m+dy, n+dx) W[.][.][py+1-2dy][px+1-2dx]
            static const int start[4] = {0, 1, 3, 5};
            for (int ph = 0; ph < 4; ++ph) {
                const int py = ph >> 1, px = ph & 1;
                stcd_conv_geom gd;
                memset(&gd, 0, sizeof(gd));
                gd.n = L.N; gd.hi = L.Ho; gd.wi = L.Wo; gd.ci = cv.dgrad.kpad; gd.ldi = L.dA.ld;
                gd.hm = L.Ho; gd.wm = L.Wo; gd.in_stride = 1; gd.ho = L.Hi; gd.wo = L.Wi; gd.out_stride = 2; gd.oy0 = py; gd.ox0 = px;
                gd.co = cv.cin; gd.ldo = L.dIn.ld;
                int t = 0;
                for (int dy = 0; dy <= py; ++dy) for (int dx = 0; dx <= px; ++dx) { gd.dy[t] = (int8_t)dy; gd.dx[t] = (int8_t)dx; ++t; }
                gd.ntaps = t;
                bind_conv(L.dgr[ph], gd, L.conv, true, start[ph], cv.cout, cv.cin);
            }
            L.ndgr = 4;
        } else {                                // 1x1 stride 2: only the even positions receive a gradient (the rest is zeroed)
            stcd_conv_geom gd;
            memset(&gd, 0, sizeof(gd));
            gd.n = L.N; gd.hi = L.Ho; gd.wi = L.Wo; gd.ci = cv.dgrad.kpad; gd.ldi = L.dA.ld;
            gd.hm = L.Ho; gd.wm = L.Wo; gd.in_stride = 1; gd.ho = L.Hi; gd.wo = L.Wi; gd.out_stride = 2;
            gd.co = cv.cin; gd.ldo = L.dIn.ld; gd.ntaps = 1;
            bind_conv(L.dgr[0], gd, L.conv, true, 0, cv.cout, cv.cin);
            L.ndgr = 1;
        }
    }
    {
        const ConvW& cv = e.convs[e.g_head_conv];
        const int nhead = D == 2 ? 3 * B : B;
        bind_conv(e.g_head_fwd, geom3(nhead, H, W, 16, 16, e.label, e.label), e.g_head_conv, false, 0, 16, e.label);
        bind_wgrad(e.g_head_wg, geom3(nhead, H, W, 16, 16, e.label, 8), e.g_head_conv, e.gX3.off, e.G.off);
        bind_conv(e.g_head_dgr, geom3(nhead, H, W, cv.dgrad.kpad, 8, 16, 16), e.g_head_conv, true, 0, e.label, 16);
    }
    e.ws_tensors.clear();
    for (const auto& L : e.g_layers) {
        auto rec = [&](const char* sfx, const TRef& t, int n, int h, int w, int c) {
            if (t.off < 0) return;
            stcd_ws_tensor r;
            memset(&r, 0, sizeof(r));
            snprintf(r.name, sizeof(r.name), "%s.%s", L.name.c_str(), sfx);
            r.offset_bytes = t.off; r.n = n; r.h = h; r.w = w; r.c = c; r.ld = t.ld; r.dtype = e.dt;
            e.ws_tensors.push_back(r);
        };
        rec("in", L.in, L.N, L.Hi, L.Wi, L.K);
        rec("Y", L.Y, L.N, L.Ho, L.Wo, L.C);
        rec("A", L.A, L.N, L.Ho, L.Wo, L.C);
        rec("dY", L.dA, L.N, L.Ho, L.Wo, L.C);
        if (L.has_dIn) rec("dIn", L.dIn, L.N, L.Hi, L.Wi, L.K);
        rec("res", L.res, L.N, L.Ho, L.Wo, L.C);
    }
    e.slab = ws.take(e.slab_floats * 4);
    e.bias_jobs.clear();
    {
        BiasJob jb{}; jb.acc_off = e.final_bias_acc; jb.out_off = e.convs[e.g_head_conv].b_off; jb.C = 8; jb.valid = e.label; jb.scale = BN_BS;
        e.bias_jobs.push_back(jb);
        e.bias_jobs_off = ws.take((int64_t)e.bias_jobs.size() * sizeof(BiasJob) + 16);
    }
    build_pack_jobs(e, ws);
    e.jobs_uploaded_ws = nullptr;
    e.ws_bytes = ws.cur;
    return 0;
}

static SliceViews to_views(const Ctx& c, const std::vector<ViewRef>& v) {
    SliceViews xs;
    xs.n = (int)v.size();
    for (int k = 0; k < xs.n; ++k) { xs.p[k] = c.at(v[k].off); xs.ld[k] = v[k].ld; xs.goff[k] = v[k].goff; xs.gmask[k] = v[k].gmask; }
    return xs;
}

static void glayer_forward(const Ctx& c, GLayer& L, float* bn_running, bool training) {
    stcd_engine& e = c.e;
    const ConvW& cv = e.convs[L.conv];
    const BnP& bn = e.bns[L.bn];
    const int64_t ppg = (int64_t)L.npg * L.Ho * L.Wo;
    float* stat = c.at<float>(L.stat);
    int fused = 0;
    if (L.kind == K_STEM7 && mfma_on(e) && L.fwd.gemm.ok && L.fwd.wf >= 0) {
        ProfScope ps(c, PC_CONV, 2.0 * L.N * L.Ho * L.Wo * 49.0 * cv.cin * cv.cout, 0.0, "k_conv_gemm<stem>");
        long long* sp = training ? c.at<long long>(L.facc) : nullptr;
        if (launch_conv_gemm(L.fwd.g, L.fwd.plan, L.fwd.gemm, c.at(L.in.off), c.at(L.fwd.wf), nullptr, c.at(L.Y.off), L.groups, sp, L.C, c.s) == 0)
            fused = sp ? 1 : 0;
    } else if (L.kind == K_STEM7) {
        ProfScope ps(c, PC_CONV, 2.0 * L.N * L.Ho * L.Wo * 49.0 * cv.cin * cv.cout, 0.0, "k_stem_fwd");
        launch_stem_fwd(e.dt, c.at(L.in.off), c.params + cv.w_off, c.at(L.Y.off), L.N, L.Hi, L.Wi, cv.cin, cv.cout, c.s);
    } else {
        StatReq sr; sr.acc = training ? c.at<long long>(L.facc) : nullptr; sr.groups = L.groups; sr.C = L.C;
        exec_conv(c, L.fwd, c.at(L.in.off), nullptr, c.at(L.Y.off), false, &sr, &fused);
    }
    BnActArgs a;
    if (training) {
        if (!fused) launch_bn_stats(e.dt, c.at(L.Y.off), L.Y.ld, L.C, L.groups, ppg, c.at<long long>(L.facc), c.s);
        a.facc = c.at<long long>(L.facc); a.gamma = c.params + bn.g_off; a.beta = c.params + bn.b_off;
        a.running_mean = bn_running + bn.run_off; a.running_var = bn_running + bn.run_off + L.C;
    } else {      // eval: the activation kernel reads the running statistics itself
        a.gamma = c.params + bn.g_off; a.beta = c.params + bn.b_off;
        a.running_mean = bn_running + bn.run_off; a.running_var = bn_running + bn.run_off + L.C;
    }
    a.Y = c.at(L.Y.off); a.ldy = L.Y.ld; a.A = c.at(L.A.off); a.lda = L.A.ld; a.a_group_off = ppg * L.A.ld;
    a.P = nullptr; a.ldp = 0; a.stat = stat; a.mask = nullptr;
    a.C = L.C; a.groups = L.groups; a.npg = L.npg; a.H = L.Ho; a.W = L.Wo; a.relu = L.relu ? 1 : 0;
    if (L.res.off >= 0) { a.res = c.at(L.res.off); a.ldres = L.res.ld; }
    a.extra = to_views(c, L.extra_dst);
    a.g_first = L.g_first;
    ProfScope ps(c, PC_BN_ACT, 0.0, 2.0 * L.N * L.Ho * L.Wo * L.C * (double)dsize(e.dt));
    launch_bn_act(e.dt, a, c.s);
}

static void glayer_backward(const Ctx& c, GLayer& L) {
    stcd_engine& e = c.e;
    const ConvW& cv = e.convs[L.conv];
    const BnP& bn = e.bns[L.bn];
    const int64_t HW = (int64_t)L.Ho * L.Wo, ppg = (int64_t)L.npg * HW;
    const float* stat = c.at<float>(L.stat);
    const void* res = L.res.off >= 0 ? c.at(L.res.off) : nullptr;
    const double act_bytes = (double)L.N * HW * L.C * (double)dsize(e.dt);
    {
        ProfScope ps(c, PC_BN_BWD_REDUCE, 0.0, 2.0 * act_bytes);
        const SliceViews xs = to_views(c, L.grad_src);
        launch_bn_bwd_reduce(e.dt, c.at(L.dA.off), L.dA.ld, ppg * L.dA.ld, c.at(L.Y.off), L.Y.ld, stat, nullptr, L.C, L.groups, L.npg, HW,
                             L.relu ? 1 : 0, c.at<long long>(L.bacc), c.s, res, L.res.ld, &xs, L.grad_base ? 1 : 0, c.at(L.dA.off));
    }
    {
        ProfScope ps(c, PC_BN_BWD_APPLY, 0.0, 3.0 * act_bytes);
        launch_bn_bwd_apply(e.dt, c.at(L.dA.off), L.dA.ld, ppg * L.dA.ld, c.at(L.dA.off), L.dA.ld, c.at(L.Y.off), L.Y.ld, stat,
                            c.at<long long>(L.bacc), c.grads + bn.g_off, c.grads + bn.b_off, nullptr, L.C, L.groups, L.npg, HW, L.relu ? 1 : 0,
                            c.s, res, L.res.ld, L.dRes.off >= 0 ? c.at(L.dRes.off) : nullptr, L.dRes.ld);
    }
    if (L.kind == K_STEM7 && L.nwg > 0) {
        for (int k = 0; k < L.nwg; ++k) exec_wgrad(c, L.wg[k], c.at(L.wg[k].in_off), c.at(L.dA.off));
        return;
    }
    if (L.kind == K_STEM7) {
        ProfScope ps(c, PC_WGRAD, 2.0 * L.N * HW * 49.0 * cv.cin * cv.cout, 0.0, "k_stem_wgrad");
        launch_stem_wgrad(e.dt, c.at(L.in.off), c.at(L.dA.off), c.grads + cv.w_off, L.N, L.Hi, L.Wi, cv.cin, cv.cout, c.s, c.at<float>(e.g_stem_part));
        return;
    }
    for (int k = 0; k < L.nwg; ++k) exec_wgrad(c, L.wg[k], c.at(L.wg[k].in_off), c.at(L.dA.off));
    if (!L.has_dIn) return;
    if (L.kind == K_CONV1_S2)
        (void)hipMemsetAsync(c.at(L.dIn.off), 0, (size_t)L.N * L.Hi * L.Wi * L.dIn.ld * dsize(e.dt), c.s);
    if (L.ndgr == 4) {
        if (!exec_conv_x4(c, L.dgr, c.at(L.dA.off), nullptr, c.at(L.dIn.off)))
            for (int ph = 0; ph < 4; ++ph) exec_conv(c, L.dgr[ph], c.at(L.dA.off), nullptr, c.at(L.dIn.off), false);
    } else {
        exec_conv(c, L.dgr[0], c.at(L.dA.off), nullptr, c.at(L.dIn.off), false);
    }
}

static int forward_segcd(stcd_engine& e, const float* x1, const float* x2, const float* params, float* bn_running, int training,
                         float* logits, void* workspace, hipStream_t s) {
    Ctx c{e, (char*)workspace, params, nullptr, s};
    const int B = e.B, dt = e.dt;
    const int64_t T = (int64_t)dsize(dt), HW = (int64_t)e.H * e.W;
    if (pack_all_weights(c, training != 0)) return 1;
    if (training) STCD_HIP(hipMemsetAsync(c.at(e.zero_begin), 0, e.zero_end - e.zero_begin, s));
    launch_in_pack(dt, x1, x2, c.at(e.X0.off), B, e.in_ch, e.H, e.W, s, e.seg_dates);
    for (const GStep& st : e.g_fwd) {
        if (st.kind == GS_LAYER) glayer_forward(c, e.g_layers[st.layer], bn_running, training != 0);
        else if (st.kind == GS_MAXPOOL) launch_maxpool3(dt, c.at(st.src.off), st.src.ld, c.at(st.dst.off), st.dst.ld, st.N, st.h, st.w, st.C, s,
                                                        training ? c.at<unsigned char>(e.g_pool_idx) : nullptr);
        else if (st.kind == GS_ABSDIFF) {
            const int64_t hw = (int64_t)st.h * st.w;
            launch_fuse(dt, 0, c.at(st.src.off), st.src.ld, (int64_t)st.N * hw * st.src.ld,
                        c.at<char>(st.src.off) + (int64_t)2 * st.N * hw * st.src.ld * T, st.src.ld, st.N, hw, st.C, s);
        } else launch_upsample2(dt, c.at(st.src.off), st.src.ld, c.at(st.dst.off), st.dst.ld, st.N, st.h, st.w, st.C, s);
    }
    if (e.seg_dates == 1) {      // UnetSeg: masks = head(decoder output)
        exec_conv(c, e.g_head_fwd, c.at(e.gX3.off), params + e.convs[e.g_head_conv].b_off, logits, true);
        STCD_HIP(hipGetLastError());
        return 0;
    }
    // head: X3[2B:3B] = |d1 - d2| ; raw = conv(X3) = [m1; m2; diffea] ; logits = [m1; m2; min(diffea, |m1 - m2|)]
    // (FFCTLCD: X3[2B:3B] is the decoder's output on |f1 - f2|, already in place)
    if (!e.seg_ffc) launch_fuse(dt, 0, c.at(e.gX3.off), 16, (int64_t)B * HW * 16, c.at<char>(e.gX3.off) + (int64_t)2 * B * HW * 16 * T, 16, B, HW, 16, s);
    exec_conv(c, e.g_head_fwd, c.at(e.gX3.off), params + e.convs[e.g_head_conv].b_off, c.at(e.g_raw3), true);
    launch_segcd_combine(c.at<float>(e.g_raw3), logits, (int64_t)B * e.label * HW, s);
    STCD_HIP(hipGetLastError());
    return 0;
}

static int backward_segcd(stcd_engine& e, const float* grad_logits, const float* params, float* grads, void* workspace, int stage,
                          hipStream_t s) {
    if (stage == 1) return 0;
    Ctx c{e, (char*)workspace, params, grads, s};
    const int B = e.B, dt = e.dt;
    const int64_t T = (int64_t)dsize(dt), HW = (int64_t)e.H * e.W;
    STCD_HIP(hipMemsetAsync(grads, 0, e.param_floats * 4, s));
    if (!mfma_on(e)) STCD_HIP(hipMemsetAsync(c.at(e.dwe_begin), 0, e.dwe_end - e.dwe_begin, s));
    if (e.seg_dates == 1) {
        launch_gout_pack(dt, grad_logits, c.at(e.G.off), B, e.label, e.H, e.W, s, c.at<long long>(e.final_bias_acc));
        exec_wgrad(c, e.g_head_wg, c.at(e.gX3.off), c.at(e.G.off));
        exec_conv(c, e.g_head_dgr, c.at(e.G.off), nullptr, c.at(e.gdX3.off), false);
    } else {
        launch_segcd_combine_bwd(c.at<float>(e.g_raw3), grad_logits, c.at<float>(e.g_draw3), (int64_t)B * e.label * HW, s);
        launch_gout_pack(dt, c.at<float>(e.g_draw3), c.at(e.G.off), 3 * B, e.label, e.H, e.W, s, c.at<long long>(e.final_bias_acc));
        exec_wgrad(c, e.g_head_wg, c.at(e.gX3.off), c.at(e.G.off));
        exec_conv(c, e.g_head_dgr, c.at(e.G.off), nullptr, c.at(e.gdX3.off), false);
        // d(d1), d(d2) += -/+ sign(d1 - d2) * d|d1 - d2| : written to a contribution buffer the last decoder layer gathers
        if (!e.seg_ffc) launch_fuse_bwd(dt, 0, c.at(e.gX3.off), 16, (int64_t)B * HW * 16, c.at<char>(e.gdX3.off) + (int64_t)2 * B * HW * 16 * T, 16,
                        c.at(e.gFuseTmp.off), 16, (int64_t)B * HW * 16, B, HW, 16, s);
    }
    for (int k = (int)e.g_fwd.size() - 1; k >= 0; --k) {
        const GStep& st = e.g_fwd[k];
        if (st.kind == GS_LAYER) glayer_backward(c, e.g_layers[st.layer]);
        else if (st.kind == GS_UPSAMPLE)
            launch_upsample2_bwd(dt, c.at(st.ddst.off), st.ddst.ld, c.at(st.dsrc.off), st.dsrc.ld, st.N, st.h, st.w, st.C, s);
        else if (st.kind == GS_ABSDIFF) {
            const int64_t hw = (int64_t)st.h * st.w;
            launch_fuse_bwd(dt, 0, c.at(st.src.off), st.src.ld, (int64_t)st.N * hw * st.src.ld,
                            c.at<char>(st.dsrc.off) + (int64_t)2 * st.N * hw * st.dsrc.ld * T, st.dsrc.ld,
                            c.at(st.ddst.off), st.ddst.ld, (int64_t)st.N * hw * st.ddst.ld, st.N, hw, st.C, s);
        }
        else {      // max-pool: d(P0) = conv1 contribution (written in place) + identity-branch contribution of layer1.0 (its
                    // down-sample's data gradient for Bottleneck encoders, the gated residual gradient itself for BasicBlock ones)
            launch_slice(dt, c.at(st.ddst.off), st.ddst.ld, c.at(e.g_pool_idc.off), e.g_pool_idc.ld, (int64_t)st.N * (st.h / 2) * (st.w / 2), st.C, 1, s);
            launch_maxpool3_bwd(dt, c.at<unsigned char>(e.g_pool_idx), c.at(st.ddst.off), st.ddst.ld, c.at(st.dsrc.off), st.dsrc.ld, st.N, st.h, st.w, st.C, s);
        }
    }
    reduce_stage(c, 0);
    launch_bias_finish(c.at<BiasJob>(e.bias_jobs_off), (int)e.bias_jobs.size(), c.ws, c.grads, s);
    STCD_HIP(hipGetLastError());
    return 0;
}

#include "engine_cf.inl"

}  // namespace stcd

// ================================================================================================ C ABI
extern "C" {

const char* stcd_last_error(void) { return stcd::g_err.c_str(); }
int stcd_abi_version(void) { return STCD_ABI_VERSION; }

static void engine_env_switches(stcd_engine* e) {
    const char* env = getenv("STCD_FORCE_REF_KERNELS");
    e->use_mfma = !(env && env[0] == '1');
    env = getenv("STCD_NO_SMALL_KERNEL");
    e->use_small = !(env && env[0] == '1');
    env = getenv("STCD_NO_GEMM_KERNEL");
    e->use_gemm = !(env && env[0] == '1');
    env = getenv("STCD_NO_RES_KERNEL");
    e->use_res = !(env && env[0] == '1');
    env = getenv("STCD_NO_WGRAD_GROUPS");
    e->use_wgroup = !(env && env[0] == '1');
    env = getenv("STCD_NO_ACT_FUSE");
    e->use_act_fuse = !(env && env[0] == '1');
    env = getenv("STCD_NO_SKIP_FUSED");
    e->use_skip_fused = !(env && env[0] == '1');
    env = getenv("STCD_NO_SKIP_RECOMPUTE");       // 1: the skip layers of diff / sub store their activations again
    e->use_skip_recompute = !(env && env[0] == '1');
    env = getenv("STCD_WGRAD_TAIL_SPLIT");        // 1: the first conv's weight gradient gets a grid of its own, so its stage-mates' grid
    e->wg_tail_split = env && env[0] == '1';      //    goes out ~120 us earlier (measured neutral: the step is HBM-bound)
    env = getenv("STCD_NO_BWDSUM_FUSE");          // 1: every layer runs its own k_bn_reduce (no sums in k_conv_small data gradients)
    e->use_bwdsum = !(env && env[0] == '1');
    env = getenv("STCD_BWDSUM_RES");              // 1: also the layers behind a k_conv_res data gradient (measured slower, see configure_fcsiam)
    e->use_bwdsum_res = env && env[0] == '1';
    env = getenv("STCD_VIRT_ACT");                // 1: virtual activations (opt-in); 0 / unset: k_bn_act per layer
    if (env) e->use_virt = atoi(env) != 0;
    env = getenv("STCD_XF_MODE");
    if (env) e->xf_mode = atoi(env);
    env = getenv("STCD_WGRAD_SIDE");              // 0: the decoder's weight gradients stay on the caller's stream
    if (env) e->wg_side_on = atoi(env) != 0;
    env = getenv("STCD_WGRAD_SIDE_DIV");          // share of the planner's block budget for stage 0's grouped grids: 1 / div
    if (env && atoi(env) > 0) e->wg_side_div = atoi(env);
    env = getenv("STCD_WGRAD_ROUNDS");
    if (env && atoi(env) > 0) e->wgroup_rounds = atoi(env);
    env = getenv("STCD_WGRAD_MIN_TILES");
    if (env && atoi(env) > 0) e->wgroup_min_tiles = atoi(env);
}

int stcd_cf_default_config(stcd_cf_config* cfg) {
    STCD_CHECK(cfg != nullptr, "cfg is null");
    static const int E[4] = {64, 128, 320, 512}, DP[4] = {3, 3, 4, 3}, HD[4] = {1, 2, 4, 8}, SR[4] = {8, 4, 2, 1};
    memset(cfg, 0, sizeof(*cfg));
    cfg->in_ch = 3; cfg->out_ch = 2;
    for (int i = 0; i < 4; ++i) { cfg->embed_dims[i] = E[i]; cfg->depths[i] = DP[i]; cfg->num_heads[i] = HD[i]; cfg->sr_ratios[i] = SR[i]; }
    cfg->mlp_ratio = 4; cfg->embedding_dim = 256; cfg->patch1 = 7; cfg->patch = 7;
    cfg->drop_rate = 0.1f; cfg->attn_drop = 0.1f; cfg->drop_path_rate = 0.1f; cfg->diff_drop = 0.6f;
    return 0;
}

int stcd_create_changeformer(const stcd_cf_config* cfg, int dtype, stcd_engine** out) {
    STCD_CHECK(out != nullptr && cfg != nullptr, "null argument");
    STCD_CHECK(cfg->in_ch >= 1 && cfg->in_ch <= 8, "in_ch must be in [1,8]");
    STCD_CHECK(cfg->out_ch >= 1 && cfg->out_ch <= 8, "out_ch must be in [1,8]");
    STCD_CHECK(dtype == STCD_DTYPE_F32 || dtype == STCD_DTYPE_BF16, "unknown dtype");
    STCD_CHECK(cfg->mlp_ratio >= 1 && cfg->embedding_dim >= 8 && cfg->embedding_dim % 8 == 0, "embedding_dim must be a positive multiple of 8");
    STCD_CHECK((cfg->embedding_dim & (cfg->embedding_dim - 1)) == 0, "embedding_dim must be a power of two (BatchNorm kernels)");
    STCD_CHECK((cfg->patch1 & 1) && (cfg->patch & 1) && cfg->patch1 >= 3 && cfg->patch >= 3, "patch sizes must be odd and >= 3");
    for (int i = 0; i < 4; ++i) {
        STCD_CHECK(cfg->embed_dims[i] >= 8 && cfg->embed_dims[i] % 8 == 0 && cfg->embed_dims[i] <= 1024, "embed_dims must be multiples of 8, <= 1024");
        STCD_CHECK(cfg->depths[i] >= 1 && cfg->num_heads[i] >= 1 && cfg->sr_ratios[i] >= 1, "depths, num_heads and sr_ratios must be >= 1");
        STCD_CHECK(cfg->embed_dims[i] % cfg->num_heads[i] == 0, "embed_dims must be divisible by num_heads");
    }
    for (float p : {cfg->drop_rate, cfg->attn_drop, cfg->drop_path_rate, cfg->diff_drop}) STCD_CHECK(p >= 0.f && p < 1.f, "drop rates must be in [0,1)");
    std::unique_ptr<stcd_engine> e(new stcd_engine());
    e->arch = STCD_ARCH_CHANGEFORMER; e->in_ch = cfg->in_ch; e->label = cfg->out_ch; e->dt = dtype;
    engine_env_switches(e.get());
    e->cf = std::make_shared<CfPlan>();
    CfPlan& P = *e->cf;
    for (int i = 0; i < 4; ++i) { P.E[i] = cfg->embed_dims[i]; P.depths[i] = cfg->depths[i]; P.heads[i] = cfg->num_heads[i]; P.srs[i] = cfg->sr_ratios[i]; }
    P.mlp_ratio = cfg->mlp_ratio; P.D = cfg->embedding_dim; P.patch1 = cfg->patch1; P.patch = cfg->patch;
    P.drop = cfg->drop_rate; P.attn_drop = cfg->attn_drop; P.drop_path = cfg->drop_path_rate; P.diff_drop = cfg->diff_drop;
    build_cf_tables(*e);
    *out = e.release();
    return 0;
}

int stcd_cf_num_sites(const stcd_engine* e) { return e && e->cf && e->configured ? (int)e->cf->sites.size() : 0; }
int stcd_cf_site_get(const stcd_engine* e, int i, stcd_cf_site* out) {
    STCD_CHECK(e && e->cf && e->configured && out && i >= 0 && i < (int)e->cf->sites.size(), "bad argument");
    memset(out, 0, sizeof(*out));
    const CfPlan::Site& s = e->cf->sites[i];
    snprintf(out->name, sizeof(out->name), "%s", s.name.c_str());
    out->ndim = s.nd;
    for (int k = 0; k < s.nd; ++k) out->dims[k] = s.dims[k];
    out->p = s.p;
    return 0;
}
uint32_t stcd_cf_site_seed(uint64_t seed, int site) { return cf_site_seed(seed, site); }
int stcd_cf_set_aux_backward(stcd_engine* e, int on) {
    STCD_CHECK(e && e->cf, "not a ChangeFormer engine");
    if (e->cf->aux_bwd != (on != 0)) e->configured = false;      // the heads' backward launches and scratch are part of the plan
    e->cf->aux_bwd = on != 0;
    return 0;
}
int stcd_set_wgrad_side(stcd_engine* e, int on) {
    STCD_CHECK(e, "null engine");
    if ((e->wg_side_on != 0) != (on != 0)) e->configured = false;        // the block budget of stage 0's grouped grids is part of the plan
    e->wg_side_on = on != 0;
    return 0;
}
int stcd_cf_set_drop_rates(stcd_engine* e, float drop_rate, float attn_drop, float diff_drop) {
    STCD_CHECK(e && e->cf, "not a ChangeFormer engine");
    for (float p : {drop_rate, attn_drop, diff_drop}) STCD_CHECK(p >= 0.f && p < 1.f, "drop rates must be in [0,1)");
    e->cf->drop = drop_rate; e->cf->attn_drop = attn_drop; e->cf->diff_drop = diff_drop;
    e->configured = false;      // the site table is rebuilt by the next stcd_configure
    return 0;
}
int64_t stcd_output_floats(const stcd_engine* e) {
    if (!e || !e->configured) return 0;
    if (e->cf) return e->cf->out_floats;
    const int maps = (is_segcd(e->arch) && !is_unetseg(e->arch)) ? 3 : 1;
    return (int64_t)maps * e->B * e->label * e->H * e->W;
}
int stcd_cf_output_info(const stcd_engine* e, int i, int64_t* offset, int* height, int* width) {
    STCD_CHECK(e && e->cf && e->configured && offset && height && width && i >= 0 && i < 5, "bad argument");
    if (i < 4) { *offset = e->cf->df[i].out_off; *height = e->cf->df[i].h; *width = e->cf->df[i].w; }
    else { *offset = e->cf->cp_off; *height = e->H; *width = e->W; }
    return 0;
}

int stcd_create(int arch, int in_ch, int label_ch, int dtype, stcd_engine** out) {
    STCD_CHECK(out != nullptr, "out is null");
    if (is_cf(arch)) {
        stcd_cf_config cfg;
        stcd_cf_default_config(&cfg);
        cfg.in_ch = in_ch; cfg.out_ch = label_ch;
        return stcd_create_changeformer(&cfg, dtype, out);
    }
    STCD_CHECK((arch >= STCD_ARCH_DIFF && arch <= STCD_ARCH_SEGCD_R152) || arch == STCD_ARCH_FCEF || arch == STCD_ARCH_XCONC || is_unetseg(arch) || is_ffctlcd(arch), "unknown arch");
    STCD_CHECK(arch != STCD_ARCH_FCEF || in_ch <= 4, "FC-EF concatenates the two dates along the channels: in_ch must be <= 4");
    STCD_CHECK(in_ch >= 1 && in_ch <= 8, "in_ch must be in [1,8]");
    STCD_CHECK(label_ch >= 1 && label_ch <= 8, "label_ch must be in [1,8]");
    STCD_CHECK(dtype == STCD_DTYPE_F32 || dtype == STCD_DTYPE_BF16, "unknown dtype");
    std::unique_ptr<stcd_engine> e(new stcd_engine());
    e->arch = arch; e->in_ch = in_ch; e->label = label_ch; e->dt = dtype;
    engine_env_switches(e.get());
    if (arch == STCD_ARCH_SNUNET) build_snunet_tables(*e);
    else if (is_segcd(arch)) build_segcd_tables(*e);
    else build_fcsiam_tables(*e);
    *out = e.release();
    return 0;
}
void stcd_destroy(stcd_engine* e) {
    if (e && e->wg_side) { (void)hipStreamSynchronize(e->wg_side); (void)hipStreamDestroy(e->wg_side); (void)hipEventDestroy(e->wg_fork); (void)hipEventDestroy(e->wg_join); }
    delete e;
}

int stcd_num_params(const stcd_engine* e) { return e ? (int)e->params.size() : 0; }
int stcd_param_info(const stcd_engine* e, int i, stcd_tensor_info* info) {
    STCD_CHECK(e && info && i >= 0 && i < (int)e->params.size(), "bad argument");
    *info = e->params[i];
    return 0;
}
int64_t stcd_param_floats(const stcd_engine* e) { return e ? e->param_floats : 0; }
int stcd_num_bn(const stcd_engine* e) { return e ? (int)e->bns.size() : 0; }
int stcd_bn_info_get(const stcd_engine* e, int i, stcd_bn_info* info) {
    STCD_CHECK(e && info && i >= 0 && i < (int)e->bns.size(), "bad argument");
    memset(info, 0, sizeof(*info));
    snprintf(info->name, sizeof(info->name), "%s", e->bns[i].name.c_str());
    info->channels = e->bns[i].C;
    info->calls_per_forward = e->bns[i].calls;
    info->offset = e->bns[i].run_off;
    return 0;
}
int64_t stcd_bn_floats(const stcd_engine* e) { return e ? e->bn_floats : 0; }

int stcd_configure(stcd_engine* e, int batch, int height, int width) {
    STCD_CHECK(e != nullptr, "engine is null");
    STCD_CHECK(batch >= 1, "batch must be >= 1");
    STCD_CHECK(height >= 16 && width >= 16, "height and width must be >= 16 (four 2x2 pools)");
    STCD_CHECK((int64_t)2 * batch * height * width * 16 < ((int64_t)1 << 31), "tensor too large for 32-bit pixel indexing");
    e->configured = false;
    e->packed_tag = 0; e->packed_ws = nullptr; e->packed_level = -1;
    e->B = batch; e->H = height; e->W = width;
    if (is_cf(e->arch)) {
        STCD_CHECK(height % 32 == 0 && width % 32 == 0, "ChangeFormer needs height and width divisible by 32 (stride-4 patch embedding, sr_ratio 8)");
        if (configure_cf(*e, batch, height, width)) return 1;
    } else if (e->arch == STCD_ARCH_SNUNET) {
        STCD_CHECK(height % 16 == 0 && width % 16 == 0, "SNUNet needs height and width divisible by 16 (the reference's cat of up-sampled maps fails otherwise)");
        if (configure_snunet(*e, batch, height, width)) return 1;
    } else if (is_segcd(e->arch)) {
        STCD_CHECK(height % 32 == 0 && width % 32 == 0, "SegCD needs height and width divisible by 32 (five stride-2 stages; the reference's cat of the x2 up-sampled maps fails otherwise)");
        STCD_CHECK((int64_t)3 * batch * height * width * 16 < ((int64_t)1 << 31), "tensor too large for 32-bit pixel indexing");
        if (configure_segcd(*e, batch, height, width)) return 1;
    } else if (configure_fcsiam(*e, batch, height, width)) return 1;
    // torch's BatchNorm2d refuses a training forward with one value per channel ("Expected more than 1 value per channel when
    // training"): remember the smallest per-channel sample count of the plan, stcd_forward checks it
    e->min_bn_count = INT64_MAX;
    for (const auto& L : e->enc) e->min_bn_count = std::min<int64_t>(e->min_bn_count, (int64_t)L.npg * L.H * L.W);
    for (const auto& L : e->dec) e->min_bn_count = std::min<int64_t>(e->min_bn_count, (int64_t)L.npg * L.H * L.W);
    for (const auto& b : e->sn_blocks) e->min_bn_count = std::min<int64_t>(e->min_bn_count, (int64_t)b.npg * b.H * b.W);
    for (const auto& L : e->g_layers) e->min_bn_count = std::min<int64_t>(e->min_bn_count, (int64_t)L.npg * L.Ho * L.Wo);
    if (e->cf) e->min_bn_count = std::min<int64_t>(e->min_bn_count, (int64_t)batch * (height / 32) * (width / 32));
    e->configured = true;
    e->fwd_training = false;
    return 0;
}
int64_t stcd_workspace_bytes(const stcd_engine* e) { return e && e->configured ? e->ws_bytes : 0; }
int stcd_num_dropout(const stcd_engine* e) { return e && e->configured ? (int)e->drops.size() : 0; }
int stcd_dropout_info_get(const stcd_engine* e, int i, stcd_dropout_info* info) {
    STCD_CHECK(e && e->configured && info && i >= 0 && i < (int)e->drops.size(), "bad argument");
    memset(info, 0, sizeof(*info));
    snprintf(info->name, sizeof(info->name), "%s", e->drops[i].name.c_str());
    info->rows = e->drops[i].rows;
    info->channels = e->drops[i].C;
    info->offset = e->drops[i].off;
    return 0;
}
int64_t stcd_dropout_floats(const stcd_engine* e) { return e && e->configured ? e->drop_floats : 0; }
int stcd_set_dropout_p(stcd_engine* e, float p) {
    STCD_CHECK(e != nullptr, "engine is null");
    STCD_CHECK(p >= 0.f && p < 1.f, "p must be in [0,1)");
    e->drop_p = p;
    return 0;
}

int stcd_forward(stcd_engine* e, const float* x1, const float* x2, const float* params, float* bn_running,
                 const float* dropout_masks, uint64_t dropout_seed, int training, float* logits, void* workspace,
                 void* hip_stream) {
    STCD_CHECK(e && e->configured, "engine not configured");
    STCD_CHECK(x1 && x2 && params && bn_running && logits && workspace, "null pointer argument");
    STCD_CHECK(!(training && e->min_bn_count <= 1), "Expected more than 1 value per channel when training: a BatchNorm layer of this network sees "
               "one value per channel at this batch / image size (torch.nn.BatchNorm2d raises the same)");
    e->fwd_training = false;
    int rc = is_cf(e->arch)
                 ? forward_cf(*e, x1, x2, params, bn_running, dropout_masks, dropout_seed, training, logits, workspace, (hipStream_t)hip_stream)
             : e->arch == STCD_ARCH_SNUNET
                 ? forward_snunet(*e, x1, x2, params, bn_running, training, logits, workspace, (hipStream_t)hip_stream)
                 : is_segcd(e->arch)
                       ? forward_segcd(*e, x1, x2, params, bn_running, training, logits, workspace, (hipStream_t)hip_stream)
                       : forward_fcsiam(*e, x1, x2, params, bn_running, dropout_masks, dropout_seed, training, logits, workspace,
                                        (hipStream_t)hip_stream);
    if (rc == 0) e->fwd_training = training != 0;
    return rc;
}

int stcd_backward(stcd_engine* e, const float* grad_logits, const float* params, float* grads, void* workspace, int stage,
                  void* hip_stream) {
    STCD_CHECK(e && e->configured, "engine not configured");
    STCD_CHECK(e->fwd_training, "backward requires a preceding training-mode forward on this engine");
    STCD_CHECK(grad_logits && params && grads && workspace, "null pointer argument");
    STCD_CHECK(stage >= -1 && stage <= 1, "stage must be -1, 0 or 1");
    if (is_cf(e->arch)) return backward_cf(*e, grad_logits, params, grads, workspace, stage, (hipStream_t)hip_stream);
    if (e->arch == STCD_ARCH_SNUNET) return backward_snunet(*e, grad_logits, params, grads, workspace, stage, (hipStream_t)hip_stream);
    if (is_segcd(e->arch)) return backward_segcd(*e, grad_logits, params, grads, workspace, stage, (hipStream_t)hip_stream);
    return backward_fcsiam(*e, grad_logits, params, grads, workspace, stage, (hipStream_t)hip_stream);
}

int stcd_set_weights_tag(stcd_engine* e, uint64_t tag) {
    STCD_CHECK(e != nullptr, "engine is null");
    e->weights_tag = tag;
    return 0;
}
int stcd_set_debug(stcd_engine* e, int flags) {
    STCD_CHECK(e != nullptr, "engine is null");
    e->debug_flags = flags;
    e->configured = false;      // takes effect at the next stcd_configure
    return 0;
}
int stcd_ws_tensor_count(const stcd_engine* e) { return e ? (int)e->ws_tensors.size() : 0; }
int stcd_ws_tensor_get(const stcd_engine* e, int index, stcd_ws_tensor* out) {
    STCD_CHECK(e != nullptr && out != nullptr, "null argument");
    STCD_CHECK(index >= 0 && index < (int)e->ws_tensors.size(), "tensor index out of range");
    *out = e->ws_tensors[index];
    return 0;
}

int stcd_grad_stage_range(const stcd_engine* e, int stage, int64_t* begin, int64_t* end) {
    STCD_CHECK(e && begin && end, "bad argument");
    STCD_CHECK(stage == 0 || stage == 1, "stage must be 0 or 1");
    if (stage == 0) { *begin = e->enc_param_end; *end = e->param_floats; }
    else { *begin = 0; *end = e->enc_param_end; }
    return 0;
}

int stcd_profile_enable(stcd_engine* e, int on) {
    STCD_CHECK(e != nullptr, "engine is null");
    e->prof.on = on != 0;
    e->prof.recs.clear();
    e->prof.names.clear();
    e->prof.used = 0;
    return 0;
}
int stcd_profile_read(stcd_engine* e, int klass, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    STCD_CHECK(e && total_ms && launches && flops && bytes, "bad argument");
    STCD_CHECK(klass >= 0 && klass < PC_COUNT, "unknown kernel class");
    *total_ms = 0.0; *launches = 0; *flops = 0.0; *bytes = 0.0;
    for (auto& r : e->prof.recs) {
        if (r.klass != klass) continue;
        STCD_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        STCD_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        *total_ms += ms; *launches += 1; *flops += r.flops; *bytes += r.bytes;
    }
    return 0;
}

int stcd_profile_num_kernels(const stcd_engine* e) { return e ? (int)e->prof.names.size() : 0; }
int stcd_profile_kernel(stcd_engine* e, int i, char* name, int name_cap, double* total_ms, int64_t* launches, double* flops,
                        double* bytes) {
    STCD_CHECK(e && name && total_ms && launches && flops && bytes && name_cap > 0, "bad argument");
    STCD_CHECK(i >= 0 && i < (int)e->prof.names.size(), "kernel index out of range");
    snprintf(name, (size_t)name_cap, "%s", e->prof.names[i].c_str());
    *total_ms = 0.0; *launches = 0; *flops = 0.0; *bytes = 0.0;
    for (auto& r : e->prof.recs) {
        if (r.name_id != i) continue;
        STCD_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        STCD_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        *total_ms += ms; *launches += 1; *flops += r.flops; *bytes += r.bytes;
    }
    return 0;
}

int64_t stcd_loss_scratch_bytes(void) { return loss_scratch_bytes(); }
int stcd_loss_ce(const float* logits, const int64_t* target, int batch, int classes, int64_t hw, int ignore_index,
                 float* loss_out, float* dlogits, void* scratch, void* hip_stream) {
    STCD_CHECK(logits && target && loss_out && scratch, "null pointer argument");
    STCD_CHECK(batch >= 1 && classes >= 2 && classes <= 16 && hw >= 1, "bad shape");
    launch_loss_ce(logits, target, batch, classes, hw, ignore_index, loss_out, dlogits, scratch, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_loss_bce_dice(const float* logits, const float* target, int64_t numel, int from_logits, float* loss_out,
                       float* dlogits, void* scratch, void* hip_stream) {
    STCD_CHECK(logits && target && loss_out && scratch, "null pointer argument");
    STCD_CHECK(numel >= 1, "bad shape");
    launch_loss_bce_dice(logits, target, numel, from_logits, loss_out, dlogits, scratch, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_loss_contrastive(const float* pred, const int64_t* cd_label, const int64_t* pse_label, int64_t numel_half, float* loss_out,
                          float* dpred, void* scratch, void* hip_stream) {
    STCD_CHECK(pred && cd_label && pse_label && loss_out && scratch, "null pointer argument");
    STCD_CHECK(numel_half >= 1, "bad shape");
    launch_loss_contrastive(pred, cd_label, pse_label, numel_half, loss_out, dpred, scratch, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_confusion_update(const float* logits, const int64_t* target, int batch, int classes, int64_t hw, int64_t* cm,
                          void* hip_stream) {
    STCD_CHECK(logits && target && cm, "null pointer argument");
    STCD_CHECK(batch >= 1 && (classes == 1 || classes == 2) && hw >= 1, "bad shape (classes must be 1 or 2)");
    launch_confusion(logits, target, batch, classes, hw, cm, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

int stcd_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t numel, int64_t step, double lr,
                   double beta1, double beta2, double eps, double weight_decay, int decoupled, void* hip_stream) {
    STCD_CHECK(params && grads && exp_avg && exp_avg_sq, "null pointer argument");
    STCD_CHECK(numel >= 1 && step >= 1, "numel and step must be >= 1");
    STCD_CHECK(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
               "buffers must be 16-byte aligned");
    // scalars are formed in double like torch's Python floats, then rounded once to fp32
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    launch_adam(params, grads, exp_avg, exp_avg_sq, numel, (float)lr, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                (float)eps, (float)weight_decay, decoupled ? 1 : 0, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)),
                (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

int stcd_adam_hyper(int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay, float* hyper_host8) {
    STCD_CHECK(hyper_host8 != nullptr && step >= 1, "bad argument");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hyper_host8[0] = (float)lr; hyper_host8[1] = (float)(1.0 - beta1); hyper_host8[2] = (float)beta2; hyper_host8[3] = (float)(1.0 - beta2);
    hyper_host8[4] = (float)eps; hyper_host8[5] = (float)weight_decay; hyper_host8[6] = (float)(lr / bc1); hyper_host8[7] = (float)(1.0 / sqrt(bc2));
    return 0;
}
int stcd_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t numel, const float* hyper_dev8,
                       int decoupled, void* hip_stream) {
    STCD_CHECK(params && grads && exp_avg && exp_avg_sq && hyper_dev8, "null pointer argument");
    STCD_CHECK(numel >= 1, "numel must be >= 1");
    launch_adam_dev(params, grads, exp_avg, exp_avg_sq, numel, hyper_dev8, decoupled ? 1 : 0, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

int stcd_pseudo_pair(const uint8_t* img_a, const uint8_t* donor, const uint8_t* mask, const uint8_t* change, const float* alpha,
                     const int32_t* erase_xywh, uint64_t seed, int batch, int height, int width, const float* mean3,
                     const float* std3, float* x1, float* x2, int64_t* c_label, int64_t* s_label_a, int64_t* s_label_b,
                     void* hip_stream) {
    STCD_CHECK(img_a && donor && mask && change && x1 && x2 && c_label && mean3 && std3, "null pointer argument");
    STCD_CHECK(batch >= 1 && height >= 1 && width >= 1, "bad shape");
    STCD_CHECK(std3[0] > 0.f && std3[1] > 0.f && std3[2] > 0.f, "std must be positive");
    launch_pseudo_pair(img_a, donor, mask, change, alpha, erase_xywh, seed, batch, height, width, mean3, std3, x1, x2, c_label,
                       s_label_a, s_label_b, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

int64_t stcd_augment_scratch_bytes(int n_images, int height, int width) {
    return (n_images >= 1 && height >= 1 && width >= 1) ? augment_scratch_bytes(n_images, height, width) : 0;
}
int stcd_augment(const float* x, const float* params, int n_images, int height, int width, const float* mean3, const float* std3,
                 float* out, void* scratch, int64_t scratch_bytes, void* hip_stream) {
    STCD_CHECK(x && params && mean3 && std3 && out && scratch, "null pointer argument");
    STCD_CHECK(n_images >= 1 && n_images <= 65535 && height >= 1 && width >= 1, "bad shape");
    STCD_CHECK(std3[0] > 0.f && std3[1] > 0.f && std3[2] > 0.f, "std must be positive");
    STCD_CHECK(scratch_bytes >= augment_scratch_bytes(n_images, height, width), "scratch too small");
    STCD_CHECK(x != out, "in-place augmentation is not supported (the blur reads neighbours)");
    launch_augment(x, params, n_images, height, width, mean3, std3, out, scratch, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

static int check_geom(const stcd_conv_geom* g) {
    STCD_CHECK(g != nullptr, "geometry is null");
    STCD_CHECK(g->ntaps >= 1 && g->ntaps <= 9, "ntaps must be in [1,9]");
    STCD_CHECK(g->ci >= 8 && g->ci % 8 == 0 && g->ldi >= g->ci && g->ldi % 8 == 0, "ci/ldi must be multiples of 8");
    STCD_CHECK(g->co >= 1 && g->ldo >= g->co, "bad co/ldo");
    STCD_CHECK(g->in_stride >= 1 && g->out_stride >= 1 && g->n >= 1 && g->hm >= 1 && g->wm >= 1, "bad sizes");
    STCD_CHECK((g->hm - 1) * g->out_stride + g->oy0 < g->ho && (g->wm - 1) * g->out_stride + g->ox0 < g->wo,
               "output positions exceed the output buffer");
    return 0;
}
// split count of a stand-alone k_wgrad_dma launch (stcd_op_wgrad impl 7): one round of blocks over the 256 CUs, >= 8 K-tiles per block
static int op_dma_splits(const stcd_conv_geom& g, const WgradMfmaPlan& p) {
    const int64_t M = (int64_t)g.n * g.hi * g.wi;
    return (int)std::max<int64_t>(1, std::min<int64_t>(256 / std::max(1, p.gy * p.gz), M / 512));
}
int64_t stcd_op_scratch_bytes(const stcd_conv_geom* g) {
    if (!g) return 0;
    ConvMfmaPlan p = conv_mfma_plan(*g);
    WgradMfmaPlan w = wgrad_mfma_plan(*g, g->ci, g->co), ww = wgrad_mfma_plan(*g, g->ci, g->co, true);
    WgradMfmaPlan wg = wgrad_gemm_plan(*g, g->ci, g->co);
    WgradMfmaPlan wd = wgrad_dma_plan(*g, g->ci, g->co);
    if (wd.ok) wgrad_dma_set_split(wd, *g, op_dma_splits(*g, wd), g->ci, g->co);
    return std::max<int64_t>(p.wf_elems * 2, std::max(std::max(std::max(w.slab_floats, ww.slab_floats), wg.ok ? wg.slab_floats : 0), wd.ok ? wd.slab_floats : 0) * 4) + 1024;
}
int stcd_op_conv(int dtype, int impl, const stcd_conv_geom* g, const void* in, const float* w, const float* bias,
                 void* out, void* scratch, int64_t scratch_bytes, void* hip_stream) {
    if (check_geom(g)) return 1;
    STCD_CHECK(in && w && out, "null pointer argument");
    if (impl == 1 || impl == 2 || impl == 3 || impl == 6) {
        STCD_CHECK(dtype == STCD_DTYPE_BF16, "the MFMA implementation is bf16 only");
        ConvMfmaPlan p = conv_mfma_plan(*g);
        STCD_CHECK(p.ok, "geometry not supported by the MFMA kernel");
        STCD_CHECK(scratch && scratch_bytes >= p.wf_elems * 2, "scratch too small for the fragment-order filter");
        launch_pack_frag(*g, p, w, g->ci, g->co, scratch, (hipStream_t)hip_stream);
        if (impl == 3) {      // resident-halo / streamed-filter kernel of the wide 3x3 layers (Ci % 64 == 0, Co % 128 == 0)
            const ConvHaloPlan hp = conv_halo_plan(*g, p);
            STCD_CHECK(hp.ok, "geometry not supported by the resident-halo kernel (3x3 stride 1, Ci % 64 == 0, Co % 128 == 0)");
            STCD_CHECK(launch_conv_halo(*g, p, hp, in, scratch, bias, out, (hipStream_t)hip_stream) == 0, "launch failed");
            STCD_HIP(hipGetLastError());
            return 0;
        }
        if (impl == 6) {      // 256 x 256 implicit-GEMM tile staged by LDS-DMA (Ci % 64 == 0, Co % 256 == 0, an even number >= 4 of K-tiles)
            const ConvDmaPlan dp = conv_dma_plan(*g, p);
            STCD_CHECK(dp.ok, "geometry not supported by the LDS-DMA kernel (Ci % 64 == 0, Co % 256 == 0, (Ci / 64) * taps even and >= 4)");
            STCD_CHECK(launch_conv_dma(*g, p, dp, in, scratch, bias, out, (hipStream_t)hip_stream) == 0, "launch failed");
            STCD_HIP(hipGetLastError());
            return 0;
        }
        if (impl == 1 && conv_small_ok(*g, p)) {
            STCD_CHECK(launch_conv_small(*g, in, scratch, bias, out, false, 1, nullptr, g->co, (hipStream_t)hip_stream) == 0,
                       "small-channel kernel rejected the geometry");
            STCD_HIP(hipGetLastError());
            return 0;
        }
        if (impl == 1 && !(getenv("STCD_NO_RES_KERNEL") && getenv("STCD_NO_RES_KERNEL")[0] == '1')) {
            const ConvResPlan rp = conv_res_plan(*g, p, 1);
            if (rp.ok && launch_conv_res(*g, p, rp, in, scratch, bias, out, 1, nullptr, g->co, (hipStream_t)hip_stream) == 0) {
                STCD_HIP(hipGetLastError());
                return 0;
            }
        }
        if (impl == 1 && !(getenv("STCD_NO_GEMM_KERNEL") && getenv("STCD_NO_GEMM_KERNEL")[0] == '1')) {
            const ConvGemmPlan gp = conv_gemm_plan(*g, p, 1);
            if (gp.ok && launch_conv_gemm(*g, p, gp, in, scratch, bias, out, 1, nullptr, g->co, (hipStream_t)hip_stream) == 0) {
                STCD_HIP(hipGetLastError());
                return 0;
            }
        }
        STCD_CHECK(launch_conv_mfma(*g, p, in, scratch, bias, out, false, (hipStream_t)hip_stream) == 0, "LDS budget exceeded");
        STCD_HIP(hipGetLastError());
        return 0;
    }
    STCD_CHECK(impl == 0, "impl must be 0 (reference FMA), 1 (MFMA, auto-selected kernel), 2 (generic MFMA kernel), 3 (resident-halo kernel) or 6 (LDS-DMA kernel)");
    launch_conv_ref(dtype, *g, in, w, g->ci, g->co, bias, out, false, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}
int stcd_op_wgrad(int dtype, int impl, const stcd_conv_geom* g, const void* in, const void* dout, float* dw, void* scratch,
                  int64_t scratch_bytes, void* hip_stream) {
    if (check_geom(g)) return 1;
    STCD_CHECK(in && dout && dw, "null pointer argument");
    if (impl == 1 || impl == 4 || impl == 5) {      // 4: the tile kernel with its 64 x 32-channel tile allowed (the SNUNet engine's choice); 5: 64 x 64
        STCD_CHECK(dtype == STCD_DTYPE_BF16, "the MFMA implementation is bf16 only");
        WgradMfmaPlan p = wgrad_mfma_plan(*g, g->ci, g->co, impl >= 4, impl == 5 ? 1 : 0);
        STCD_CHECK(p.ok, "geometry not supported by the MFMA kernel");
        STCD_CHECK(scratch && scratch_bytes >= p.slab_floats * 4, "scratch too small for the partial slabs");
        STCD_CHECK(launch_wgrad_mfma(*g, p, in, dout, (float*)scratch, g->ci, g->co, (hipStream_t)hip_stream) == 0, "LDS budget exceeded");
        launch_reduce_dw((const float*)scratch, p.gx, *g, g->ci, g->co, g->ci, g->co, nullptr, dw, (hipStream_t)hip_stream);
        STCD_HIP(hipGetLastError());
        return 0;
    }
    if (impl == 7) {      // the LDS-DMA kernel: 256 x 256 channel tile per (position split, tap) block (ChangeFormer's 256 -> 256 3x3 layers)
        STCD_CHECK(dtype == STCD_DTYPE_BF16, "the MFMA implementation is bf16 only");
        WgradMfmaPlan p = wgrad_dma_plan(*g, g->ci, g->co);
        STCD_CHECK(p.ok, "geometry not supported by the LDS-DMA weight-gradient kernel (stride 1, full map >= 64 wide, Ci % 256 == 0, Co % 256 == 0)");
        wgrad_dma_set_split(p, *g, op_dma_splits(*g, p), g->ci, g->co);
        const int64_t table = (p.slab_floats * 4 + 255) & ~(int64_t)255;
        STCD_CHECK(scratch && scratch_bytes >= table + (int64_t)sizeof(WgradJob), "scratch too small for the partial slabs + job table");
        WgradJob j = wgrad_dma_make_job(*g, p, (int64_t)(intptr_t)in, (int64_t)(intptr_t)dout, (int64_t)(intptr_t)scratch, g->ci, g->co);
        STCD_HIP(hipMemcpyAsync((char*)scratch + table, &j, sizeof(j), hipMemcpyHostToDevice, (hipStream_t)hip_stream));
        STCD_HIP(hipStreamSynchronize((hipStream_t)hip_stream));       // `j` is a stack object
        launch_wgrad_dma_group((const WgradJob*)((char*)scratch + table), 1, p.gx * p.gy * p.gz, nullptr, (hipStream_t)hip_stream);
        launch_reduce_dw((const float*)scratch, p.gx, *g, g->ci, g->co, g->ci, g->co, nullptr, dw, (hipStream_t)hip_stream);
        STCD_HIP(hipGetLastError());
        return 0;
    }
    if (impl == 3) {      // one-tap launches on the position-GEMM kernel (the engine's choice for Ci, Co >= 64)
        STCD_CHECK(dtype == STCD_DTYPE_BF16, "the MFMA implementation is bf16 only");
        WgradMfmaPlan p = wgrad_gemm_plan(*g, g->ci, g->co);
        STCD_CHECK(p.ok, "geometry not supported by the GEMM weight-gradient kernel (one tap, Ci >= 64, Co >= 64)");
        const int64_t table = (p.slab_floats * 4 + 255) & ~(int64_t)255;
        STCD_CHECK(scratch && scratch_bytes >= table + (int64_t)sizeof(WgradJob), "scratch too small for the partial slabs + job table");
        WgradJob j = wgrad_gemm_make_job(*g, p, (int64_t)(intptr_t)in, (int64_t)(intptr_t)dout, (int64_t)(intptr_t)scratch, g->ci, g->co);
        STCD_HIP(hipMemcpyAsync((char*)scratch + table, &j, sizeof(j), hipMemcpyHostToDevice, (hipStream_t)hip_stream));
        STCD_HIP(hipStreamSynchronize((hipStream_t)hip_stream));       // `j` is a stack object
        launch_wgrad_gemm_group(p.gemm, (const WgradJob*)((char*)scratch + table), 1, p.gx * p.gy * p.gz, nullptr, (hipStream_t)hip_stream);
        launch_reduce_dw((const float*)scratch, p.gx, *g, g->ci, g->co, g->ci, g->co, nullptr, dw, (hipStream_t)hip_stream);
        STCD_HIP(hipGetLastError());
        return 0;
    }
    STCD_CHECK(impl == 0, "impl must be 0 (reference FMA), 1 / 4 (MFMA tile kernel, 4: wide tile allowed), 3 (MFMA position-GEMM kernel) or 7 (LDS-DMA kernel)");
    STCD_HIP(hipMemsetAsync(dw, 0, (size_t)g->ntaps * g->ci * g->co * 4, (hipStream_t)hip_stream));
    launch_wgrad_ref(dtype, *g, in, dout, dw, g->ci, g->co, (hipStream_t)hip_stream);
    STCD_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
