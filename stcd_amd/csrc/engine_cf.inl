// engine_cf.inl -- the ChangeFormerV6 family (BASELINE.json configs[4]); included by engine.hip inside
// namespace stcd (it reuses the conv / weight-gradient / filter-packing machinery of that file).
//
// Reference: /root/reference/models/ChangeFormer.py  ChangeFormerV6 :1669-1701, EncoderTransformer_v3 :1342-1473 (OverlapPatchEmbed
// :195-236, Block :472-509, Attention :298-358, Mlp :260-295, DWConv :512-523), DecoderTransformer_v3 :1475-1631 (MLP :677-688,
// conv_diff :1138-1148, make_prediction :1151-1157); /root/reference/models/ChangeFormerBaseNetworks.py :85-120.
//
// Layout: tokens are NHWC pixels ([n, h*w, C] == [n, h, w, C]); the shared encoder runs both dates as ONE batch of 2B images
// (LayerNorm / attention / DropPath are per token / per image, so nothing couples the dates); the decoder works on B images.
// Every matrix product is a 1x1-geometry launch of the tap-list GEMM kernels: Linear layers directly, the strided patch-embedding
// and spatial-reduction convolutions over an im2col matrix in the reference's own K order (k_im2col), so filters and their
// gradients keep the reference layouts.  Every tensor has its own value and gradient buffer (no reuse): the weight gradients
// of a backward stage then run as the grouped launches of build_pack_jobs / reduce_stage.
// Dropout / DropPath masks are never stored: each site hashes (site seed, element index) in the forward and again in the backward.

struct CfT { TRef v, g; int n = 0, h = 0, w = 0, c = 0; };          // value + gradient, [n, h, w, c] NHWC (ld = c unless a slice)

struct CfGemm {                 // y[n, ho, wo, N] = conv-like(x) ; one ConvW, forward / data-gradient / weight-gradient launches
    int conv = -1;
    ConvOp fwd, dgr; WgradOp wg;
    bool has_dgr = false;
    TRef x, y, dy, dx;
};
struct CfLN { int64_t g_off = 0, b_off = 0; int C = 0; float eps = 1e-5f; int64_t stats = -1; int64_t M = 0; };
struct CfBN {                   // BatchNorm2d of the decoder (one group: the B change maps)
    int bn = -1, C = 0, n = 0, h = 0, w = 0;
    int64_t stat = -1, facc = -1, bacc = -1;
};
struct CfBlock {
    std::string name;
    int site0 = 0;              // attn_drop, proj_drop, drop_path1, mlp.drop1, mlp.drop2, drop_path2
    float dpr = 0.f;
    CfLN n1, n2, nsr;
    CfGemm q, kv, proj, fc1, fc2, sr;
    int64_t dw_w = 0, dw_b = 0;
    CfT x, xn, qt, pc, st, sn, kvt, ao, pr, x1, xn2, hd, u, a, f2, x2;
    int64_t lse = -1;
};
struct CfStage {
    int C = 0, heads = 0, sr = 1, k = 7, stride = 4, hin = 0, win = 0, h = 0, w = 0, hk = 0, wk = 0, cin = 0, Kp = 0;
    CfGemm pe;
    CfLN pe_norm, out_norm;
    CfT col, peo, tok, out;
    TRef in_v, in_g;            // the stage's input map (values; gradient target, off < 0 for the image)
    int in_ld = 0;
    std::vector<CfBlock> blocks;
};
struct CfDiff {                 // linear_c + conv_diff + make_prediction of one scale
    int s = 0, h = 0, w = 0, site0 = 0;
    CfGemm lin, ca, cb;
    int64_t alpha_a = 0, alpha_b = 0;
    CfBN bna, bnb;
    CfT lo, cat, ya, za, ba, aa, yb, zb, bb, c;       // c: the scale's change feature (after the up-sampled sum)
    // auxiliary head
    int aux_conv0 = -1, aux_bn = -1; int64_t aux_w3 = 0, aux_b3 = 0;
    ConvOp aux_fwd; int64_t aux_y = -1, aux_stat = -1; int64_t out_off = 0;
    // its backward (multi_scale_train): fp32 scratch of the small maps, the packed gradient of the first conv's output, that conv's
    // weight / data gradient launches, a temporary for the feature gradient it adds to c.g
    ConvOp aux_dgr; WgradOp aux_wg;
    int64_t aux_dz = -1, aux_dy = -1, aux_sums = -1, aux_G = -1, aux_dtmp = -1, aux_bias_acc = -1;
};
struct CfUp {                   // ConvTranspose2d(D, D, 4, stride 2, padding 1): 4 sub-pixel phases of 2x2 taps
    int conv[4] = {-1, -1, -1, -1};
    ConvOp fwd[4], dgr[2]; WgradOp wg[4];
    int64_t b_off = 0;
    CfT in, out; TRef dtmp;
};
struct CfRes {                  // ResidualBlock: conv2(relu(conv1 x)) * 0.1 + x
    CfGemm c1, c2;
    CfT x, r1, y2, out; TRef dtmp;
};
struct CfPlan {
    int E[4] = {64, 128, 320, 512}, depths[4] = {3, 3, 4, 3}, heads[4] = {1, 2, 4, 8}, srs[4] = {8, 4, 2, 1};
    int mlp_ratio = 4, D = 256, patch1 = 7, patch = 7;
    float drop = 0.1f, attn_drop = 0.1f, drop_path = 0.1f, diff_drop = 0.6f;
    CfStage st[4];
    CfDiff df[4];               // index 0: scale 4 ... index 3: scale 1 (forward order)
    bool aux_bwd = false;       // propagate the gradients of p_c4 ... p_c1 too (stcd_cf_set_aux_backward)
    CfGemm fuse; CfBN fuse_bn; CfT fcat, fy, fa;
    CfUp up2, up1; CfRes res2, res1;
    int head_conv = -1; ConvOp head_fwd, head_dgr; WgradOp head_wg;
    int64_t scratch = -1, scratch_floats = 0;
    std::vector<ColsumJob> cjobs[2];          // bias gradients per backward stage (grouped launch at the end of the stage)
    int64_t cjobs_off[2] = {-1, -1}, cpart_off = -1;
    int cj_blocks[2] = {0, 0}, cj_fin[2] = {0, 0};
    int nsites = 0;
    uint64_t seed = 0;
    int64_t out_floats = 0, cp_off = 0;
    struct Site { std::string name; int dims[4]; int nd; float p; };
    std::vector<Site> sites;
};

static bool is_cf(int arch) { return arch == STCD_ARCH_CHANGEFORMER; }

// a GEMM-shaped parameter: reference tensor `shape`, used as [N][K] (K = product of the trailing dims), K padded to kpad rows
static int cf_add_gemm(stcd_engine& e, const std::string& name, std::initializer_list<int64_t> shape, int K, int N, bool need_dgrad,
                       int kpad = 0, bool bias = true) {
    ConvW c;
    c.name = name; c.kind = K_CONV1; c.cin = K; c.cout = N;
    c.kin_p = kpad > 0 ? kpad : round8(K); c.nout_p = round8(N);
    add_param(e, name + ".weight", shape, &c.w_off);
    c.b_off = -1;
    if (bias) add_param(e, name + ".bias", {N}, &c.b_off);
    make_specs(c, need_dgrad);
    e.convs.push_back(c);
    return (int)e.convs.size() - 1;
}
static CfLN cf_add_ln(stcd_engine& e, const std::string& name, int C, float eps) {
    CfLN l; l.C = C; l.eps = eps;
    add_param(e, name + ".weight", {C}, &l.g_off);
    add_param(e, name + ".bias", {C}, &l.b_off);
    return l;
}

static void build_cf_tables(stcd_engine& e) {
    CfPlan& P = *e.cf;
    const std::string pre = "Tenc_x2.";
    int cin = e.in_ch;
    for (int s = 0; s < 4; ++s) {          // EncoderTransformer_v3.__init__ :1354-1361
        CfStage& S = P.st[s];
        S.C = P.E[s]; S.heads = P.heads[s]; S.sr = P.srs[s]; S.k = s == 0 ? P.patch1 : P.patch; S.stride = s == 0 ? 4 : 2; S.cin = cin;
        const int K = cin * S.k * S.k;
        S.Kp = ((K + 63) / 64) * 64;       // im2col row pitch: whole 64-channel chunks for the GEMM kernels
        const std::string n = pre + "patch_embed" + std::to_string(s + 1);
        S.pe.conv = cf_add_gemm(e, n + ".proj", {S.C, cin, S.k, S.k}, K, S.C, s > 0, S.Kp);
        S.pe_norm = cf_add_ln(e, n + ".norm", S.C, 1e-5f);
        cin = S.C;
    }
    int site = 0, j = 0, total_depth = 0;
    for (int s = 0; s < 4; ++s) total_depth += P.depths[s];
    for (int s = 0; s < 4; ++s) {          // block_k, norm_k per stage :1364-1401
        CfStage& S = P.st[s];
        const int C = S.C, Ch = P.mlp_ratio * C;
        S.blocks.resize(P.depths[s]);
        for (int i = 0; i < P.depths[s]; ++i, ++j) {
            CfBlock& b = S.blocks[i];
            b.name = pre + "block" + std::to_string(s + 1) + "." + std::to_string(i);
            b.site0 = site; site += 6;
            b.dpr = total_depth > 1 ? P.drop_path * (float)j / (float)(total_depth - 1) : 0.f;      // torch.linspace(0, rate, sum(depths))
            b.n1 = cf_add_ln(e, b.name + ".norm1", C, 1e-6f);
            b.q.conv = cf_add_gemm(e, b.name + ".attn.q", {C, C}, C, C, true);
            b.kv.conv = cf_add_gemm(e, b.name + ".attn.kv", {2 * C, C}, C, 2 * C, true);
            b.proj.conv = cf_add_gemm(e, b.name + ".attn.proj", {C, C}, C, C, true);
            if (S.sr > 1) {
                b.sr.conv = cf_add_gemm(e, b.name + ".attn.sr", {C, C, S.sr, S.sr}, C * S.sr * S.sr, C, true);
                b.nsr = cf_add_ln(e, b.name + ".attn.norm", C, 1e-5f);
            }
            b.n2 = cf_add_ln(e, b.name + ".norm2", C, 1e-6f);
            b.fc1.conv = cf_add_gemm(e, b.name + ".mlp.fc1", {Ch, C}, C, Ch, true);
            add_param(e, b.name + ".mlp.dwconv.dwconv.weight", {Ch, 1, 3, 3}, &b.dw_w);
            add_param(e, b.name + ".mlp.dwconv.dwconv.bias", {Ch}, &b.dw_b);
            b.fc2.conv = cf_add_gemm(e, b.name + ".mlp.fc2", {C, Ch}, Ch, C, true);
        }
        S.out_norm = cf_add_ln(e, pre + "norm" + std::to_string(s + 1), C, 1e-6f);
    }
    e.enc_param_end = e.param_floats;
    const std::string d = "TDec_x2.";
    const int D = P.D;
    for (int k = 0; k < 4; ++k) {          // DecoderTransformer_v3.__init__ :1498-1531, registration order c4, c3, c2, c1
        const int s = 4 - k;
        P.df[k].s = s;
        P.df[k].lin.conv = cf_add_gemm(e, d + "linear_c" + std::to_string(s) + ".proj", {D, P.E[s - 1]}, P.E[s - 1], D, true);
    }
    for (int k = 0; k < 4; ++k) {
        CfDiff& F = P.df[k];
        const std::string n = d + "diff_c" + std::to_string(F.s);
        F.ca.conv = add_conv(e, n + ".0", K_CONV3, 2 * D, D, true);
        add_param(e, n + ".1.weight", {1}, &F.alpha_a);
        F.bna.bn = add_bn(e, n + ".2", D, 1);
        F.cb.conv = add_conv(e, n + ".4", K_CONV3, D, D, true);
        add_param(e, n + ".5.weight", {1}, &F.alpha_b);
        F.bnb.bn = add_bn(e, n + ".6", D, 1);
        F.site0 = site; site += 2;
    }
    for (int k = 0; k < 4; ++k) {
        CfDiff& F = P.df[k];
        const std::string n = d + "make_pred_c" + std::to_string(F.s);
        F.aux_conv0 = add_conv(e, n + ".0", K_CONV3, D, e.label, true);
        F.aux_bn = add_bn(e, n + ".2", e.label, 1);
        add_param(e, n + ".3.weight", {e.label, e.label, 3, 3}, &F.aux_w3);
        add_param(e, n + ".3.bias", {e.label}, &F.aux_b3);
    }
    P.fuse.conv = add_conv(e, d + "linear_fuse.0", K_CONV1, 4 * D, D, true);
    P.fuse_bn.bn = add_bn(e, d + "linear_fuse.1", D, 1);
    auto add_up = [&](CfUp& U, const std::string& n) {
        // one reference tensor [Cin][Cout][4][4], four ConvW views of it: phase (py, px) owns the taps ky == py + 1 (mod 2), likewise kx;
        // the data gradient (a 16-tap stride-2 gather over d(out)) is split in two 8-tap launches kept in phases 0 and 1
        int64_t w_off = 0, b_off = 0;
        add_param(e, n + ".weight", {D, D, 4, 4}, &w_off);
        add_param(e, n + ".bias", {D}, &b_off);
        U.b_off = b_off;
        for (int ph = 0; ph < 4; ++ph) {
            const int py = ph >> 1, px = ph & 1;
            ConvW c;
            c.name = n + ".phase" + std::to_string(ph); c.kind = K_CONVT2; c.cin = D; c.cout = D; c.kin_p = round8(D); c.nout_p = round8(D);
            c.w_off = w_off; c.b_off = b_off;
            PackSpec& f = c.fwd;
            f = PackSpec{};
            f.ks = 4; f.kn_major = 1; f.K = D; f.N = D; f.kpad = c.kin_p; f.wld = c.nout_p; f.ntaps = 4;
            // out(2m + py) takes in(m + dy) through ky = py + 1 - 2 dy: py = 0: (dy, ky) = (0, 1), (-1, 3); py = 1: (1, 0), (0, 2)
            int t = 0;
            for (int a = 0; a < 2; ++a)
                for (int b2 = 0; b2 < 2; ++b2) {
                    const int dy = py == 0 ? -a : 1 - a, dx = px == 0 ? -b2 : 1 - b2;
                    f.ky[t] = (int8_t)(py + 1 - 2 * dy); f.kx[t] = (int8_t)(px + 1 - 2 * dx); ++t;
                }
            PackSpec& g = c.dgrad;
            g = PackSpec{};
            g.ks = 4; g.kn_major = 0; g.K = D; g.N = D; g.kpad = c.nout_p; g.wld = round8(D); g.ntaps = 0;
            if (ph < 2) {       // d(in)(m, n) = sum_{ky, kx} d(out)(2m - 1 + ky, 2n - 1 + kx) W[ci][co][ky][kx]; launch ph: ky in {2 ph, 2 ph + 1}
                g.ntaps = 8;
                for (int q = 0; q < 8; ++q) { g.ky[q] = (int8_t)(2 * ph + q / 4); g.kx[q] = (int8_t)(q % 4); }
            }
            e.convs.push_back(c);
            U.conv[ph] = (int)e.convs.size() - 1;
        }
    };
    auto add_res = [&](CfRes& R, const std::string& n) {
        R.c1.conv = add_conv(e, n + ".conv1.conv2d", K_CONV3, D, D, true);
        R.c2.conv = add_conv(e, n + ".conv2.conv2d", K_CONV3, D, D, true);
    };
    add_up(P.up2, d + "convd2x.conv2d");
    add_res(P.res2, d + "dense_2x.0");
    add_up(P.up1, d + "convd1x.conv2d");
    add_res(P.res1, d + "dense_1x.0");
    P.head_conv = add_conv(e, d + "change_probability.conv2d", K_CONV3, D, e.label, true);
    P.nsites = site;
}

static int configure_cf(stcd_engine& e, int B, int H, int W) {
    CfPlan& P = *e.cf;
    const int64_t T = (int64_t)dsize(e.dt);
    const int D = P.D, N2 = 2 * B;
    e.drops.clear(); e.drop_floats = 0;
    e.conv_ops.clear(); e.wgrad_ops.clear(); e.slab_floats = 0;
    e.ws_tensors.clear();
    P.sites.clear();
    std::vector<CfDiff*> aux_pending;      // the auxiliary heads' backward launches are bound once the zero arena exists
    Bump ws;
    auto mk = [&](int n, int h, int w, int c, bool grad = true) {
        CfT t; t.n = n; t.h = h; t.w = w; t.c = c;
        t.v.off = ws.take((int64_t)n * h * w * c * T); t.v.ld = c;
        if (grad) { t.g.off = ws.take((int64_t)n * h * w * c * T); t.g.ld = c; }
        return t;
    };
    auto rec = [&](const std::string& name, const TRef& t, int n, int h, int w, int c) {
        if (t.off < 0) return;
        stcd_ws_tensor r;
        memset(&r, 0, sizeof(r));
        snprintf(r.name, sizeof(r.name), "%s", name.c_str());
        r.offset_bytes = t.off; r.n = n; r.h = h; r.w = w; r.c = c; r.ld = t.ld; r.dtype = e.dt;
        e.ws_tensors.push_back(r);
    };
    auto recT = [&](const std::string& name, const CfT& t) { rec(name, t.v, t.n, t.h, t.w, t.c); rec(name + ".grad", t.g, t.n, t.h, t.w, t.c); };
    int64_t scratch_floats = 4096;
    auto need = [&](int64_t f) { scratch_floats = std::max(scratch_floats, f); };
    int64_t cpart_floats[2] = {0, 0};
    P.cjobs[0].clear(); P.cjobs[1].clear(); P.cj_blocks[0] = P.cj_blocks[1] = 0; P.cj_fin[0] = P.cj_fin[1] = 0;
    auto add_colsum = [&](int stage, const TRef& dy, int64_t M, int C, int64_t out_off) {
        ColsumJob j{};
        j.x_off = dy.off; j.M = M; j.out_off = out_off; j.ld = dy.ld; j.C = C;
        j.nblocks = colsum_job_blocks(M); j.start_block = P.cj_blocks[stage]; j.fin_start = P.cj_fin[stage];
        j.part_off = cpart_floats[stage];
        P.cj_blocks[stage] += j.nblocks; P.cj_fin[stage] += (C + 7) / 8; cpart_floats[stage] += (int64_t)j.nblocks * C;
        P.cjobs[stage].push_back(j);
    };
    auto bind_conv = [&](ConvOp& op, const stcd_conv_geom& g, int conv, bool dgrad, int tap0, int kreal, int nreal, int groups = 1) {
        op.g = g; op.conv = conv; op.dgrad = dgrad; op.tap0 = tap0; op.kreal = kreal; op.nreal = nreal;
        op.plan = ConvMfmaPlan(); op.wf = -1; op.small = false; op.res = ConvResPlan(); op.res_groups = groups; op.gemm = ConvGemmPlan();
        if (e.dt == BF16) {
            op.plan = conv_mfma_plan(g);
            if (op.plan.ok) op.wf = ws.take(op.plan.wf_elems * 2);
            op.small = conv_small_ok(g, op.plan);
            if (e.use_res && !op.small) op.res = conv_res_plan(g, op.plan, groups);
            pick_gemm_or_res(e, op, g, groups);
        }
        e.conv_ops.push_back(&op);
    };
    auto bind_wgrad = [&](WgradOp& op, const stcd_conv_geom& g, int conv, int64_t in_off, int64_t dout_off, int stage, int tap0 = 0) {
        const ConvW& cv = e.convs[conv];
        op.g = g; op.conv = conv; op.tap0 = tap0; op.kreal = cv.cin; op.nreal = cv.cout;
        op.in_off = in_off; op.dout_off = dout_off; op.grouped = false; op.own_taps = false;
        op.plan = WgradMfmaPlan(); op.slab = -1; op.stage = stage;
        if (e.dt == BF16) op.plan = pick_wgrad_plan(e, g, cv.fwd.kpad, cv.fwd.wld);
        e.wgrad_ops.push_back(&op);
    };
    // a 1x1-geometry GEMM over an [n, h, w, K] map (x.ld may exceed K: a slice)
    auto bind_linear = [&](CfGemm& G, const TRef& x, const TRef& dx, int n, int h, int w, const TRef& y, const TRef& dy, int stage, bool bias_grad = true) {
        const ConvW& cv = e.convs[G.conv];
        G.x = x; G.y = y; G.dy = dy; G.dx = dx; G.has_dgr = dx.off >= 0 && cv.dgrad.ntaps > 0;
        if (bias_grad && cv.b_off >= 0) add_colsum(stage, dy, (int64_t)n * h * w, cv.cout, cv.b_off);
        bind_conv(G.fwd, geom1(n, h, w, cv.kin_p, x.ld, cv.cout, y.ld), G.conv, false, 0, cv.cin, cv.cout);
        bind_wgrad(G.wg, geom1(n, h, w, cv.kin_p, x.ld, cv.cout, dy.ld), G.conv, x.off, dy.off, stage);
        if (G.has_dgr) bind_conv(G.dgr, geom1(n, h, w, cv.dgrad.kpad, dy.ld, cv.cin, dx.ld), G.conv, true, 0, cv.cout, cv.cin);
    };
    auto bind_conv3 = [&](CfGemm& G, const TRef& x, const TRef& dx, int K, int n, int h, int w, const TRef& y, const TRef& dy, int stage) {
        const ConvW& cv = e.convs[G.conv];
        G.x = x; G.y = y; G.dy = dy; G.dx = dx; G.has_dgr = dx.off >= 0;
        if (cv.b_off >= 0) add_colsum(stage, dy, (int64_t)n * h * w, cv.cout, cv.b_off);
        bind_conv(G.fwd, geom3(n, h, w, K, x.ld, cv.cout, y.ld), G.conv, false, 0, cv.cin, cv.cout);
        bind_wgrad(G.wg, geom3(n, h, w, K, x.ld, cv.cout, dy.ld), G.conv, x.off, dy.off, stage);
        if (G.has_dgr) bind_conv(G.dgr, geom3(n, h, w, cv.dgrad.kpad, dy.ld, cv.cin, dx.ld), G.conv, true, 0, cv.cout, cv.cin);
    };
    auto add_site = [&](const std::string& name, std::initializer_list<int> dims, float p) {
        CfPlan::Site s; s.name = name; s.p = p; s.nd = 0;
        for (int d_ : dims) s.dims[s.nd++] = d_;
        P.sites.push_back(s);
    };

    e.X0 = TRef(); e.X0.off = ws.take((int64_t)N2 * H * W * 8 * T); e.X0.ld = 8;
    // ------------------------------------------------------------------------------------------ encoder
    int hin = H, win = W;
    TRef in_v = e.X0, in_g;
    int in_ld = 8;
    for (int s = 0; s < 4; ++s) {
        CfStage& S = P.st[s];
        S.hin = hin; S.win = win; S.in_v = in_v; S.in_g = in_g; S.in_ld = in_ld;
        const int pad = S.k / 2;
        S.h = (hin + 2 * pad - S.k) / S.stride + 1; S.w = (win + 2 * pad - S.k) / S.stride + 1;
        S.hk = S.sr > 1 ? S.h / S.sr : S.h; S.wk = S.sr > 1 ? S.w / S.sr : S.w;
        if (S.sr > 1 && (S.h % S.sr || S.w % S.sr)) { set_error("ChangeFormer: the token map of every stage must be divisible by its sr_ratio (H, W divisible by 32)"); return 1; }
        const int C = S.C, Ch = P.mlp_ratio * C, d = C / S.heads;
        if (C % S.heads || d % 8 || d > 128 || C % 8) { set_error("ChangeFormer: head dimension must be a multiple of 8, <= 128"); return 1; }
        const int64_t M = (int64_t)N2 * S.h * S.w, Mk = (int64_t)N2 * S.hk * S.wk;
        if ((int64_t)N2 * S.heads * S.h * S.w * (int64_t)S.hk * S.wk >= ((int64_t)1 << 32)) { set_error("ChangeFormer: attention map too large for 32-bit dropout indices"); return 1; }
        if ((int64_t)S.k * S.k * (S.cin + 2) * T > 64 * 1024) { set_error("ChangeFormer: patch too large for the im2col LDS tile"); return 1; }
        S.col = mk(N2, S.h, S.w, S.Kp, s > 0);
        S.peo = mk(N2, S.h, S.w, C);
        S.tok = mk(N2, S.h, S.w, C);
        bind_linear(S.pe, S.col.v, S.col.g, N2, S.h, S.w, S.peo.v, S.peo.g, 1);
        S.pe_norm.stats = ws.take(M * 8); S.pe_norm.M = M;
        need(layernorm_bwd_scratch_floats(M, C)); need(colsum_scratch_floats(M, std::max(Ch, 2 * C)));
        need(dwgelu_bwd_scratch_floats(N2, S.h, S.w, Ch));
        need(attn_bwd_scratch_floats(N2, S.h * S.w, S.hk * S.wk, S.heads, d));
        const std::string sn = "stage" + std::to_string(s + 1);
        recT(sn + ".col", S.col); recT(sn + ".peo", S.peo); recT(sn + ".tok", S.tok);
        CfT cur = S.tok;
        for (size_t i = 0; i < S.blocks.size(); ++i) {
            CfBlock& b = S.blocks[i];
            b.x = cur;
            b.xn = mk(N2, S.h, S.w, C); b.qt = mk(N2, S.h, S.w, C);
            b.n1.stats = ws.take(M * 8); b.n1.M = M;
            bind_linear(b.q, b.xn.v, b.xn.g, N2, S.h, S.w, b.qt.v, b.qt.g, 1);
            b.kvt = mk(N2, S.hk, S.wk, 2 * C);
            if (S.sr > 1) {
                if ((int64_t)S.sr * S.sr * (C + 2) * T > 64 * 1024) { set_error("ChangeFormer: sr patch too large for the im2col LDS tile"); return 1; }
                b.pc = mk(N2, S.hk, S.wk, C * S.sr * S.sr);
                b.st = mk(N2, S.hk, S.wk, C);
                b.sn = mk(N2, S.hk, S.wk, C);
                bind_linear(b.sr, b.pc.v, b.pc.g, N2, S.hk, S.wk, b.st.v, b.st.g, 1);
                b.nsr.stats = ws.take(Mk * 8); b.nsr.M = Mk;
                bind_linear(b.kv, b.sn.v, b.sn.g, N2, S.hk, S.wk, b.kvt.v, b.kvt.g, 1);
            } else {
                b.sn = mk(N2, S.h, S.w, C);       // only the gradient half is used: the second contribution to d(xn)
                bind_linear(b.kv, b.xn.v, b.sn.g, N2, S.h, S.w, b.kvt.v, b.kvt.g, 1);
            }
            b.ao = mk(N2, S.h, S.w, C);
            b.lse = ws.take((int64_t)N2 * S.heads * S.h * S.w * 4);
            b.pr = mk(N2, S.h, S.w, C);
            bind_linear(b.proj, b.ao.v, b.ao.g, N2, S.h, S.w, b.pr.v, b.pr.g, 1);
            b.x1 = mk(N2, S.h, S.w, C);
            b.xn2 = mk(N2, S.h, S.w, C);
            b.n2.stats = ws.take(M * 8); b.n2.M = M;
            b.hd = mk(N2, S.h, S.w, Ch); b.u = mk(N2, S.h, S.w, Ch, false); b.a = mk(N2, S.h, S.w, Ch);
            bind_linear(b.fc1, b.xn2.v, b.xn2.g, N2, S.h, S.w, b.hd.v, b.hd.g, 1);
            b.f2 = mk(N2, S.h, S.w, C);
            bind_linear(b.fc2, b.a.v, b.a.g, N2, S.h, S.w, b.f2.v, b.f2.g, 1);
            b.x2 = mk(N2, S.h, S.w, C);
            const int Nq = S.h * S.w, Nk = S.hk * S.wk;
            add_site(b.name + ".attn.attn_drop", {N2, S.heads, Nq, Nk}, P.attn_drop);
            add_site(b.name + ".attn.proj_drop", {N2, Nq, C}, P.drop);
            add_site(b.name + ".drop_path1", {N2}, b.dpr);
            add_site(b.name + ".mlp.drop1", {N2, Nq, Ch}, P.drop);
            add_site(b.name + ".mlp.drop2", {N2, Nq, C}, P.drop);
            add_site(b.name + ".drop_path2", {N2}, b.dpr);
            const std::string bn_ = sn + ".block" + std::to_string(i);
            recT(bn_ + ".x", b.x); recT(bn_ + ".xn", b.xn); recT(bn_ + ".q", b.qt); recT(bn_ + ".kv", b.kvt); recT(bn_ + ".ao", b.ao);
            recT(bn_ + ".pr", b.pr); recT(bn_ + ".x1", b.x1); recT(bn_ + ".xn2", b.xn2); recT(bn_ + ".hd", b.hd); recT(bn_ + ".u", b.u);
            recT(bn_ + ".a", b.a); recT(bn_ + ".f2", b.f2); recT(bn_ + ".x2", b.x2);
            if (S.sr > 1) { recT(bn_ + ".pc", b.pc); recT(bn_ + ".st", b.st); recT(bn_ + ".sn", b.sn); }
            cur = b.x2;
        }
        S.out = mk(N2, S.h, S.w, C);
        S.out_norm.stats = ws.take(M * 8); S.out_norm.M = M;
        recT(sn + ".out", S.out);
        in_v = S.out.v; in_g = S.out.g; in_ld = C; hin = S.h; win = S.w;
    }
    // ------------------------------------------------------------------------------------------ decoder
    const int h1 = P.st[0].h, w1 = P.st[0].w;
    if (2 * P.st[1].h != h1 || 2 * P.st[2].h != P.st[1].h || 2 * P.st[3].h != P.st[2].h || 4 * h1 != H || 4 * w1 != W ||
        2 * P.st[1].w != w1 || 2 * P.st[2].w != P.st[1].w || 2 * P.st[3].w != P.st[2].w) {
        set_error("ChangeFormer: height and width must be divisible by 32"); return 1;
    }
    P.fcat = mk(B, h1, w1, 4 * D);
    int64_t out_off = 0;
    for (int k = 0; k < 4; ++k) {
        CfDiff& F = P.df[k];
        const CfStage& S = P.st[F.s - 1];
        F.h = S.h; F.w = S.w;
        F.lo = mk(N2, F.h, F.w, D);
        bind_linear(F.lin, S.out.v, S.out.g, N2, F.h, F.w, F.lo.v, F.lo.g, 0);
        F.cat = mk(B, F.h, F.w, 2 * D);
        F.ya = mk(B, F.h, F.w, D); F.za = mk(B, F.h, F.w, D); F.ba = mk(B, F.h, F.w, D); F.aa = mk(B, F.h, F.w, D);
        bind_conv3(F.ca, F.cat.v, F.cat.g, 2 * D, B, F.h, F.w, F.ya.v, F.ya.g, 0);
        F.yb = mk(B, F.h, F.w, D); F.zb = mk(B, F.h, F.w, D); F.bb = mk(B, F.h, F.w, D);
        bind_conv3(F.cb, F.aa.v, F.aa.g, D, B, F.h, F.w, F.yb.v, F.yb.g, 0);
        if (F.s == 1) {       // c1 lives in the last channel slice of the fusion concat (cat((_c4_up, _c3_up, _c2_up, _c1)) :1609)
            F.c.n = B; F.c.h = F.h; F.c.w = F.w; F.c.c = D;
            F.c.v = P.fcat.v; F.c.v.off += (int64_t)3 * D * T; F.c.g = P.fcat.g; F.c.g.off += (int64_t)3 * D * T;
        } else F.c = mk(B, F.h, F.w, D);
        for (CfBN* bn : {&F.bna, &F.bnb}) { bn->C = D; bn->n = B; bn->h = F.h; bn->w = F.w; bn->stat = ws.take((int64_t)2 * 4 * D * 4); }
        add_site("TDec_x2.diff_c" + std::to_string(F.s) + ".3", {B, F.h, F.w, D}, P.diff_drop);
        add_site("TDec_x2.diff_c" + std::to_string(F.s) + ".7", {B, F.h, F.w, D}, P.diff_drop);
        // auxiliary head: conv(D -> label) straight to fp32 NCHW, then ReLU - BN - conv on the small maps
        F.aux_y = ws.take((int64_t)B * e.label * F.h * F.w * 4);
        F.aux_stat = ws.take(64 * 4);
        bind_conv(F.aux_fwd, geom3(B, F.h, F.w, D, F.c.v.ld, e.label, e.label), F.aux_conv0, false, 0, D, e.label);
        if (P.aux_bwd) {       // backward of the head: planned only with stcd_cf_set_aux_backward (multi_scale_train)
            const ConvW& cv0 = e.convs[F.aux_conv0];
            F.aux_dz = ws.take((int64_t)B * e.label * F.h * F.w * 4); F.aux_dy = ws.take((int64_t)B * e.label * F.h * F.w * 4);
            F.aux_sums = ws.take(64 * 4);
            F.aux_dtmp = ws.take((int64_t)B * F.h * F.w * D * T);
            aux_pending.push_back(&F);
            (void)cv0;
        }
        F.out_off = out_off; out_off += (int64_t)B * e.label * F.h * F.w;
        const std::string dn = "dec.c" + std::to_string(F.s);
        recT(dn + ".lo", F.lo); recT(dn + ".cat", F.cat); recT(dn + ".ya", F.ya); recT(dn + ".aa", F.aa); recT(dn + ".yb", F.yb); recT(dn + ".c", F.c);
        need(colsum_scratch_floats((int64_t)N2 * F.h * F.w, D));
    }
    P.cp_off = out_off; P.out_floats = out_off + (int64_t)B * e.label * H * W;
    P.fy = mk(B, h1, w1, D); P.fa = mk(B, h1, w1, D);
    bind_linear(P.fuse, P.fcat.v, P.fcat.g, B, h1, w1, P.fy.v, P.fy.g, 0, false);      // bias in front of a train-mode BatchNorm: zero gradient
    P.fuse_bn.C = D; P.fuse_bn.n = B; P.fuse_bn.h = h1; P.fuse_bn.w = w1; P.fuse_bn.stat = ws.take((int64_t)2 * 4 * D * 4);
    recT("dec.fcat", P.fcat); recT("dec.fy", P.fy); recT("dec.fa", P.fa);
    auto bind_up = [&](CfUp& U, const CfT& in, int h, int w) {
        U.in = in;
        U.out = mk(B, 2 * h, 2 * w, D);
        U.dtmp = TRef(); U.dtmp.off = ws.take((int64_t)B * h * w * D * T); U.dtmp.ld = D;
        add_colsum(0, U.out.g, (int64_t)B * 4 * h * w, D, U.b_off);
        for (int ph = 0; ph < 4; ++ph) {
            const int py = ph >> 1, px = ph & 1;
            const ConvW& cv = e.convs[U.conv[ph]];
            stcd_conv_geom g;
            memset(&g, 0, sizeof(g));
            g.n = B; g.hi = h; g.wi = w; g.ci = D; g.ldi = in.v.ld; g.hm = h; g.wm = w; g.in_stride = 1;
            g.ho = 2 * h; g.wo = 2 * w; g.out_stride = 2; g.oy0 = py; g.ox0 = px; g.co = D; g.ldo = D; g.ntaps = 4;
            for (int t = 0; t < 4; ++t) { g.dy[t] = (int8_t)((py + 1 - cv.fwd.ky[t]) / 2); g.dx[t] = (int8_t)((px + 1 - cv.fwd.kx[t]) / 2); }
            bind_conv(U.fwd[ph], g, U.conv[ph], false, 0, D, D);
            bind_wgrad(U.wg[ph], g, U.conv[ph], in.v.off, U.out.g.off, 0);
        }
        for (int k = 0; k < 2; ++k) {
            const ConvW& cv = e.convs[U.conv[k]];
            stcd_conv_geom g;
            memset(&g, 0, sizeof(g));
            g.n = B; g.hi = 2 * h; g.wi = 2 * w; g.ci = cv.dgrad.kpad; g.ldi = D; g.hm = h; g.wm = w; g.in_stride = 2;
            g.ho = h; g.wo = w; g.out_stride = 1; g.co = D; g.ldo = D; g.ntaps = 8;
            for (int t = 0; t < 8; ++t) { g.dy[t] = (int8_t)(cv.dgrad.ky[t] - 1); g.dx[t] = (int8_t)(cv.dgrad.kx[t] - 1); }
            bind_conv(U.dgr[k], g, U.conv[k], true, 0, D, D);
        }
    };
    auto bind_res = [&](CfRes& R, const CfT& x, int h, int w) {
        R.x = x;
        R.r1 = mk(B, h, w, D); R.y2 = mk(B, h, w, D); R.out = mk(B, h, w, D);
        R.dtmp = TRef(); R.dtmp.off = ws.take((int64_t)B * h * w * D * T); R.dtmp.ld = D;
        bind_conv3(R.c1, x.v, R.dtmp, D, B, h, w, R.r1.v, R.r1.g, 0);
        bind_conv3(R.c2, R.r1.v, R.r1.g, D, B, h, w, R.y2.v, R.y2.g, 0);
    };
    bind_up(P.up2, P.fa, h1, w1);
    bind_res(P.res2, P.up2.out, 2 * h1, 2 * w1);
    bind_up(P.up1, P.res2.out, 2 * h1, 2 * w1);
    bind_res(P.res1, P.up1.out, H, W);
    recT("dec.up2", P.up2.out); recT("dec.res2", P.res2.out); recT("dec.up1", P.up1.out); recT("dec.res1", P.res1.out);
    recT("dec.res1.r1", P.res1.r1); recT("dec.res2.r1", P.res2.r1);
    need(colsum_scratch_floats((int64_t)B * H * W, D)); need(bn_precise_scratch_floats((int64_t)B * h1 * w1, D));
    e.G = TRef(); e.G.off = ws.take((int64_t)B * H * W * 8 * T); e.G.ld = 8;
    {
        const ConvW& cv = e.convs[P.head_conv];
        bind_conv(P.head_fwd, geom3(B, H, W, D, D, e.label, e.label), P.head_conv, false, 0, D, e.label);
        bind_wgrad(P.head_wg, geom3(B, H, W, D, D, e.label, 8), P.head_conv, P.res1.out.v.off, e.G.off, 0);
        bind_conv(P.head_dgr, geom3(B, H, W, cv.dgrad.kpad, 8, D, D), P.head_conv, true, 0, e.label, D);
    }
    // ---- zero arena (BatchNorm accumulators, the head's bias accumulator): one memset per training forward
    e.zero_begin = ws.cur;
    for (int k = 0; k < 4; ++k)
        for (CfBN* bn : {&P.df[k].bna, &P.df[k].bnb}) { bn->facc = ws.take(bn_acc_bytes(2, D)); bn->bacc = ws.take(bn_acc_bytes(2, D)); }
    P.fuse_bn.facc = ws.take(bn_acc_bytes(2, D)); P.fuse_bn.bacc = ws.take(bn_acc_bytes(2, D));
    e.final_bias_acc = ws.take(bn_acc_bytes(1, 8));
    for (CfDiff* Fp : aux_pending) { Fp->aux_bias_acc = ws.take(bn_acc_bytes(1, 8)); Fp->aux_G = ws.take((int64_t)B * Fp->h * Fp->w * 8 * T); }
    e.zero_end = ws.cur;
    for (CfDiff* Fp : aux_pending) {       // (bound OUTSIDE the arena: bind_conv carves the launch's fragment-order filter image from `ws`)
        CfDiff& F = *Fp;
        const ConvW& cv0 = e.convs[F.aux_conv0];
        bind_wgrad(F.aux_wg, geom3(B, F.h, F.w, D, F.c.v.ld, e.label, 8), F.aux_conv0, F.c.v.off, F.aux_G, 0);
        bind_conv(F.aux_dgr, geom3(B, F.h, F.w, cv0.dgrad.kpad, 8, D, D), F.aux_conv0, true, 0, e.label, D);
    }
    e.scratch8 = ws.take(256);
    e.masks = ws.take(256);
    P.scratch_floats = scratch_floats; P.scratch = ws.take(scratch_floats * 4);
    P.cpart_off = ws.take(std::max(cpart_floats[0], cpart_floats[1]) * 4 + 16);
    for (int st = 0; st < 2; ++st) P.cjobs_off[st] = ws.take((int64_t)P.cjobs[st].size() * sizeof(ColsumJob) + 16);
    for (auto& c : e.convs) {
        c.wpk_fwd = ws.take((int64_t)c.fwd.ntaps * c.fwd.kpad * c.fwd.wld * 4);
        if (c.dgrad.ntaps) c.wpk_dgrad = ws.take((int64_t)c.dgrad.ntaps * c.dgrad.kpad * c.dgrad.wld * 4);
    }
    e.dwe_begin = ws.cur;
    for (auto& c : e.convs) {
        c.dwe_floats = (int64_t)c.fwd.ntaps * c.fwd.kpad * c.fwd.wld;
        c.dwe = (e.dt == BF16 && e.use_mfma) ? -1 : ws.take(c.dwe_floats * 8);      // fp64 accumulators of the reference weight-gradient path only
    }
    e.dwe_end = ws.cur;
    e.slab = ws.take(e.slab_floats * 4);
    e.bias_jobs.clear();
    {
        BiasJob jb{}; jb.acc_off = e.final_bias_acc; jb.out_off = e.convs[P.head_conv].b_off; jb.C = 8; jb.valid = e.label; jb.scale = BN_BS;
        e.bias_jobs.push_back(jb);
        for (int k = 0; k < 4 && P.aux_bwd; ++k) {       // the auxiliary heads' first-conv biases
            BiasJob ja{}; ja.acc_off = P.df[k].aux_bias_acc; ja.out_off = e.convs[P.df[k].aux_conv0].b_off; ja.C = 8; ja.valid = e.label; ja.scale = BN_BS;
            e.bias_jobs.push_back(ja);
        }
        e.bias_jobs_off = ws.take((int64_t)e.bias_jobs.size() * sizeof(BiasJob) + 16);
    }
    build_pack_jobs(e, ws);
    e.jobs_uploaded_ws = nullptr;
    e.ws_bytes = ws.cur;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------ execution
static void cf_gemm_fwd(const Ctx& c, const CfGemm& G, const StatReq* sr = nullptr, int* fused = nullptr) {
    const ConvW& cv = c.e.convs[G.conv];
    exec_conv(c, G.fwd, c.at(G.x.off), cv.b_off >= 0 ? c.params + cv.b_off : nullptr, c.at(G.y.off), false, sr, fused);
}
// weight gradient (grouped launch at the end of the stage), bias gradient, data gradient
// (the bias gradient -- the column sums of dy -- runs in the stage's grouped launch: cf_colsum_stage)
static void cf_gemm_bwd(const Ctx& c, const CfGemm& G) {
    exec_wgrad(c, G.wg, c.at(G.x.off), c.at(G.dy.off));
    if (G.has_dgr) exec_conv(c, G.dgr, c.at(G.dy.off), nullptr, c.at(G.dx.off), false);
}
static void cf_colsum_stage(const Ctx& c, int stage) {
    const CfPlan& P = *c.e.cf;
    ProfScope ps(c, PC_POOL_FUSE, 0.0, 0.0, "k_colsum_group");
    launch_colsum_group(c.e.dt, c.at<ColsumJob>(P.cjobs_off[stage]), (int)P.cjobs[stage].size(), P.cj_blocks[stage], P.cj_fin[stage], c.ws,
                        c.at<float>(P.cpart_off), c.grads, c.s);
}
static void cf_ln_fwd(const Ctx& c, const CfLN& L, const TRef& x, const TRef& y) {
    ProfScope ps(c, PC_BN_ACT, 0.0, 2.0 * L.M * L.C * (double)dsize(c.e.dt), "k_ln_fwd");
    launch_layernorm(c.e.dt, c.at(x.off), x.ld, c.at(y.off), y.ld, c.params + L.g_off, c.params + L.b_off, c.at<float>(L.stats), L.M, L.C,
                     L.eps, c.s);
}
// dx = [add] + LN-backward(dy [+ dy2])
static void cf_ln_bwd(const Ctx& c, const CfLN& L, const TRef& dy, const TRef* dy2, const TRef& x, const TRef* add, const TRef& dx) {
    ProfScope ps(c, PC_BN_BWD_APPLY, 0.0, 3.0 * L.M * L.C * (double)dsize(c.e.dt), "k_ln_bwd");
    launch_layernorm_bwd(c.e.dt, c.at(dy.off), dy.ld, dy2 ? c.at(dy2->off) : nullptr, dy2 ? dy2->ld : 0, c.at(x.off), x.ld,
                         c.at<float>(L.stats), c.params + L.g_off, add ? c.at(add->off) : nullptr, add ? add->ld : 0, c.at(dx.off), dx.ld,
                         c.grads + L.g_off, c.grads + L.b_off, c.at<float>(c.e.cf->scratch), L.M, L.C, c.s);
}
static DropSite cf_site(const stcd_engine& e, int site, float p, bool training) { return cf_make_site(e.cf->seed, site, p, training); }

static void cf_block_forward(const Ctx& c, const CfStage& S, const CfBlock& b, bool training) {
    stcd_engine& e = c.e;
    const CfPlan& P = *e.cf;
    const int dt = e.dt, N2 = 2 * e.B, C = S.C, Ch = P.mlp_ratio * C, d = C / S.heads;
    const int Nq = S.h * S.w, Nk = S.hk * S.wk;
    cf_ln_fwd(c, b.n1, b.x.v, b.xn.v);
    cf_gemm_fwd(c, b.q);
    if (S.sr > 1) {
        {
            ProfScope ps(c, PC_POOL_FUSE, 0.0, 0.0, "k_im2col");
            launch_im2col(dt, c.at(b.xn.v.off), C, c.at(b.pc.v.off), b.pc.v.ld, N2, S.h, S.w, C, S.sr, S.sr, 0, S.hk, S.wk, c.s);
        }
        cf_gemm_fwd(c, b.sr);
        cf_ln_fwd(c, b.nsr, b.st.v, b.sn.v);
    }
    cf_gemm_fwd(c, b.kv);
    {
        ProfScope ps(c, PC_CONV, 4.0 * N2 * (double)Nq * Nk * C, 0.0, "k_attn_fwd");
        launch_attn_fwd(dt, c.at(b.qt.v.off), C, c.at(b.kvt.v.off), 2 * C, c.at(b.ao.v.off), C, c.at<float>(b.lse), N2, Nq, Nk, S.heads, d,
                        1.f / sqrtf((float)d), cf_site(e, b.site0, P.attn_drop, training), c.s);
    }
    cf_gemm_fwd(c, b.proj);
    {
        ProfScope ps(c, PC_BN_ACT, 0.0, 0.0, "k_resid_drop");
        launch_resid_drop(dt, c.at(b.x.v.off), c.at(b.pr.v.off), c.at(b.x1.v.off), N2, Nq, C, cf_site(e, b.site0 + 1, P.drop, training),
                          cf_site(e, b.site0 + 2, b.dpr, training), c.s);
    }
    cf_ln_fwd(c, b.n2, b.x1.v, b.xn2.v);
    cf_gemm_fwd(c, b.fc1);
    {
        ProfScope ps(c, PC_BN_ACT, 0.0, 0.0, "k_dwgelu_fwd");
        launch_dwgelu_fwd(dt, c.at(b.hd.v.off), c.at(b.u.v.off), c.at(b.a.v.off), c.params + b.dw_w, c.params + b.dw_b, N2, S.h, S.w, Ch,
                          cf_site(e, b.site0 + 3, P.drop, training), c.s);
    }
    cf_gemm_fwd(c, b.fc2);
    {
        ProfScope ps(c, PC_BN_ACT, 0.0, 0.0, "k_resid_drop");
        launch_resid_drop(dt, c.at(b.x1.v.off), c.at(b.f2.v.off), c.at(b.x2.v.off), N2, Nq, C, cf_site(e, b.site0 + 4, P.drop, training),
                          cf_site(e, b.site0 + 5, b.dpr, training), c.s);
    }
}

// d(x2) arrives in b.x2.g; leaves d(x) in b.x.g
static void cf_block_backward(const Ctx& c, const CfStage& S, const CfBlock& b) {
    stcd_engine& e = c.e;
    const CfPlan& P = *e.cf;
    const int dt = e.dt, N2 = 2 * e.B, C = S.C, Ch = P.mlp_ratio * C, d = C / S.heads;
    const int Nq = S.h * S.w, Nk = S.hk * S.wk;
    launch_resid_drop_bwd(dt, c.at(b.x2.g.off), c.at(b.f2.g.off), N2, Nq, C, cf_site(e, b.site0 + 4, P.drop, true), cf_site(e, b.site0 + 5, b.dpr, true), c.s);
    cf_gemm_bwd(c, b.fc2);                                               // -> d(a)
    {
        ProfScope ps(c, PC_BN_BWD_APPLY, 0.0, 0.0, "k_dwgelu_bwd");
        launch_dwgelu_bwd(dt, c.at(b.hd.v.off), c.at(b.u.v.off), c.at(b.a.g.off), c.at(b.hd.g.off), c.params + b.dw_w, c.grads + b.dw_w,
                          c.grads + b.dw_b, c.at<float>(P.scratch), N2, S.h, S.w, Ch, cf_site(e, b.site0 + 3, P.drop, true), c.s);
    }
    cf_gemm_bwd(c, b.fc1);                                               // -> d(xn2)
    cf_ln_bwd(c, b.n2, b.xn2.g, nullptr, b.x1.v, &b.x2.g, b.x1.g);        // d(x1) = d(x2) + LN2^T
    launch_resid_drop_bwd(dt, c.at(b.x1.g.off), c.at(b.pr.g.off), N2, Nq, C, cf_site(e, b.site0 + 1, P.drop, true), cf_site(e, b.site0 + 2, b.dpr, true), c.s);
    cf_gemm_bwd(c, b.proj);                                              // -> d(ao)
    {
        ProfScope ps(c, PC_CONV, 10.0 * N2 * (double)Nq * Nk * C, 0.0, "k_attn_bwd");
        launch_attn_bwd(dt, c.at(b.qt.v.off), C, c.at(b.kvt.v.off), 2 * C, c.at(b.ao.v.off), C, c.at(b.ao.g.off), C, c.at<float>(b.lse),
                        c.at(b.qt.g.off), C, c.at(b.kvt.g.off), 2 * C, c.at<float>(P.scratch), N2, Nq, Nk, S.heads, d, 1.f / sqrtf((float)d),
                        cf_site(e, b.site0, P.attn_drop, true), c.s);
    }
    cf_gemm_bwd(c, b.q);                                                 // -> d(xn)
    cf_gemm_bwd(c, b.kv);                                                // -> d(sn) (sr > 1) or the second contribution to d(xn) (in b.sn.g)
    if (S.sr > 1) {
        cf_ln_bwd(c, b.nsr, b.sn.g, nullptr, b.st.v, nullptr, b.st.g);
        cf_gemm_bwd(c, b.sr);                                            // -> d(pc)
        ProfScope ps(c, PC_POOL_FUSE, 0.0, 0.0, "k_col2im");
        launch_col2im(dt, c.at(b.pc.g.off), b.pc.g.ld, c.at(b.xn.g.off), C, N2, S.h, S.w, C, S.sr, S.sr, 0, S.hk, S.wk, 1, c.s);
        cf_ln_bwd(c, b.n1, b.xn.g, nullptr, b.x.v, &b.x1.g, b.x.g);
    } else {
        cf_ln_bwd(c, b.n1, b.xn.g, &b.sn.g, b.x.v, &b.x1.g, b.x.g);
    }
}

static void cf_bn_forward(const Ctx& c, const CfBN& bn, const TRef& z, const TRef& out, float* bn_running, bool training) {
    stcd_engine& e = c.e;
    const BnP& p = e.bns[bn.bn];
    const int64_t ppg = (int64_t)bn.n * bn.h * bn.w;
    BnActArgs a;
    a.Y = c.at(z.off); a.ldy = z.ld; a.A = c.at(out.off); a.lda = out.ld; a.a_group_off = ppg * out.ld;
    a.P = nullptr; a.ldp = 0; a.stat = c.at<float>(bn.stat); a.mask = nullptr;
    a.C = bn.C; a.groups = 1; a.npg = bn.n; a.H = bn.h; a.W = bn.w; a.relu = 0;
    if (training) {      // batch statistics in double (the decoder's maps can hold a handful of values per channel); k_bn_act reads `stat`
        ProfScope ps(c, PC_BN_STATS, 0.0, (double)ppg * bn.C * (double)dsize(e.dt), "k_chan_moments");
        launch_bn_stats_precise(e.dt, c.at(z.off), z.ld, ppg, bn.C, c.params + p.g_off, c.params + p.b_off, bn_running + p.run_off,
                                bn_running + p.run_off + bn.C, c.at<float>(bn.stat), c.at<float>(e.cf->scratch), 0.1f, 1e-5f, c.s);
    } else {             // evaluation: the running statistics, read by the activation kernel itself
        a.gamma = c.params + p.g_off; a.beta = c.params + p.b_off;
        a.running_mean = bn_running + p.run_off; a.running_var = bn_running + p.run_off + bn.C;
    }
    ProfScope ps(c, PC_BN_ACT, 0.0, 2.0 * ppg * bn.C * (double)dsize(e.dt));
    launch_bn_act(e.dt, a, c.s);
}
// dz (in place over dout's buffer -> dz buffer): BatchNorm backward without activation
static void cf_bn_backward(const Ctx& c, const CfBN& bn, const TRef& dout, const TRef& z, const TRef& dz) {
    stcd_engine& e = c.e;
    const BnP& p = e.bns[bn.bn];
    const int64_t HW = (int64_t)bn.h * bn.w;
    const float* stat = c.at<float>(bn.stat);
    launch_bn_bwd_reduce(e.dt, c.at(dout.off), dout.ld, 0, c.at(z.off), z.ld, stat, nullptr, bn.C, 1, bn.n, HW, 0, c.at<long long>(bn.bacc), c.s);
    launch_bn_bwd_apply(e.dt, c.at(dout.off), dout.ld, 0, c.at(dz.off), dz.ld, c.at(z.off), z.ld, stat, c.at<long long>(bn.bacc),
                        c.grads + p.g_off, c.grads + p.b_off, nullptr, bn.C, 1, bn.n, HW, 0, c.s);
}

static void cf_up_forward(const Ctx& c, const CfUp& U) {
    for (int ph = 0; ph < 4; ++ph) exec_conv(c, U.fwd[ph], c.at(U.in.v.off), c.params + U.b_off, c.at(U.out.v.off), false);
}
static void cf_up_backward(const Ctx& c, const CfUp& U) {
    stcd_engine& e = c.e;
    const int D = U.out.c;
    const int64_t rows_in = (int64_t)U.in.n * U.in.h * U.in.w;
    for (int ph = 0; ph < 4; ++ph) exec_wgrad(c, U.wg[ph], c.at(U.in.v.off), c.at(U.out.g.off));
    exec_conv(c, U.dgr[0], c.at(U.out.g.off), nullptr, c.at(U.in.g.off), false);
    exec_conv(c, U.dgr[1], c.at(U.out.g.off), nullptr, c.at(U.dtmp.off), false);
    launch_axpby(e.dt, 1.f, c.at(U.in.g.off), U.in.g.ld, 1.f, c.at(U.dtmp.off), U.dtmp.ld, c.at(U.in.g.off), U.in.g.ld, rows_in, D, c.s);
}
// conv + fused epilogue where the kernel has one (k_conv_gemm), else conv then the same arithmetic element-wise
static void cf_conv_epi(const Ctx& c, const ConvOp& op, const void* in, const float* bias, void* out, int ldo, const ConvEpi& ep, int64_t rows, int C) {
    int fused = 0;
    exec_conv(c, op, in, bias, out, false, nullptr, nullptr, &ep, &fused);
    if (fused) return;
    const int dt = c.e.dt;
    if (ep.relu) launch_relu(dt, out, ldo, out, ldo, rows, C, c.s);
    if (ep.gate) launch_relu_bwd(dt, out, ldo, ep.gate, ep.ldg, out, ldo, rows, C, c.s);
    if (ep.res) launch_axpby(dt, ep.alpha, out, ldo, ep.beta, ep.res, ep.ldr, out, ldo, rows, C, c.s);
}
static void cf_res_forward(const Ctx& c, const CfRes& R) {
    stcd_engine& e = c.e;
    const int D = R.x.c;
    const int64_t rows = (int64_t)R.x.n * R.x.h * R.x.w;
    ConvEpi e1; e1.relu = 1;                                                         // r1 = relu(conv1(x))
    cf_conv_epi(c, R.c1.fwd, c.at(R.x.v.off), c.params + e.convs[R.c1.conv].b_off, c.at(R.r1.v.off), D, e1, rows, D);
    ConvEpi e2; e2.res = c.at(R.x.v.off); e2.ldr = R.x.v.ld; e2.alpha = 0.1f; e2.beta = 1.f;      // out = conv2(r1) * 0.1 + x
    cf_conv_epi(c, R.c2.fwd, c.at(R.r1.v.off), c.params + e.convs[R.c2.conv].b_off, c.at(R.out.v.off), D, e2, rows, D);
}
// d(out) in R.out.g -> d(x) in R.x.g
static void cf_res_backward(const Ctx& c, const CfRes& R) {
    stcd_engine& e = c.e;
    const int D = R.x.c;
    const int64_t rows = (int64_t)R.x.n * R.x.h * R.x.w;
    launch_axpby(e.dt, 0.1f, c.at(R.out.g.off), D, 0.f, nullptr, 0, c.at(R.y2.g.off), D, rows, D, c.s);      // d(y2): the weight gradient reads it
    exec_wgrad(c, R.c2.wg, c.at(R.c2.x.off), c.at(R.c2.dy.off));
    ConvEpi g1; g1.gate = c.at(R.r1.v.off); g1.ldg = D;                                                     // d(y1) = dgrad(conv2) * [r1 > 0]
    cf_conv_epi(c, R.c2.dgr, c.at(R.y2.g.off), nullptr, c.at(R.r1.g.off), D, g1, rows, D);
    exec_wgrad(c, R.c1.wg, c.at(R.c1.x.off), c.at(R.c1.dy.off));
    ConvEpi g2; g2.res = c.at(R.out.g.off); g2.ldr = D; g2.alpha = 1.f; g2.beta = 1.f;                      // d(x) = dgrad(conv1) + d(out)
    cf_conv_epi(c, R.c1.dgr, c.at(R.r1.g.off), nullptr, c.at(R.x.g.off), R.x.g.ld, g2, rows, D);
}

static int forward_cf(stcd_engine& e, const float* x1, const float* x2, const float* params, float* bn_running, const float* masks,
                      uint64_t seed, int training, float* logits, void* workspace, hipStream_t s) {
    STCD_CHECK(masks == nullptr, "ChangeFormer draws its dropout masks from the seed (counter hash); explicit masks are not supported");
    Ctx c{e, (char*)workspace, params, nullptr, s};
    CfPlan& P = *e.cf;
    const bool tr = training != 0;
    const int B = e.B, N2 = 2 * B, dt = e.dt, D = P.D;
    const int64_t T = (int64_t)dsize(dt);
    P.seed = seed;
    if (e.jobs_uploaded_ws != (const void*)c.ws)
        for (int st = 0; st < 2; ++st)
            if (!P.cjobs[st].empty())
                STCD_HIP(hipMemcpyAsync(c.at(P.cjobs_off[st]), P.cjobs[st].data(), P.cjobs[st].size() * sizeof(ColsumJob), hipMemcpyHostToDevice, s));
    if (pack_all_weights(c, tr)) return 1;
    if (tr) STCD_HIP(hipMemsetAsync(c.at(e.zero_begin), 0, e.zero_end - e.zero_begin, s));
    launch_in_pack(dt, x1, x2, c.at(e.X0.off), B, e.in_ch, e.H, e.W, s);
    for (int si = 0; si < 4; ++si) {
        const CfStage& S = P.st[si];
        {
            ProfScope ps(c, PC_POOL_FUSE, 0.0, 0.0, "k_im2col");
            launch_im2col(dt, c.at(S.in_v.off), S.in_ld, c.at(S.col.v.off), S.Kp, N2, S.hin, S.win, S.cin, S.k, S.stride, S.k / 2, S.h, S.w, s);
        }
        cf_gemm_fwd(c, S.pe);
        cf_ln_fwd(c, S.pe_norm, S.peo.v, S.tok.v);
        for (const CfBlock& b : S.blocks) cf_block_forward(c, S, b, tr);
        const CfT& last = S.blocks.empty() ? S.tok : S.blocks.back().x2;
        cf_ln_fwd(c, S.out_norm, last.v, S.out.v);
    }
    // ---- decoder (DecoderTransformer_v3.forward :1563-1631)
    const int h1 = P.st[0].h, w1 = P.st[0].w;
    for (int k = 0; k < 4; ++k) {
        const CfDiff& F = P.df[k];
        const int64_t rows = (int64_t)B * F.h * F.w;
        cf_gemm_fwd(c, F.lin);                                            // MLP on both dates
        launch_slice(dt, c.at(F.cat.v.off), 2 * D, c.at(F.lo.v.off), D, rows, D, 0, s);                                    // cat((_c_1, _c_2), dim=1)
        launch_slice(dt, c.at<char>(F.cat.v.off) + D * T, 2 * D, c.at<char>(F.lo.v.off) + rows * D * T, D, rows, D, 0, s);
        cf_gemm_fwd(c, F.ca);
        launch_prelu(dt, c.at(F.ya.v.off), D, c.at(F.za.v.off), D, params + F.alpha_a, rows, D, s);
        cf_bn_forward(c, F.bna, F.za.v, F.ba.v, bn_running, tr);
        launch_dropout_ew(dt, c.at(F.ba.v.off), D, c.at(F.aa.v.off), D, rows, D, cf_site(e, F.site0, P.diff_drop, tr), s);
        cf_gemm_fwd(c, F.cb);
        launch_prelu(dt, c.at(F.yb.v.off), D, c.at(F.zb.v.off), D, params + F.alpha_b, rows, D, s);
        cf_bn_forward(c, F.bnb, F.zb.v, F.bb.v, bn_running, tr);
        launch_dropout_ew(dt, c.at(F.bb.v.off), D, c.at(F.c.v.off), F.c.v.ld, rows, D, cf_site(e, F.site0 + 1, P.diff_drop, tr), s);
        if (k > 0) {       // + F.interpolate(_c(s+1), scale_factor=2, mode="bilinear")
            const CfDiff& Pv = P.df[k - 1];
            launch_bilinear(dt, c.at(Pv.c.v.off), Pv.c.v.ld, c.at(F.c.v.off), F.c.v.ld, B, Pv.h, Pv.w, F.h, F.w, D, 1, s);
        }
        // auxiliary prediction of the scale (make_prediction): outputs [p_c4, p_c3, p_c2, p_c1] in front of cp
        {
            const ConvW& c0 = e.convs[F.aux_conv0];
            const BnP& bp = e.bns[F.aux_bn];
            exec_conv(c, F.aux_fwd, c.at(F.c.v.off), params + c0.b_off, c.at(F.aux_y), true);
            launch_aux_head(c.at<float>(F.aux_y), logits + F.out_off, params + bp.g_off, params + bp.b_off, bn_running + bp.run_off,
                            bn_running + bp.run_off + bp.C, params + F.aux_w3, params + F.aux_b3, c.at<float>(F.aux_stat), B, e.label, F.h, F.w,
                            tr ? 1 : 0, s);
        }
        if (F.s > 1)       // resize(_c, size=c1 size, bilinear, align_corners=False) into its slice of the fusion concat
            launch_bilinear(dt, c.at(F.c.v.off), F.c.v.ld, c.at<char>(P.fcat.v.off) + (int64_t)k * D * T, 4 * D, B, F.h, F.w, h1, w1, D, 0, s);
    }
    cf_gemm_fwd(c, P.fuse);
    cf_bn_forward(c, P.fuse_bn, P.fy.v, P.fa.v, bn_running, tr);
    cf_up_forward(c, P.up2);
    cf_res_forward(c, P.res2);
    cf_up_forward(c, P.up1);
    cf_res_forward(c, P.res1);
    exec_conv(c, P.head_fwd, c.at(P.res1.out.v.off), params + e.convs[P.head_conv].b_off, logits + P.cp_off, true);
    STCD_HIP(hipGetLastError());
    return 0;
}

// grad_logits: gradients in the OUTPUT layout (p_c4, p_c3, p_c2, p_c1, cp back to back).  cp's is always propagated; the auxiliary
// heads' only with stcd_cf_set_aux_backward(e, 1) (the reference's default loss uses G_pred[-1] alone: trainer.py:311; with
// multi_scale_train == "True" it is a weighted sum over all five maps, :300-309)
static int backward_cf(stcd_engine& e, const float* grad_logits, const float* params, float* grads, void* workspace, int stage,
                       hipStream_t s) {
    Ctx c{e, (char*)workspace, params, grads, s};
    CfPlan& P = *e.cf;
    const int B = e.B, N2 = 2 * B, dt = e.dt, D = P.D;
    const int64_t T = (int64_t)dsize(dt);
    const int h1 = P.st[0].h, w1 = P.st[0].w;
    // both stages in one call: the grouped weight gradients go out with their last member on the engine's side stream (the head's
    // LDS-DMA group right after the second up-sampling layer's backward) and run beside the rest of the chain
    static const bool cf_side = [] { const char* v = getenv("STCD_CF_SIDE"); return !(v && v[0] == '0'); }();
    hipStream_t side = (stage < 0 && cf_side) ? wgrad_side_stream(e, s) : nullptr;
    EarlyScope early_scope(e, side != nullptr, side);
    for (int st = 0; st < 2; ++st)
        if (stage < 0 || stage == st)
            for (WgradGroup& G : e.wgroups[st]) { G.seen = 0; G.launched = false; }
    if (stage <= 0) {
        STCD_HIP(hipMemsetAsync(grads, 0, e.param_floats * 4, s));
        if (!mfma_on(e)) STCD_HIP(hipMemsetAsync(c.at(e.dwe_begin), 0, e.dwe_end - e.dwe_begin, s));
        launch_gout_pack(dt, grad_logits + P.cp_off, c.at(e.G.off), B, e.label, e.H, e.W, s, c.at<long long>(e.final_bias_acc));
        exec_wgrad(c, P.head_wg, c.at(P.res1.out.v.off), c.at(e.G.off));
        exec_conv(c, P.head_dgr, c.at(e.G.off), nullptr, c.at(P.res1.out.g.off), false);
        cf_res_backward(c, P.res1);
        cf_up_backward(c, P.up1);
        cf_res_backward(c, P.res2);
        cf_up_backward(c, P.up2);
        cf_bn_backward(c, P.fuse_bn, P.fa.g, P.fy.v, P.fy.g);
        cf_gemm_bwd(c, P.fuse);
        for (int k = 3; k >= 0; --k) {
            const CfDiff& F = P.df[k];
            const int64_t rows = (int64_t)B * F.h * F.w;
            // d(c_s): from the fusion concat (resize gradient; c1 sits in the concat itself), plus the x2 up-sampling path of scale s - 1
            if (F.s > 1) launch_bilinear_bwd(dt, c.at<char>(P.fcat.g.off) + (int64_t)k * D * T, 4 * D, c.at(F.c.g.off), F.c.g.ld, B, F.h, F.w, h1, w1, D, 0, s);
            if (k < 3) {
                const CfDiff& Nx = P.df[k + 1];
                launch_bilinear_bwd(dt, c.at(Nx.c.g.off), Nx.c.g.ld, c.at(F.c.g.off), F.c.g.ld, B, F.h, F.w, Nx.h, Nx.w, D, 1, s);
            }
            if (P.aux_bwd) {       // + the gradient arriving through the scale's auxiliary prediction head (trainer.py:300-309)
                const BnP& bp = e.bns[F.aux_bn];
                launch_aux_head_bwd(c.at<float>(F.aux_y), c.at<float>(F.aux_stat), grad_logits + F.out_off, params + F.aux_w3, c.at<float>(F.aux_dz),
                                    c.at<float>(F.aux_dy), c.at<float>(F.aux_sums), grads + F.aux_w3, grads + F.aux_b3, grads + bp.g_off,
                                    grads + bp.b_off, B, e.label, F.h, F.w, s);
                launch_gout_pack(dt, c.at<float>(F.aux_dy), c.at(F.aux_G), B, e.label, F.h, F.w, s, c.at<long long>(F.aux_bias_acc));
                exec_wgrad(c, F.aux_wg, c.at(F.c.v.off), c.at(F.aux_G));
                exec_conv(c, F.aux_dgr, c.at(F.aux_G), nullptr, c.at(F.aux_dtmp), false);
                launch_slice(dt, c.at(F.c.g.off), F.c.g.ld, c.at(F.aux_dtmp), D, rows, D, 1, s);
            }
            launch_dropout_ew(dt, c.at(F.c.g.off), F.c.g.ld, c.at(F.bb.g.off), D, rows, D, cf_site(e, F.site0 + 1, P.diff_drop, true), s);
            cf_bn_backward(c, F.bnb, F.bb.g, F.zb.v, F.zb.g);
            launch_prelu_bwd(dt, c.at(F.zb.g.off), D, c.at(F.yb.v.off), D, c.at(F.yb.g.off), D, params + F.alpha_b, grads + F.alpha_b, c.at<float>(P.scratch), rows, D, s);
            cf_gemm_bwd(c, F.cb);                  // (PReLU sits between the conv and the BatchNorm: the biases have gradients)
            launch_dropout_ew(dt, c.at(F.aa.g.off), D, c.at(F.ba.g.off), D, rows, D, cf_site(e, F.site0, P.diff_drop, true), s);
            cf_bn_backward(c, F.bna, F.ba.g, F.za.v, F.za.g);
            launch_prelu_bwd(dt, c.at(F.za.g.off), D, c.at(F.ya.v.off), D, c.at(F.ya.g.off), D, params + F.alpha_a, grads + F.alpha_a, c.at<float>(P.scratch), rows, D, s);
            cf_gemm_bwd(c, F.ca);
            launch_slice(dt, c.at(F.lo.g.off), D, c.at(F.cat.g.off), 2 * D, rows, D, 0, s);
            launch_slice(dt, c.at<char>(F.lo.g.off) + rows * D * T, D, c.at<char>(F.cat.g.off) + D * T, 2 * D, rows, D, 0, s);
            cf_gemm_bwd(c, F.lin);                 // -> d(stage output) (written; the next stage's patch embedding accumulates)
        }
        if (side) {
            STCD_HIP(hipEventRecord(e.wg_fork, s));
            STCD_HIP(hipStreamWaitEvent(side, e.wg_fork, 0));
            Ctx cs{e, (char*)workspace, params, grads, side};
            reduce_stage(cs, 0);
            cf_colsum_stage(cs, 0);
            launch_bias_finish(c.at<BiasJob>(e.bias_jobs_off), (int)e.bias_jobs.size(), c.ws, c.grads, side);
            STCD_HIP(hipEventRecord(e.wg_join, side));
        } else {
            reduce_stage(c, 0);
            cf_colsum_stage(c, 0);
            launch_bias_finish(c.at<BiasJob>(e.bias_jobs_off), (int)e.bias_jobs.size(), c.ws, c.grads, s);
        }
    }
    if (stage < 0 || stage == 1) {
        for (int si = 3; si >= 0; --si) {
            const CfStage& S = P.st[si];
            const CfT& last = S.blocks.empty() ? S.tok : S.blocks.back().x2;
            cf_ln_bwd(c, S.out_norm, S.out.g, nullptr, last.v, nullptr, last.g);
            for (int i = (int)S.blocks.size() - 1; i >= 0; --i) cf_block_backward(c, S, S.blocks[i]);
            cf_ln_bwd(c, S.pe_norm, S.tok.g, nullptr, S.peo.v, nullptr, S.peo.g);
            cf_gemm_bwd(c, S.pe);
            if (si > 0) {
                ProfScope ps(c, PC_POOL_FUSE, 0.0, 0.0, "k_col2im");
                launch_col2im(dt, c.at(S.col.g.off), S.Kp, c.at(S.in_g.off), S.in_ld, N2, S.hin, S.win, S.cin, S.k, S.stride, S.k / 2, S.h, S.w, 1, s);
            }
        }
        if (side) {
            STCD_HIP(hipEventRecord(e.wg_fork, s));
            STCD_HIP(hipStreamWaitEvent(side, e.wg_fork, 0));
            Ctx cs{e, (char*)workspace, params, grads, side};
            reduce_stage(cs, 1);
            cf_colsum_stage(cs, 1);
            STCD_HIP(hipEventRecord(e.wg_join, side));
            STCD_HIP(hipStreamWaitEvent(s, e.wg_join, 0));
        } else {
            reduce_stage(c, 1);
            cf_colsum_stage(c, 1);
        }
    }
    STCD_HIP(hipGetLastError());
    return 0;
}
