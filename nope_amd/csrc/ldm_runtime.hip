// Host-side runtime of the LDM cross-attention U-Net variant: `UNetModelPose`
// (src/model/u_net/ldm/adapt_openaimodel.py:14-158 over openaimodel.py:428-760 and attention.py:149-277) -- the variant whose
// pose conditioning is the cross-attention `context = pose_mlp(pose).unsqueeze(1)` north_star names.
//
// Op order follows UNetModelPose.forward exactly; the execution model is the one of unet_runtime.hip (NHWC activations = token
// major, every conv / linear on the implicit-GEMM MFMA kernels, all pose hypotheses of a reference latent as one batch, bump
// arena over caller-provided workspace).  What the MI355X schedule does differently from the module tree, same arithmetic:
//   * SpatialTransformer tokens are the NHWC activations themselves: the two `rearrange`s are no-ops;
//   * attn1's to_q / to_k / to_v are one GEMM against the row-concatenated weights;
//   * attn2 is cross-attention against ONE context token: softmax over a single key is exactly 1, so its output is
//     to_out(to_v(context)) for every query; to_q, to_k and norm2 never influence the result and are not evaluated (their
//     weights are still validated at create time).  Two linear maps in a row are one: W_out W_v is formed at create time, the
//     products of ALL transformer blocks are stacked, and a forward computes every block's per-sample row in ONE small
//     linear launch (context -> sum of the blocks' channels), each block adding its slice to its tokens;
//   * with injecting_condition_twice = false the timestep embedding is zeros, so emb_layers(emb) is its bias: folded into the
//     bias of in_layers' conv at create time (use_scale_shift_norm: that bias row is the FiLM (scale | shift) of every sample);
//   * FiLM (use_scale_shift_norm) costs no pass: out_norm(h) * (1 + scale) + shift is folded into the per-channel affine the
//     GroupNorm-apply kernel evaluates anyway (kernels_norm.hip);
//   * th.cat((h, skip)) is materialised (GroupNorm(32) groups of the following ResBlock can straddle the two halves).
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "nope_common.h"
#include "x2_range.h"

using namespace nope;

namespace {

struct LConv { void* w = nullptr; float* bias = nullptr; int Cin = 0, Cout = 0, ntaps = 1, mode = NOPE_CONV_PLAIN; void* w_x2 = nullptr; int x2_id = -1; };   // w_x2 / x2_id: NOPE_F16X2, the 3x3 convs' second pack and its slot in the range table (x2_range.h)
struct LNorm { float* gamma = nullptr; float* beta = nullptr; int C = 0; };
struct LRes { LNorm n1, n2; LConv c1, c2, skip; bool has_skip = false; float *emb_w = nullptr, *emb_b = nullptr; int Cin = 0, Cout = 0; };
struct LTB { LNorm ln1, ln3; LConv qkv, out1, ff1, ff2; int u_off = 0; };         // one BasicTransformerBlock; u_off: its slice of nope_ldm::u_w
struct LST { LNorm norm; LConv proj_in, proj_out; std::vector<LTB> blocks; int C = 0; };
struct LBlock { bool has_res = false, has_st = false, has_resample = false; LRes res; LST st; LConv resample; };

}  // namespace

struct nope_ldm {
    nope_ldm_config cfg;
    int dt = NOPE_F32;      // compute dtype (conv kernels, weight packing)
    int sdt = NOPE_F32;     // storage dtype of the activations (every other kernel)
    bool x2 = false;        // NOPE_F16X2: dt = NOPE_BF16X3 everywhere, plus a second weight pack per 3x3 conv (ResBlock convs, nearest-x2 up-sampling) for the
                            // ping-pong kernels' f16 + MX-fp8 tile; the 1x1 convs / linears of the transformer blocks stay three-pass
    mutable X2Range x2r;    // ... and the activation-range tracking that keeps the tile inside its accurate window (x2_range.h)
    std::vector<void*> allocs;
    LConv conv_in, conv_out;
    LNorm norm_out;
    std::vector<LBlock> input_blocks, output_blocks;     // input_blocks[0] is conv_in
    LRes mid1, mid2;
    LST mid_st;
    std::vector<int> skip_ch;                            // channels pushed by each input block
    float *pose_w0 = nullptr, *pose_b0 = nullptr, *pose_w2 = nullptr, *pose_b2 = nullptr;
    float *tw = nullptr, *tb = nullptr;                  // pose_mlp_timesteps (injecting_condition_twice)
    float *u_w = nullptr, *u_b = nullptr;                // stacked attn2 maps to_out.0 o to_v of every transformer block: [u_total][context_dim], [u_total]
    int u_total = 0;
    int emb_dim = 0;
};

namespace {

struct Loader {
    nope_ldm* net;
    hipStream_t s;
    std::map<std::string, const nope_tensor_desc*> tab;
    int err = NOPE_OK;
    std::string missing;
    struct UPart { float* comb; float* bias; int C, off; };
    std::vector<UPart> u_parts;
    int u_total = 0;
    void fail(const std::string& n) { if (err == NOPE_OK) { err = NOPE_ERR_WEIGHT; missing = n; } }
    // device-to-device copy of a state-dict tensor at create time; a refused copy (bad pointer, wrong device) fails the create call itself,
    // not just the stream synchronisation that ends it
    void copy_d2d(void* dst, const void* src, size_t bytes) {
        if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess && err == NOPE_OK) err = NOPE_ERR_LAUNCH;
    }
    void chk(int e) { if (e && err == NOPE_OK) err = e; }
    const nope_tensor_desc* get(const std::string& name, std::initializer_list<int64_t> shape) {
        auto it = tab.find(name);
        if (it == tab.end() || !it->second->data || it->second->ndim != (int)shape.size()) { fail(name); return nullptr; }
        int i = 0;
        for (int64_t v : shape) if (it->second->shape[i++] != v) { fail(name); return nullptr; }
        return it->second;
    }
    void* dmalloc(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { if (err == NOPE_OK) err = NOPE_ERR_ALLOC; return nullptr; }
        net->allocs.push_back(p);
        return p;
    }
    std::vector<void*> temps;                  // staging buffers of create time, freed after its final synchronize
    void* tmalloc(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { if (err == NOPE_OK) err = NOPE_ERR_ALLOC; return nullptr; }
        temps.push_back(p);
        return p;
    }
    void free_temps() { for (void* p : temps) hipFree(p); temps.clear(); }
    // GEGLU's projection (attention.py:37-44: `x, gate = proj(x).chunk(2, dim=-1)`): rows (x_j, gate_j) interleaved, so that a lane of the conv
    // epilogue holds whole pairs (ConvArgs::geglu); the unfused path reads the same layout (launch_geglu(..., interleaved))
    LConv conv_geglu(const std::string& pfx, int Cin, int D) {
        LConv c;
        c.Cin = Cin; c.Cout = 2 * D; c.mode = NOPE_CONV_PLAIN; c.ntaps = 1;
        const nope_tensor_desc* d = get(pfx + "weight", {2 * D, Cin});
        const nope_tensor_desc* bd = get(pfx + "bias", {2 * D});
        if (!d || !bd) return c;
        float* wi = (float*)tmalloc((size_t)2 * D * Cin * 4);
        c.w = dmalloc((size_t)2 * D * Cin * (size_t)dt_es(net->dt));
        c.bias = (float*)dmalloc((size_t)2 * D * 4);
        if (!wi || !c.w || !c.bias) return c;
        const size_t rb = (size_t)Cin * 4;
        const float* w0 = (const float*)d->data;
        const float* b0 = (const float*)bd->data;
        bool ok = hipMemcpy2DAsync(wi, 2 * rb, w0, rb, rb, D, hipMemcpyDeviceToDevice, s) == hipSuccess &&
                  hipMemcpy2DAsync(wi + Cin, 2 * rb, w0 + (size_t)D * Cin, rb, rb, D, hipMemcpyDeviceToDevice, s) == hipSuccess &&
                  hipMemcpy2DAsync(c.bias, 8, b0, 4, 4, D, hipMemcpyDeviceToDevice, s) == hipSuccess &&
                  hipMemcpy2DAsync(c.bias + 1, 8, b0 + D, 4, 4, D, hipMemcpyDeviceToDevice, s) == hipSuccess;
        if (!ok) { chk(NOPE_ERR_LAUNCH); return c; }
        chk(launch_pack_conv_w(net->dt, wi, c.w, 2 * D, Cin, 1, NOPE_CONV_PLAIN, s));
        return c;
    }
    float* copy_f32(const std::string& name, std::initializer_list<int64_t> shape) {
        const nope_tensor_desc* d = get(name, shape);
        if (!d) return nullptr;
        size_t n = 1;
        for (int64_t v : shape) n *= (size_t)v;
        float* p = (float*)dmalloc(n * 4);
        if (p) copy_d2d(p, d->data, n * 4);
        return p;
    }
    // conv (4-d weight) or linear (2-d weight) packed for the implicit-GEMM kernel
    // (Cin_pad > Cin: the kernel sees Cin_pad input channels, the last ones zero -- the 4-channel latent of vae_cin_ldm.yaml padded
    //  to one 16-byte vector of the 16-bit modes)
    LConv conv(const std::string& pfx, int Cin, int Cout, int ksz, int mode, bool has_bias, bool linear = false, int Cin_pad = 0) {
        LConv c;
        const int Ck = Cin_pad > Cin ? Cin_pad : Cin;
        c.Cin = Ck; c.Cout = Cout; c.mode = mode;
        c.ntaps = mode == NOPE_CONV_UP2P ? 4 : ksz * ksz;
        const nope_tensor_desc* d = linear ? get(pfx + "weight", {Cout, Cin}) : get(pfx + "weight", {Cout, Cin, ksz, ksz});
        if (d) {
            const size_t es = (size_t)dt_es(net->dt);
            c.w = dmalloc((size_t)Cout * c.ntaps * Ck * es * (mode == NOPE_CONV_UP2P ? 4 : 1));
            if (c.w) chk(launch_pack_conv_w(net->dt, d->data, c.w, Cout, Ck, c.ntaps, mode, s, nullptr, nullptr, Cin));
            if (net->x2 && !linear && ksz == 3 && (mode == NOPE_CONV_PLAIN || mode == NOPE_CONV_UP2P) && Ck == Cin && Cin % 32 == 0) {
                const size_t x2b = conv_w_x2_bytes(Cout, Cin, c.ntaps, mode);
                c.w_x2 = dmalloc(x2b);
                if (c.w_x2) { chk(launch_pack_conv_w_x2((const float*)d->data, c.w_x2, Cout, Cin, s, c.ntaps, mode)); c.x2_id = net->x2r.add_layer(c.w_x2, x2b); }
            }
        }
        if (has_bias) c.bias = copy_f32(pfx + "bias", {Cout});
        return c;
    }
    LNorm norm(const std::string& pfx, int C) {
        LNorm n;
        n.C = C;
        n.gamma = copy_f32(pfx + "weight", {C});
        n.beta = copy_f32(pfx + "bias", {C});
        return n;
    }
    LRes res(const std::string& p, int Cin, int Cout) {
        LRes r;
        r.Cin = Cin; r.Cout = Cout;
        r.n1 = norm(p + "in_layers.0.", Cin);
        r.c1 = conv(p + "in_layers.2.", Cin, Cout, 3, NOPE_CONV_PLAIN, true);
        const int film = net->cfg.use_scale_shift_norm ? 2 : 1;          // emb_layers.1: Linear(emb, 2 C) for FiLM, openaimodel.py:233-239
        r.emb_w = copy_f32(p + "emb_layers.1.weight", {film * Cout, net->emb_dim});
        r.emb_b = copy_f32(p + "emb_layers.1.bias", {film * Cout});
        r.n2 = norm(p + "out_layers.0.", Cout);
        r.c2 = conv(p + "out_layers.3.", Cout, Cout, 3, NOPE_CONV_PLAIN, true);
        r.has_skip = Cin != Cout;
        if (r.has_skip) r.skip = conv(p + "skip_connection.", Cin, Cout, 1, NOPE_CONV_PLAIN, true);
        if (!net->cfg.injecting_condition_twice && !net->cfg.use_scale_shift_norm && r.c1.bias && r.emb_b)     // emb == 0: emb_layers(emb) = its bias, folded into conv1's
            chk(launch_add_rowvec(NOPE_F32, r.c1.bias, r.c1.bias, r.emb_b, 1, 1, Cout, s));
        return r;
    }
    LST st(const std::string& p, int C) {
        LST T;
        T.C = C;
        const int ctx = net->cfg.context_dim;
        T.norm = norm(p + "norm.", C);
        T.proj_in = conv(p + "proj_in.", C, C, 1, NOPE_CONV_PLAIN, true);
        const int depth = net->cfg.transformer_depth > 0 ? net->cfg.transformer_depth : 1;
        for (int d = 0; d < depth; ++d) {                      // attention.py:251-258: `depth` blocks in sequence
        LTB t;
        const std::string b = p + "transformer_blocks." + std::to_string(d) + ".";
        t.ln1 = norm(b + "norm1.", C);
        t.ln3 = norm(b + "norm3.", C);
        get(b + "norm2.weight", {C}); get(b + "norm2.bias", {C});                     // validated, provably without effect (see header)
        get(b + "attn2.to_q.weight", {C, C}); get(b + "attn2.to_k.weight", {C, ctx});
        // attn1: q, k, v as one [3C][C] weight
        const nope_tensor_desc* wq = get(b + "attn1.to_q.weight", {C, C});
        const nope_tensor_desc* wk = get(b + "attn1.to_k.weight", {C, C});
        const nope_tensor_desc* wv = get(b + "attn1.to_v.weight", {C, C});
        t.qkv.Cin = C; t.qkv.Cout = 3 * C; t.qkv.ntaps = 1;
        if (wq && wk && wv) {
            float* cat = (float*)tmalloc((size_t)3 * C * C * 4);
            const size_t es = (size_t)dt_es(net->dt);
            t.qkv.w = dmalloc((size_t)3 * C * C * es);
            if (cat && t.qkv.w) {
                copy_d2d(cat, wq->data, (size_t)C * C * 4);
                copy_d2d(cat + (size_t)C * C, wk->data, (size_t)C * C * 4);
                copy_d2d(cat + (size_t)2 * C * C, wv->data, (size_t)C * C * 4);
                chk(launch_pack_conv_w(net->dt, cat, t.qkv.w, 3 * C, C, 1, NOPE_CONV_PLAIN, s));
            }
        }
        t.out1 = conv(b + "attn1.to_out.0.", C, C, 1, NOPE_CONV_PLAIN, true, true);
        {   // attn2: W = to_out.0.weight x to_v.weight, [C][ctx] (see the header)
            const nope_tensor_desc* wv2 = get(b + "attn2.to_v.weight", {C, ctx});
            const nope_tensor_desc* wo2 = get(b + "attn2.to_out.0.weight", {C, C});
            float* o2_b = copy_f32(b + "attn2.to_out.0.bias", {C});
            float* vT = (float*)dmalloc((size_t)ctx * C * 4);
            float* comb = (float*)dmalloc((size_t)C * ctx * 4);
            if (wv2 && wo2 && vT && comb) {
                chk(launch_nhwc_to_nchw_f32(NOPE_F32, wv2->data, vT, 1, ctx, C, s));                 // to_v.weight^T: [ctx][C]
                chk(launch_linear_naive(wo2->data, vT, nullptr, comb, C, ctx, C, 0, ctx, s));        // comb[c][j] = sum_k Wout[c][k] Wv[k][j]
            }
            t.u_off = u_total;
            u_parts.push_back(UPart{comb, o2_b, C, u_total});
            u_total += C;
        }
        t.ff1 = conv_geglu(b + "ff.net.0.proj.", C, 4 * C);
        t.ff2 = conv(b + "ff.net.2.", 4 * C, C, 1, NOPE_CONV_PLAIN, true, true);
        T.blocks.push_back(t);
        }
        T.proj_out = conv(p + "proj_out.", C, C, 1, NOPE_CONV_PLAIN, true);
        return T;
    }
};

struct Arena {
    unsigned char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    bool dry = false;
    void* alloc(size_t bytes) {
        const size_t o = align_up(off, 256);
        off = o + bytes;
        if (off > peak) peak = off;
        if (dry) return (void*)(uintptr_t)(0x1000 + o);
        if (off > cap) return nullptr;
        return base + o;
    }
};

struct Act { void* p = nullptr; int C = 0, H = 0, W = 0; };

struct Fwd {
    const nope_ldm* net;
    hipStream_t s;
    Arena ar;
    int nhyp = 0, err = NOPE_OK;
    size_t es = 4;
    float* gn_partial = nullptr;
    const float* ctx = nullptr;       // (nhyp, context_dim)
    const float* emb = nullptr;       // (nhyp, emb_dim) or null (zeros)
    const float* u_all = nullptr;     // (nhyp, u_total): every transformer block's to_out(to_v(context)) row
    X2Fwd x2;                         // NOPE_F16X2 range tracking of this forward (x2_range.h)
    bool tracking() const { return x2.on && err == NOPE_OK; }

    void chk(int e) { if (e != NOPE_OK && err == NOPE_OK) err = e; }
    bool live() const { return !ar.dry && err == NOPE_OK; }
    void* alloc_act(size_t elems) {
        void* p = ar.alloc(elems * es);
        if (!p && err == NOPE_OK) err = NOPE_ERR_WORKSPACE;
        return p;
    }
    float* alloc_f32(size_t n) {
        float* p = (float*)ar.alloc(n * 4);
        if (!p && err == NOPE_OK) err = NOPE_ERR_WORKSPACE;
        return p;
    }
    void conv(const LConv& c, const Act& a, void* out, int Ho, int Wo, const void* resid = nullptr, int out_nchw = 0, int out_dt = NOPE_F32,
              int rep = 1, int n = -1) {
        if (!live()) return;
        ConvArgs ca;
        ca.src1 = a.p; ca.C1 = a.C; ca.rep1 = rep; ca.Hs = a.H; ca.Ws = a.W; ca.Ho = Ho; ca.Wo = Wo;
        ca.mode = c.mode; ca.ntaps = c.ntaps; ca.w = c.w; ca.bias = c.bias; ca.resid = resid; ca.out = out; ca.Cout = c.Cout;
        ca.nhyp = n < 0 ? nhyp : n; ca.out_nchw = out_nchw; ca.out_dt = out_dt;
        if (a.C != c.Cin) { chk(NOPE_ERR_ARG); return; }
        if (c.w_x2 && !net->x2r.off) { ca.w_x2 = c.w_x2; ca.x2_t_zero = net->x2r.t_zero(c.x2_id) ? 1 : 0; }
        if (tracking()) {
            if (ca.w_x2 && conv_takes_x2(net->dt, ca)) {      // the two-pass tile: the layer's range shift follows its input's maximum
                x2.consumes(c.x2_id, x2.slot_for(a.p, (size_t)(ca.nhyp / rep) * a.H * a.W * a.C));
                chk(x2.err);
            }
            x2.overwritten(out);           // (conv epilogues record no maximum here: a two-pass consumer of `out` takes an absmax pass)
        }
        chk(launch_conv(net->dt, ca, s));
    }
    // y = [silu](GroupNorm(32, eps)(x))
    void gn(const LNorm& nm, const void* x, void* y, int HW, int act, float eps, const float* film = nullptr, int film_stride = 0) {
        if (!live()) return;
        const int nch = gn_stats_chunks(HW, nm.C, net->sdt);
        chk(launch_gn_stats(net->sdt, x, gn_partial, nhyp, HW, nm.C, 32, nch, s));
        GnApplyArgs ga;
        ga.x = x; ga.y = y; ga.partial = gn_partial; ga.nchunk = nch; ga.gamma = nm.gamma; ga.beta = nm.beta;
        ga.nhyp = nhyp; ga.HW = HW; ga.C = nm.C; ga.G = 32; ga.act = act; ga.eps = eps;
        ga.film = film; ga.film_stride = film_stride;
        ga.fast_silu = net->dt != NOPE_F32 ? 1 : 0;      // (f32 storage of the split-precision modes: hardware exp / rcp; the f32 mode keeps expf and the division)
        if (tracking()) {                                // (the FiLM instantiation records no maximum: its consumer takes an absmax pass)
            if (!film) { const int sl = x2.produce(y); if (sl >= 0) ga.amax_out = x2.slot_ptr(sl); }
            else x2.overwritten(y);
        }
        chk(launch_gn_apply(net->sdt, ga, s));
    }
    // ResBlock._forward, openaimodel.py:262-288 (no up/down)
    void res(const LRes& R, const Act& x, void* out) {
        const int HW = x.H * x.W;
        const size_t M = (size_t)nhyp * HW;
        const size_t mark = ar.off;
        const bool film_on = net->cfg.use_scale_shift_norm != 0;
        void* t = alloc_act(M * R.Cin);
        void* h = alloc_act(M * R.Cout);
        gn(R.n1, x.p, t, HW, 1, 1e-5f);
        conv(R.c1, Act{t, R.Cin, x.H, x.W}, h, x.H, x.W);
        const float* film = nullptr;                // FiLM rows [scale | shift]: emb == 0 -> the bias row, shared by every sample
        int film_stride = 0;
        if (film_on && !emb) film = R.emb_b;
        if (emb) {                                  // emb_layers(emb): added to h, or (FiLM) applied by the GroupNorm below
            const int ne = film_on ? 2 * R.Cout : R.Cout;
            float* e = alloc_f32((size_t)nhyp * ne);
            if (live()) {
                chk(launch_linear_naive(emb, R.emb_w, R.emb_b, e, nhyp, ne, net->emb_dim, 1, ne, s));
                if (!film_on) chk(launch_add_rowvec(net->sdt, h, h, e, (long long)M, HW, R.Cout, s));
            }
            if (film_on) { film = e; film_stride = ne; }
        }
        void* t2 = alloc_act(M * R.Cout);
        gn(R.n2, h, t2, HW, 1, 1e-5f, film, film_stride);
        const void* resid = x.p;
        if (R.has_skip) {
            void* sk = alloc_act(M * R.Cout);
            conv(R.skip, x, sk, x.H, x.W);
            resid = sk;
        }
        conv(R.c2, Act{t2, R.Cout, x.H, x.W}, out, x.H, x.W, resid);
        ar.off = mark;
    }
    // SpatialTransformer.forward, attention.py:264-277: GroupNorm, proj_in, `depth` BasicTransformerBlocks (:192-212), proj_out + x
    void st(const LST& T, const Act& x, void* out) {
        const int HW = x.H * x.W, C = T.C;
        const long long M = (long long)nhyp * HW;
        const size_t mark = ar.off;
        void* xn = alloc_act((size_t)M * C);
        void* tok = alloc_act((size_t)M * C);
        void* qkv = alloc_act((size_t)M * 3 * C);
        void* o = alloc_act((size_t)M * C);
        void* tok1 = alloc_act((size_t)M * C);
        void* g = alloc_act((size_t)M * 8 * C);
        void* gg = alloc_act((size_t)M * 4 * C);
        gn(T.norm, x.p, xn, HW, 0, 1e-6f);
        conv(T.proj_in, Act{xn, C, x.H, x.W}, tok, x.H, x.W);
        for (const LTB& B : T.blocks) {
            // attn1 (self-attention) + residual
            void* a = xn;                                    // reuse: LN1(tok)
            if (live()) chk(launch_layernorm(net->sdt, tok, a, B.ln1.gamma, B.ln1.beta, M, C, 1e-5f, s));
            conv(B.qkv, Act{a, C, x.H, x.W}, qkv, x.H, x.W);
            if (live()) chk(launch_token_attention(net->dt, qkv, o, nhyp, HW, C, 32, s));
            conv(B.out1, Act{o, C, x.H, x.W}, tok1, x.H, x.W, tok);
            // attn2 against the single pose token: + to_out(to_v(context)) for every token -- this block's slice of u_all
            if (live()) chk(launch_add_rowvec(net->sdt, tok1, tok1, u_all + B.u_off, M, HW, C, s, net->u_total));
            // feed-forward (GEGLU) + residual
            void* f = o;                                     // reuse: LN3(tok1)
            if (live()) chk(launch_layernorm(net->sdt, tok1, f, B.ln3.gamma, B.ln3.beta, M, C, 1e-5f, s));
            // (x_j, gate_j) column pairs: x * gelu(gate) in the projection's epilogue where the launch qualifies (16-bit modes on the
            //  128 x 192 kernel), else the projection as stored + geglu_kernel on the same layout -- bit-identical results
            if (live()) {
                ConvArgs ca;
                ca.src1 = f; ca.C1 = C; ca.Hs = ca.Ho = x.H; ca.Ws = ca.Wo = x.W; ca.ntaps = 1; ca.w = B.ff1.w; ca.bias = B.ff1.bias;
                ca.Cout = 8 * C; ca.nhyp = nhyp; ca.out = gg;
                if (conv_geglu_fusable(net->dt, ca)) {
                    ca.geglu = 1;
                    chk(launch_conv(net->dt, ca, s));
                } else {
                    conv(B.ff1, Act{f, C, x.H, x.W}, g, x.H, x.W);
                    chk(launch_geglu(net->sdt, g, gg, M, 4 * C, s, 1));
                }
            }
            conv(B.ff2, Act{gg, 4 * C, x.H, x.W}, tok, x.H, x.W, tok1);      // (tok is dead after the attn1 residual: the block's output)
        }
        conv(T.proj_out, Act{tok, C, x.H, x.W}, out, x.H, x.W, x.p);
        ar.off = mark;
    }
};

int run_forward(const nope_ldm* net, const float* x, int n_src, int x_rep, const float* pose, int n_hyp, int H, int W, void* out,
                int out_dtype, void* ws, size_t ws_bytes, hipStream_t s, bool dry, size_t* peak) {
    const nope_ldm_config& cfg = net->cfg;
    Fwd f;
    f.net = net; f.s = s; f.nhyp = n_hyp; f.es = (size_t)dt_es(net->dt);
    f.ar.base = (unsigned char*)ws; f.ar.cap = ws_bytes; f.ar.dry = dry;
    f.x2.r = &net->x2r; f.x2.s = s; f.x2.on = net->x2 && net->x2r.active() && !dry;
    const int HW = H * W;
    const int cin_k = net->conv_in.Cin;          // in_channels rounded up to a whole 16-byte vector
    void* x_in = f.alloc_act((size_t)n_src * HW * cin_k);
    float* ctx = f.alloc_f32((size_t)n_hyp * cfg.context_dim);
    float* ctx2 = f.alloc_f32((size_t)n_hyp * cfg.context_dim);
    float* emb = cfg.injecting_condition_twice ? f.alloc_f32((size_t)n_hyp * net->emb_dim) : nullptr;
    float* u_all = f.alloc_f32((size_t)n_hyp * (net->u_total > 0 ? net->u_total : 1));
    f.u_all = u_all;
    f.gn_partial = f.alloc_f32((size_t)n_hyp * 16 * 32 * 2);
    if (f.err) return f.err;
    if (f.live()) {
        f.chk(launch_nchw_to_nhwc(net->sdt, x, x_in, n_src, cin_k, HW, s, cfg.in_channels));
        // context = pose_mlp(pose), adapt_openaimodel.py:105-116,145
        f.chk(launch_linear_naive(pose, net->pose_w0, net->pose_b0, ctx, n_hyp, cfg.context_dim, cfg.pose_dim, 0, cfg.context_dim, s));
        if (cfg.pose_mlp_layers == 2) {
            f.chk(launch_linear_naive(ctx, net->pose_w2, net->pose_b2, ctx2, n_hyp, cfg.context_dim, cfg.context_dim, 2, cfg.context_dim, s));
            f.ctx = ctx2;
        } else f.ctx = ctx;
        if (net->u_total > 0)     // attn2 of every transformer block at once (see the header)
            f.chk(launch_linear_naive(f.ctx, net->u_w, net->u_b, u_all, n_hyp, net->u_total, cfg.context_dim, 0, net->u_total, s));
        if (emb) {     // emb = pose_mlp_timesteps(pose), :119-123,141-142
            f.chk(launch_linear_naive(pose, net->tw, net->tb, emb, n_hyp, net->emb_dim, cfg.pose_dim, 0, net->emb_dim, s));
            f.emb = emb;
        }
    } else if (emb) f.emb = emb;

    // persistent skip stack
    std::vector<Act> hs;
    int curH = H, curW = W;
    // input_blocks[0]: the input conv, evaluated once per hypothesis from the shared latent (source broadcast)
    Act h{f.alloc_act((size_t)n_hyp * HW * net->conv_in.Cout), net->conv_in.Cout, H, W};
    f.conv(net->conv_in, Act{x_in, cin_k, H, W}, h.p, H, W, nullptr, 0, NOPE_F32, x_rep);
    hs.push_back(h);
    for (size_t b = 1; b < net->input_blocks.size(); ++b) {
        const LBlock& B = net->input_blocks[b];
        Act nxt;
        if (B.has_resample) {            // Downsample: conv 3x3, stride 2, pad 1 (openaimodel.py:143-174)
            nxt = Act{f.alloc_act((size_t)n_hyp * (curH / 2) * (curW / 2) * B.resample.Cout), B.resample.Cout, curH / 2, curW / 2};
            f.conv(B.resample, h, nxt.p, curH / 2, curW / 2);
            curH /= 2; curW /= 2;
        } else {
            nxt = Act{f.alloc_act((size_t)n_hyp * curH * curW * B.res.Cout), B.res.Cout, curH, curW};
            if (B.has_st) {
                const size_t mark = f.ar.off;
                void* t = f.alloc_act((size_t)n_hyp * curH * curW * B.res.Cout);
                f.res(B.res, h, t);
                f.st(B.st, Act{t, B.res.Cout, curH, curW}, nxt.p);
                f.ar.off = mark;
            } else f.res(B.res, h, nxt.p);
        }
        h = nxt;
        hs.push_back(h);
    }
    // middle block
    {
        const size_t e = (size_t)n_hyp * curH * curW * h.C;
        Act a{f.alloc_act(e), h.C, curH, curW}, b{f.alloc_act(e), h.C, curH, curW}, c{f.alloc_act(e), h.C, curH, curW};
        f.res(net->mid1, h, a.p);
        f.st(net->mid_st, a, b.p);
        f.res(net->mid2, b, c.p);
        h = c;
    }
    // output blocks
    for (size_t b = 0; b < net->output_blocks.size(); ++b) {
        const LBlock& B = net->output_blocks[b];
        const Act sk = hs.back();
        hs.pop_back();
        const long long M = (long long)n_hyp * curH * curW;
        Act cat{f.alloc_act((size_t)M * (h.C + sk.C)), h.C + sk.C, curH, curW};
        if (f.live()) {
            f.chk(launch_copy_cols(net->sdt, h.p, cat.p, M, h.C, cat.C, 0, s));
            f.chk(launch_copy_cols(net->sdt, sk.p, cat.p, M, sk.C, cat.C, h.C, s));
        }
        Act r{f.alloc_act((size_t)M * B.res.Cout), B.res.Cout, curH, curW};
        f.res(B.res, cat, r.p);
        if (B.has_st) {
            Act t{f.alloc_act((size_t)M * B.res.Cout), B.res.Cout, curH, curW};
            f.st(B.st, r, t.p);
            r = t;
        }
        if (B.has_resample) {            // Upsample: nearest x2 + conv 3x3 (openaimodel.py:93-124) as four 2x2 phase convs
            Act u{f.alloc_act((size_t)M * 4 * B.resample.Cout), B.resample.Cout, curH * 2, curW * 2};
            f.conv(B.resample, r, u.p, curH * 2, curW * 2);
            curH *= 2; curW *= 2;
            r = u;
        }
        h = r;
    }
    // out: GroupNorm32 + SiLU + conv 3x3 (openaimodel.py:733-737) straight into the NCHW output
    {
        void* t = f.alloc_act((size_t)n_hyp * HW * h.C);
        f.gn(net->norm_out, h.p, t, HW, 1, 1e-5f);
        f.conv(net->conv_out, Act{t, h.C, H, W}, out, H, W, nullptr, 1, out_dtype);
    }
    if (f.tracking())      // the forward's verdict; NaNs over the output of a forward whose layers left their windows (x2_range.h)
        f.chk(f.x2.finish(out, (size_t)n_hyp * net->cfg.out_channels * HW * (size_t)(out_dtype == NOPE_F32 ? 4 : 2), out_dtype));
    if (peak) *peak = f.ar.peak;
    return f.err;
}

int check_shape(const nope_ldm* net, int n_hyp, int n_src, int x_rep, int H, int W) {
    if (!net || n_hyp <= 0 || n_src <= 0 || x_rep <= 0 || (long long)n_src * x_rep != n_hyp || H <= 0 || W <= 0) return NOPE_ERR_ARG;
    const int f = 1 << (net->cfg.n_levels - 1);
    if (H % f || W % f) return NOPE_ERR_UNSUPPORTED;
    return NOPE_OK;
}

}  // namespace

extern "C" {

int nope_ldm_create(const nope_ldm_config* cfg, const nope_tensor_desc* tensors, int n_tensors, nope_stream_t stream, nope_ldm** out) {
    if (!cfg || !tensors || !out || n_tensors <= 0) return NOPE_ERR_ARG;
    if (cfg->n_levels < 1 || cfg->n_levels > 8 || cfg->num_res_blocks < 1 || cfg->num_head_channels != 32) return NOPE_ERR_UNSUPPORTED;
    if (!dt_is_compute(cfg->compute_dtype)) return NOPE_ERR_UNSUPPORTED;
    if (cfg->pose_mlp_layers != 1 && cfg->pose_mlp_layers != 2) return NOPE_ERR_UNSUPPORTED;
    if (cfg->model_channels % 32 || cfg->in_channels < 1 || cfg->context_dim <= 0 || cfg->transformer_depth < 0 || cfg->transformer_depth > 16) return NOPE_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    nope_ldm* net = new nope_ldm();
    net->cfg = *cfg;
    net->x2 = cfg->compute_dtype == NOPE_F16X2;
    net->dt = dt_base(cfg->compute_dtype);
    net->sdt = dt_storage(net->dt);
    net->emb_dim = cfg->model_channels * 4;
    const int mc = cfg->model_channels;
    Loader ld;
    ld.net = net; ld.s = s;
    for (int i = 0; i < n_tensors; ++i)
        if (tensors[i].name) ld.tab[tensors[i].name] = &tensors[i];

    net->pose_w0 = ld.copy_f32("pose_mlp.0.weight", {cfg->context_dim, cfg->pose_dim});
    net->pose_b0 = ld.copy_f32("pose_mlp.0.bias", {cfg->context_dim});
    if (cfg->pose_mlp_layers == 2) {
        net->pose_w2 = ld.copy_f32("pose_mlp.2.weight", {cfg->context_dim, cfg->context_dim});
        net->pose_b2 = ld.copy_f32("pose_mlp.2.bias", {cfg->context_dim});
    }
    if (cfg->injecting_condition_twice) {
        net->tw = ld.copy_f32("pose_mlp_timesteps.0.weight", {net->emb_dim, cfg->pose_dim});
        net->tb = ld.copy_f32("pose_mlp_timesteps.0.bias", {net->emb_dim});
    }
    // openaimodel.py:511-612 -- input blocks
    net->conv_in = ld.conv("input_blocks.0.0.", cfg->in_channels, mc, 3, NOPE_CONV_PLAIN, true, false, (cfg->in_channels + 7) / 8 * 8);
    net->input_blocks.emplace_back();
    std::vector<int> chans{mc};
    int ch = mc, idx = 1;
    for (int level = 0; level < cfg->n_levels; ++level) {
        for (int r = 0; r < cfg->num_res_blocks; ++r) {
            LBlock B;
            const std::string p = "input_blocks." + std::to_string(idx) + ".";
            B.has_res = true;
            B.res = ld.res(p + "0.", ch, cfg->channel_mult[level] * mc);
            ch = cfg->channel_mult[level] * mc;
            if (cfg->attn_levels[level]) { B.has_st = true; B.st = ld.st(p + "1.", ch); }
            net->input_blocks.push_back(B);
            chans.push_back(ch);
            ++idx;
        }
        if (level != cfg->n_levels - 1) {
            LBlock B;
            B.has_resample = true;
            B.resample = ld.conv("input_blocks." + std::to_string(idx) + ".0.op.", ch, ch, 3, NOPE_CONV_STRIDE2, true);
            net->input_blocks.push_back(B);
            chans.push_back(ch);
            ++idx;
        }
    }
    // :618-648 -- middle block
    net->mid1 = ld.res("middle_block.0.", ch, ch);
    net->mid_st = ld.st("middle_block.1.", ch);
    net->mid2 = ld.res("middle_block.2.", ch, ch);
    // :651-731 -- output blocks
    idx = 0;
    for (int level = cfg->n_levels - 1; level >= 0; --level) {
        for (int i = 0; i <= cfg->num_res_blocks; ++i) {
            const int ich = chans.back();
            chans.pop_back();
            LBlock B;
            const std::string p = "output_blocks." + std::to_string(idx) + ".";
            B.has_res = true;
            B.res = ld.res(p + "0.", ch + ich, mc * cfg->channel_mult[level]);
            ch = mc * cfg->channel_mult[level];
            int sub = 1;
            if (cfg->attn_levels[level]) { B.has_st = true; B.st = ld.st(p + std::to_string(sub++) + ".", ch); }
            if (level && i == cfg->num_res_blocks) {
                B.has_resample = true;
                B.resample = ld.conv(p + std::to_string(sub) + ".conv.", ch, ch, 3, NOPE_CONV_UP2P, true);
            }
            net->output_blocks.push_back(B);
            ++idx;
        }
    }
    net->norm_out = ld.norm("out.0.", ch);
    net->conv_out = ld.conv("out.2.", mc, cfg->out_channels, 3, NOPE_CONV_PLAIN, true);
    if (ch != mc) ld.fail("out.2.weight");

    // stack the attn2 maps of all transformer blocks (in load order: LST::u_off)
    net->u_total = ld.u_total;
    if (ld.err == NOPE_OK && ld.u_total > 0) {
        const int ctx = cfg->context_dim;
        net->u_w = (float*)ld.dmalloc((size_t)ld.u_total * ctx * 4);
        net->u_b = (float*)ld.dmalloc((size_t)ld.u_total * 4);
        if (net->u_w && net->u_b)
            for (const auto& up : ld.u_parts) {
                if (!up.comb || !up.bias) continue;
                ld.copy_d2d(net->u_w + (size_t)up.off * ctx, up.comb, (size_t)up.C * ctx * 4);
                ld.copy_d2d(net->u_b + up.off, up.bias, (size_t)up.C * 4);
            }
    }
    if (ld.err == NOPE_OK) { const int e = net->x2r.init([&](size_t bytes) { return ld.dmalloc(bytes); }, s); if (e) ld.err = e; }
    if (hipStreamSynchronize(s) != hipSuccess && ld.err == NOPE_OK) ld.err = NOPE_ERR_LAUNCH;
    ld.free_temps();
    if (ld.err != NOPE_OK) {
        if (!ld.missing.empty()) fprintf(stderr, "nope_ldm_create: missing or mis-shaped tensor '%s'\n", ld.missing.c_str());
        nope_ldm_destroy(net);
        return ld.err;
    }
    *out = net;
    return NOPE_OK;
}

void nope_ldm_destroy(nope_ldm* net) {
    if (!net) return;
    for (void* p : net->allocs) hipFree(p);
    net->x2r.destroy();
    delete net;
}

// NOPE_F16X2 activation ranges of the LDM variant: as nope_unet_x2_poll / _x2_range_check / _x2_enable (include/nope_hip.h)
int nope_ldm_x2_poll(nope_ldm* net, nope_stream_t stream, int* n_out_of_range, int* n_adjusted, float* max_abs) {
    if (n_out_of_range) *n_out_of_range = 0;
    if (n_adjusted) *n_adjusted = 0;
    if (max_abs) *max_abs = 0.f;
    if (!net) return NOPE_ERR_ARG;
    if (!net->x2) return NOPE_OK;
    return net->x2r.poll((hipStream_t)stream, n_out_of_range, n_adjusted, max_abs);
}
int nope_ldm_x2_range_check(nope_ldm* net, nope_stream_t stream, int* n_out_of_range, int* n_adjusted, float* max_abs) {
    if (!net) return NOPE_ERR_ARG;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return NOPE_ERR_LAUNCH;
    return nope_ldm_x2_poll(net, stream, n_out_of_range, n_adjusted, max_abs);
}
int nope_ldm_x2_enable(nope_ldm* net, int on) {
    if (!net) return NOPE_ERR_ARG;
    net->x2r.off = on == 0;
    return NOPE_OK;
}

size_t nope_ldm_workspace_bytes(const nope_ldm* net, int n_hyp, int n_src, int H, int W) {
    if (!net || n_src <= 0 || n_hyp % n_src) return 0;
    if (check_shape(net, n_hyp, n_src, n_hyp / n_src, H, W) != NOPE_OK) return 0;
    size_t peak = 0;
    run_forward(net, nullptr, n_src, n_hyp / n_src, nullptr, n_hyp, H, W, nullptr, NOPE_F32, nullptr, 0, nullptr, true, &peak);
    return align_up(peak, 256) + 256;
}

int nope_ldm_forward(const nope_ldm* net, const float* x, int n_src, int x_rep, const float* pose, int n_hyp, int H, int W, void* out,
                     int out_dtype, void* workspace, size_t workspace_bytes, nope_stream_t stream) {
    int e = check_shape(net, n_hyp, n_src, x_rep, H, W);
    if (e) return e;
    if (!x || !pose || !out || !workspace) return NOPE_ERR_ARG;
    if (out_dtype != NOPE_F32 && out_dtype != NOPE_BF16 && out_dtype != NOPE_F16) return NOPE_ERR_UNSUPPORTED;
    unsigned char* base = (unsigned char*)(((uintptr_t)workspace + 255) / 256 * 256);
    const size_t lost = (size_t)(base - (unsigned char*)workspace);
    if (workspace_bytes < lost) return NOPE_ERR_WORKSPACE;
    if (net->x2 && net->x2r.active()) (void)net->x2r.poll((hipStream_t)stream, nullptr, nullptr, nullptr);      // verdicts that have arrived: re-centre first
    return run_forward(net, x, n_src, x_rep, pose, n_hyp, H, W, out, out_dtype, base, workspace_bytes - lost, (hipStream_t)stream, false, nullptr);
}

}  // extern "C"
