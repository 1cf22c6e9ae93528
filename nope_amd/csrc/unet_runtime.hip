// Host-side runtime of the pose-conditioned U-Net: weight repacking (nope_unet_create) and
// the launch schedule of one forward over a batch of pose hypotheses (nope_unet_forward).
//
// Op order follows UNet.forward, u_net.py:160-198, exactly (mid block twice with shared
// weights :177-183, final_conv.0 without the embedding :154-157,197).  What differs is the
// execution model, chosen for MI355X:
//   * NHWC activations, every conv/linear an implicit GEMM on MFMA (kernels_gemm.hip);
//     torch.cat / nn.Upsample / Rearrange are folded into that kernel's loader, so none of
//     those tensors ever exists in HBM;
//   * all N pose hypotheses of a reference image run as ONE batch (n_hyp = B*N): at the 4x4
//     bottleneck a single hypothesis has only 16 GEMM rows, the batch has 16*n_hyp;
//   * pose-independent work is done once per reference image instead of once per template:
//     init_conv, the skip `r`, and conv+GroupNorm+SiLU of downs[0][0].block1 (the embedding is
//     only added after that activation, model_utils.py:272-276) -- the reference re-runs all
//     of it, and the ResNet encoder, N times (model.py:115,219);
//   * the 19 per-block `Linear(SiLU(c))` embeddings (model_utils.py:261-265,274-275) are one
//     f32 GEMM against the row-concatenated weights;
//   * a bump arena over caller-provided workspace: no allocation, no host sync, one stream.
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "nope_common.h"
#include "x2_range.h"

using namespace nope;

namespace {

struct Conv { void* w = nullptr; float* bias = nullptr; int Cin = 0, Cout = 0, ntaps = 1, mode = NOPE_CONV_PLAIN; void* w_x2 = nullptr; int x2_id = -1; };   // w_x2: NOPE_F16X2 only: the same weights in the f16 + MX-fp8 tile's layout; x2_id: its slot in the net's range table
struct Norm { float* gamma = nullptr; float* beta = nullptr; int C = 0; };
struct Res { Conv c1, c2, res; Norm n1, n2; bool has_res = false; int emb_off = -1; };
// PreNorm's GroupNorm(1) is folded into the qkv conv: gamma into the packed weights, c0 = W beta, c1 = W gamma
// (f32 [3*heads*dim_head]); mean / rstd enter in the conv epilogue.
struct LinAttn { Norm pre, post; Conv qkv, out; float *c0 = nullptr, *c1 = nullptr; };
struct Attn { Norm pre; Conv qkv, out; float *c0 = nullptr, *c1 = nullptr; };
struct Level { Res r0, r1; LinAttn attn; Conv resample; };

struct Act { void* p = nullptr; int C = 0, H = 0, W = 0, rep = 1; };

}  // namespace

// One captured launch sequence of a forward: valid for exactly this (workspace, shape, output type); x, pose and the output are
// staged through the workspace so that the caller's pointers stay out of the graph.
struct UGraph { void* ws; size_t ws_bytes; int n_hyp, n_src, H, W, out_dt; hipGraphExec_t exec; };

struct nope_unet {
    nope_unet_config cfg;
    int dt = NOPE_F32;      // compute dtype: what the conv kernels and the weight packing see
    int sdt = NOPE_F32;     // storage dtype of the activations: what every other kernel sees (NOPE_BF16X3 keeps f32 activations)
    bool x2 = false;        // NOPE_F16X2: dt = NOPE_BF16X3 everywhere, plus a second weight pack per 3x3 layer for the tap-resident kernel's f16 + MX-fp8 tile
    std::vector<void*> allocs;
    int dims[9];
    int classes = 0;
    Conv init_conv, final_conv1;
    float* final_w_raw = nullptr;      // final_conv.1.weight as stored, [out_dim][u_net_dim] f32: the fused tail (gn_apply_proj, kernels_norm.hip)
    Level downs[8], ups[8];
    Res mid1, mid2, final_res, final_conv0;
    Attn mid_attn;
    float *pose_w0 = nullptr, *pose_b0 = nullptr, *pose_w2 = nullptr, *pose_b2 = nullptr;
    float *emb_w = nullptr, *emb_b = nullptr;
    int emb_total = 0;
    // optional per-launch timing of the implicit-GEMM kernel (bench.py roofline leg)
    struct Ev { hipEvent_t a, b; double flops, bytes; nope_conv_launch_info info; };
    mutable bool profile = false;
    mutable std::vector<Ev> evs;
    // hipGraph cache for SMALL hypothesis batches (the reference evaluates on 26 / 91 / 341 templates, shapeNet.py:252-263): there
    // a forward is ~150 dependent launches of 5-20 us each and the gaps between them are a visible share of the pass.
    // OPT-IN (nope_unet_graph_limit, or NOPE_UNET_GRAPH read once at create time; default 0 = never): measured +-0 against direct launches
    // (profiles/r03c_small_banks.txt -- a 64-hypothesis pass is GPU-bound, not launch-bound), and a cached graph freezes the launch plan
    // it was captured with (the NOPE_* tuning variables are not part of the key).  The cache is guarded by a mutex: ctypes callers
    // have released the GIL.
    mutable std::vector<UGraph> graphs;
    mutable bool graphs_ok = true;           // cleared when capture is unavailable: direct launches from then on
    mutable std::mutex graph_mu;
    mutable int graph_replays = 0;           // forwards served by a graph replay since create (tests assert the path really ran)
    long long graph_max = 0;                 // largest n_hyp * H * W that replays a graph; 0 = off
    // NOPE_F16X2 activation ranges: per-layer shifts of the tile's operands, producer-side maxima, device-side verdict (x2_range.h)
    mutable X2Range x2r;
};

namespace {

struct Loader {
    nope_unet* net;
    hipStream_t s;
    std::map<std::string, const nope_tensor_desc*> tab;
    int err = NOPE_OK;
    std::string missing;

    const nope_tensor_desc* get(const std::string& name, std::initializer_list<int64_t> shape) {
        auto it = tab.find(name);
        if (it == tab.end() || !it->second->data) { fail(name); return nullptr; }
        const nope_tensor_desc* d = it->second;
        if (d->ndim != (int)shape.size()) { fail(name); return nullptr; }
        int i = 0;
        for (int64_t v : shape) if (d->shape[i++] != v) { fail(name); return nullptr; }
        return d;
    }
    void fail(const std::string& n) { if (err == NOPE_OK) { err = NOPE_ERR_WEIGHT; missing = n; } }
    // device-to-device copy of a state-dict tensor at create time; a refused copy (bad pointer, wrong device) fails the create call itself,
    // not just the stream synchronisation that ends it
    void copy_d2d(void* dst, const void* src, size_t bytes) {
        if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess && err == NOPE_OK) err = NOPE_ERR_LAUNCH;
    }
    void* dmalloc(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { if (err == NOPE_OK) err = NOPE_ERR_ALLOC; return nullptr; }
        net->allocs.push_back(p);
        return p;
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
    // (ksz 4 with UP2P: the weight is a ConvTranspose2d(4, 2, 1)'s, [Cin][Cout][4][4], repacked into the same four phase sets)
    // (Cin_pad > Cin: the kernel sees Cin_pad input channels, the last ones zero -- a latent whose channel count is not a whole 16-byte
    //  vector, e.g. a 4-channel VAE latent, zero-padded at pack time and in the NHWC input)
    Conv conv(const std::string& pfx, int Cin, int Cout, int ksz, int mode, bool has_bias, const float* cin_scale = nullptr, int Cin_pad = 0) {
        Conv c;
        const int Csrc = Cin;
        if (Cin_pad > Cin) Cin = Cin_pad;
        c.Cin = Cin; c.Cout = Cout; c.mode = mode;
        c.ntaps = (mode == NOPE_CONV_DOWN2 || mode == NOPE_CONV_UP2P) ? 4 : ksz * ksz;
        const bool convT = mode == NOPE_CONV_UP2P && ksz == 4;
        const nope_tensor_desc* d = mode == NOPE_CONV_DOWN2 ? get(pfx + "weight", {Cout, (int64_t)Cin * 4, 1, 1})
                                    : convT             ? get(pfx + "weight", {Cin, Cout, 4, 4})
                                                        : get(pfx + "weight", {Cout, Csrc, ksz, ksz});
        if (d) {
            const size_t es = (size_t)dt_es(net->dt);
            c.w = dmalloc((size_t)Cout * c.ntaps * Cin * es * (mode == NOPE_CONV_UP2P ? 4 : 1));
            if (c.w) { int e = launch_pack_conv_w(net->dt, d->data, c.w, Cout, Cin, convT ? 16 : c.ntaps, mode, s, cin_scale, nullptr, Csrc); if (e && err == NOPE_OK) err = e; }
            // NOPE_F16X2: every layer a ping-pong kernel may run (3x3, 1x1, space-to-depth, phase convs) carries the second pack; launch_conv
            // takes it when the launch's shape lands on one of them
            if (net->x2 && (mode == NOPE_CONV_PLAIN || mode == NOPE_CONV_DOWN2 || mode == NOPE_CONV_UP2P) && (ksz == 3 || ksz == 1 || mode != NOPE_CONV_PLAIN) &&
                !cin_scale && Csrc == Cin && Cin % 32 == 0) {
                const size_t x2b = conv_w_x2_bytes(Cout, Cin, c.ntaps, mode);
                c.w_x2 = dmalloc(x2b);
                if (c.w_x2) {
                    int e = launch_pack_conv_w_x2((const float*)d->data, c.w_x2, Cout, Cin, s, convT ? 16 : c.ntaps, mode); if (e && err == NOPE_OK) err = e;
                    c.x2_id = net->x2r.add_layer(c.w_x2, x2b);
                }
            }
        }
        if (has_bias) c.bias = copy_f32(pfx + "bias", {Cout});
        return c;
    }
    Norm norm(const std::string& pfx, int C) {
        Norm n;
        n.C = C;
        n.gamma = copy_f32(pfx + "weight", {C});
        n.beta = copy_f32(pfx + "bias", {C});
        return n;
    }
    // qkv conv of an attention block with its PreNorm folded in (see LinAttn)
    void prenorm_qkv(const std::string& p, int C, int N, Norm& pre, Conv& qkv, float*& c0, float*& c1) {
        pre = norm(p + "fn.norm.", C);
        qkv = conv(p + "fn.fn.to_qkv.", C, N, 1, NOPE_CONV_PLAIN, false, pre.gamma);
        c0 = (float*)dmalloc((size_t)N * 4);
        c1 = (float*)dmalloc((size_t)N * 4);
        const nope_tensor_desc* w = get(p + "fn.fn.to_qkv.weight", {N, C, 1, 1});
        if (w && c0 && c1 && pre.beta && qkv.w) {
            int e = launch_linear_naive(pre.beta, w->data, nullptr, c0, 1, N, C, 0, N, s);
            if (!e) e = launch_rowsum(net->dt, qkv.w, c1, N, C, s);
            if (e && err == NOPE_OK) err = e;
        }
    }
    Res res(const std::string& pfx, int Cin, int Cout, bool use_emb, std::vector<std::pair<std::string, int>>& embs) {
        Res r;
        r.c1 = conv(pfx + "block1.proj.", Cin, Cout, 3, NOPE_CONV_PLAIN, true);
        r.n1 = norm(pfx + "block1.norm.", Cout);
        r.c2 = conv(pfx + "block2.proj.", Cout, Cout, 3, NOPE_CONV_PLAIN, true);
        r.n2 = norm(pfx + "block2.norm.", Cout);
        r.has_res = Cin != Cout;
        if (r.has_res) r.res = conv(pfx + "res_conv.", Cin, Cout, 1, NOPE_CONV_PLAIN, true);
        if (use_emb) { r.emb_off = net->emb_total; net->emb_total += Cout; embs.push_back({pfx + "mlp.1.", Cout}); }
        return r;
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
        if (dry) return (void*)(uintptr_t)(0x1000 + o);   // never dereferenced
        if (off > cap) return nullptr;
        return base + o;
    }
};

// Fused GroupNorm statistics of a conv output: [n][blocks][C][2] column sums per row block (null: the conv cannot emit them --
// the GroupNorm takes its own statistics pass)
struct Stats { float* cs = nullptr; int blocks = 0; };

struct Fwd {
    const nope_unet* net;
    hipStream_t s;
    Arena ar;
    int nhyp = 0, err = NOPE_OK;
    size_t es = 4;
    float* gn_partial = nullptr;
    float* pn_partial = nullptr;   // (sum, sum sq) partials of the tensor that feeds the next attention block
    float* pn_ms = nullptr;        // its per-hypothesis (mean, rstd)
    const float* emb_all = nullptr;

    X2Fwd x2;                                    // NOPE_F16X2 range tracking of this forward (x2_range.h)
    bool tracking() const { return x2.on && err == NOPE_OK; }

    bool dry() const { return ar.dry; }
    void chk(int e) { if (e != NOPE_OK && err == NOPE_OK) err = e; }
    void* alloc_act(size_t elems) {
        void* p = ar.alloc(elems * es);
        if (!p && err == NOPE_OK) err = NOPE_ERR_WORKSPACE;
        return p;
    }
    bool live() const { return !ar.dry && err == NOPE_OK; }

    // out = conv(a [cat b]) (+bias) (+resid);  n = number of samples computed (nhyp or fewer).  `stats`: also emit the column
    // statistics of the output when this launch can (conv_stat_rows: whole 64-row blocks per sample on every kernel, 16 / 32-pixel
    // maps on the small-tile kernel); the scratch comes from the arena and lives until the caller's release.
    void conv(const Conv& c, const Act& a, const Act* b, void* out, int Ho, int Wo, int n, int rep1, int rep2,
              const void* resid = nullptr, int out_nchw = 0, int out_dt = NOPE_F32, Stats* stats = nullptr,
              const float* pn_c0 = nullptr, const float* pn_c1 = nullptr, bool track_out = false) {
        if (err != NOPE_OK) return;
        ConvArgs ca;
        if (pn_c0) { ca.pn_ms = pn_ms; ca.pn_c0 = pn_c0; ca.pn_c1 = pn_c1; }
        ca.src1 = a.p; ca.C1 = a.C; ca.rep1 = rep1;
        if (b) { ca.src2 = b->p; ca.C2 = b->C; ca.rep2 = rep2; }
        ca.Hs = a.H; ca.Ws = a.W; ca.Ho = Ho; ca.Wo = Wo;
        ca.mode = c.mode; ca.ntaps = c.ntaps; ca.w = c.w; ca.w_x2 = net->x2r.off ? nullptr : c.w_x2; ca.bias = c.bias; ca.resid = resid;
        if (ca.w_x2 && c.x2_id >= 0) ca.x2_t_zero = net->x2r.t_zero(c.x2_id) ? 1 : 0;
        ca.out = out; ca.Cout = c.Cout; ca.nhyp = n; ca.out_nchw = out_nchw; ca.out_dt = out_dt;
        if (a.C + (b ? b->C : 0) != c.Cin) { chk(NOPE_ERR_ARG); return; }
        float* colstats = nullptr;
        if (stats) {
            const int sr = conv_stat_rows(net->dt, ca);
            if (sr > 0) {
                const int HWo = Ho * Wo;
                colstats = (float*)ar.alloc((size_t)n * (HWo / sr) * c.Cout * 2 * sizeof(float));
                if (!colstats) { chk(NOPE_ERR_WORKSPACE); return; }
                ca.colstats = colstats; ca.stat_rows = sr;
                stats->cs = colstats; stats->blocks = HWo / sr;
            }
        }
        // few output tiles + long K (small hypothesis batches at the 4x4 level): deterministic split-K through a
        // scratch taken from the arena for the duration of the launch
        const size_t sk_mark = ar.off;
        if (!pn_c0 && !out_nchw) {
            const int S = conv_splitk_factor(net->dt, ca);      // (1 for a conv that emits its statistics itself)
            if (S > 1) {
                ca.splitk_bytes = (size_t)S * n * Ho * Wo * c.Cout * 4;
                ca.splitk_ws = ar.alloc(ca.splitk_bytes);
                if (!ca.splitk_ws) { chk(NOPE_ERR_WORKSPACE); return; }
            }
        }
        struct Release { Arena& a; size_t m; ~Release() { a.off = m; } } release{ar, sk_mark};
        if (!live()) return;               // workspace-size query: only the arena bookkeeping above matters
        if (tracking()) {
            if (ca.w_x2 && c.x2_id >= 0 && conv_takes_x2(net->dt, ca)) {      // this launch runs the two-pass tile: its layer's shift follows its inputs' maxima
                x2.consumes(c.x2_id, x2.slot_for(a.p, (size_t)(n / rep1) * a.H * a.W * a.C));
                if (b) x2.consumes(c.x2_id, x2.slot_for(b->p, (size_t)(n / rep2) * b->H * b->W * b->C));
                chk(x2.err);
            }
            // `track_out`: this conv's output goes straight into f16x2 convs (the resampling convs, the bottleneck attention's output
            // projection): its epilogue records max |out| when it is one that can (the wide NHWC epilogue); otherwise a later f16x2
            // consumer of `out` takes an absmax pass over it
            ConvArgs probe = ca;
            probe.out_amax = net->x2r.amax;
            if (track_out && conv_records_out_amax(net->dt, probe)) {
                const int sl = x2.produce(out);
                if (sl >= 0) ca.out_amax = x2.slot_ptr(sl);
            } else x2.overwritten(out);
        }
        if (net->profile) {
            nope_unet::Ev ev;
            hipEventCreate(&ev.a); hipEventCreate(&ev.b);
            ev.flops = conv_executed_flops(net->dt, ca);   // executed MACs (UP2P: 4 taps per output pixel; padding taps of small maps skipped)
            // algorithmic HBM bytes: every input, weight and output element exactly once
            ev.bytes = ((double)(n / rep1) * a.H * a.W * a.C + (b ? (double)(n / rep2) * a.H * a.W * b->C : 0.0) +
                        (double)c.Cout * c.ntaps * c.Cin * (c.mode == NOPE_CONV_UP2P ? 4 : 1) + (double)n * Ho * Wo * c.Cout) * (double)es;
            ev.info = nope_conv_launch_info{0.0, ev.flops, ev.bytes, conv_kernel_kind(net->dt, ca), c.mode, c.ntaps, c.Cin, c.Cout, a.H, a.W, n,
                                            net->dt != NOPE_BF16X3 ? 1 : conv_takes_x2(net->dt, ca) ? 2 : 3,
                                            conv_is_posmajor(net->dt, ca) ? 1 : 0};
            hipEventRecord(ev.a, s);
            chk(launch_conv(net->dt, ca, s));
            hipEventRecord(ev.b, s);
            net->evs.push_back(ev);
        } else {
            chk(launch_conv(net->dt, ca, s));
        }
    }
    // y = act(GN(x)) [+emb] [+resid]; x holds n_x = nhyp / x_rep samples; `colstats` != null: statistics were
    // produced by the conv epilogue and only need folding.
    // proj (with proj_out): GroupNorm + SiLU + residual + this 1x1 conv in one pass (launch_gn_apply_proj: y is not written) when the arguments
    // qualify; returns whether it did -- the caller launches the conv itself otherwise
    bool gn(const Norm& nm, int G, const void* x, int x_rep, void* y, int HW, int act, int emb_off, const void* resid,
            int resid_rep, const Stats& st = Stats(), float* out_stats = nullptr, const Conv* proj = nullptr, void* proj_out = nullptr, int proj_out_dt = NOPE_F32) {
        if (!live()) return false;
        const int nx = nhyp / x_rep;
        int nch = 1;
        GnApplyArgs ga;
        if (st.cs) {
            // Every gn_apply workgroup folds its sample's column statistics itself (st.blocks * C * 8 bytes out of L2 per workgroup; a
            // separate fold launch costs ~7 us + a kernel boundary).  Round 2 measured the inline fold +0.25 ms per 512-hypothesis step
            // (one thread per group then); with the wave-wide group sums of round 4 it is -0.02 .. -0.05 ms there and -0.2 ms at 64
            // hypotheses (profiles/r04o_fold_inline_ab.txt): always on.  NOPE_GN_FOLD_INLINE = most re-read bytes per launch (0 = never).
            const long long fold_inline_max = NOPE_ENV_LL("NOPE_GN_FOLD_INLINE", 1ll << 50);
            const long long refold = (long long)nhyp * gn_apply_blocks(HW, nm.C, net->sdt, nhyp) * st.blocks * nm.C * 8;
            if (refold <= fold_inline_max) { ga.colstats = st.cs; ga.stat_blocks = st.blocks; }
            else chk(launch_gn_fold(st.cs, gn_partial, nx, st.blocks, nm.C, G, s));
        } else {
            nch = gn_stats_chunks(HW, nm.C, net->sdt);
            chk(launch_gn_stats(net->sdt, x, gn_partial, nx, HW, nm.C, G, nch, s));
        }
        ga.x = x; ga.y = y; ga.partial = gn_partial; ga.nchunk = nch; ga.gamma = nm.gamma; ga.beta = nm.beta;
        ga.nhyp = nhyp; ga.HW = HW; ga.C = nm.C; ga.G = G; ga.act = act;
        if (emb_off >= 0) { ga.emb = emb_all + emb_off; ga.emb_stride = net->emb_total; }
        ga.resid = resid; ga.x_rep = x_rep; ga.resid_rep = resid_rep; ga.out_stats = out_stats;
        if (tracking()) { const int sl = x2.produce(y); if (sl >= 0) ga.amax_out = x2.slot_ptr(sl); }
        ga.fast_silu = net->dt != NOPE_F32 ? 1 : 0;      // (f32 storage of the split-precision modes: hardware exp / rcp; the f32 mode keeps expf and the division)
        if (proj && proj_out && net->final_w_raw) {
            ga.proj_w = net->final_w_raw; ga.proj_b = proj->bias; ga.proj_cout = proj->Cout; ga.proj_out = proj_out; ga.proj_out_dt = proj_out_dt;
            if (gn_apply_proj_ok(net->sdt, ga)) {
                if (tracking()) x2.overwritten(y);        // (y keeps whatever it held: no maximum recorded for it)
                chk(launch_gn_apply_proj(net->sdt, ga, s));
                return true;
            }
        }
        chk(launch_gn_apply(net->sdt, ga, s));
        return false;
    }

    // ResnetBlock, model_utils.py:271-279.  `a` may be shared by a.rep hypotheses (rep > 1 only
    // for the very first block, where b == nullptr).
    // `next_is_attention`: also emit the GroupNorm(1) partials of the block's output into pn_partial.
    // proj / proj_out: the 1x1 conv that is the ONLY reader of the block's output (the U-Net's tail), fused into the block's last pass where gn() can;
    // returns whether it was (then `out` holds the un-normalised conv output and must not be read)
    bool resnet(const Res& R, const Act& a, const Act* b, bool use_emb, void* out, bool next_is_attention = false, const Conv* proj = nullptr,
                void* proj_out = nullptr, int proj_out_dt = NOPE_F32) {
        const int HW = a.H * a.W, G = net->cfg.groups;
        const size_t M = (size_t)nhyp * HW;
        const size_t mark = ar.off;
        void* t1 = alloc_act(M * R.c1.Cout);
        const int emb_off = use_emb ? R.emb_off : -1;
        if (a.rep > 1 && !b) {
            // pose-independent prefix: conv + GN statistics once per reference sample
            const int ns = nhyp / a.rep;
            void* t1s = alloc_act((size_t)ns * HW * R.c1.Cout);
            Stats cs;
            conv(R.c1, a, nullptr, t1s, a.H, a.W, ns, 1, 1, nullptr, 0, NOPE_F32, &cs);
            gn(R.n1, G, t1s, a.rep, t1, HW, 1, emb_off, nullptr, 1, cs);
        } else {
            Stats cs;
            conv(R.c1, a, b, t1, a.H, a.W, nhyp, a.rep, b ? b->rep : 1, nullptr, 0, NOPE_F32, &cs);
            gn(R.n1, G, t1, 1, t1, HW, 1, emb_off, nullptr, 1, cs);
        }
        Act h{t1, R.c1.Cout, a.H, a.W, 1};
        Stats cs2;
        conv(R.c2, h, nullptr, out, a.H, a.W, nhyp, 1, 1, nullptr, 0, NOPE_F32, &cs2);
        const void* resid = a.p;
        int resid_rep = a.rep;
        if (R.has_res) {
            void* t3 = alloc_act(M * R.res.Cout);
            conv(R.res, a, b, t3, a.H, a.W, nhyp, a.rep, b ? b->rep : 1);
            resid = t3; resid_rep = 1;
        } else if (b) { chk(NOPE_ERR_ARG); }
        const bool fused = gn(R.n2, G, out, 1, out, HW, 1, -1, resid, resid_rep, cs2, next_is_attention ? pn_partial : nullptr, proj, proj_out, proj_out_dt);
        ar.off = mark;
        return fused;
    }

    // PreNorm folded into the qkv conv: finalize (mean, rstd) of x from the producer's partials, then
    // qkv = rstd * ((W gamma) x - mean * c1) + c0 in the conv epilogue -- x is read once, never re-written.
    void qkv_prenorm(const Conv& qkvw, const float* c0, const float* c1, const Act& x, void* qkv) {
        if (!live()) return;
        const int HW = x.H * x.W;
        chk(launch_gn_finalize(pn_partial, pn_ms, nhyp, gn_apply_blocks(HW, x.C, net->sdt, nhyp), (float)HW * (float)x.C, 1e-5f, s));
        conv(qkvw, x, nullptr, qkv, x.H, x.W, nhyp, 1, 1, nullptr, 0, NOPE_F32, nullptr, c0, c1);
    }

    // Residual(PreNorm(LinearAttention)), model_utils.py:198-204,226-234,393-418.  x must come from
    // resnet(..., next_is_attention = true).
    void linattn(const LinAttn& L, const Act& x, void* out) {
        const int HW = x.H * x.W, heads = net->cfg.heads, dh = net->cfg.dim_head;
        const size_t M = (size_t)nhyp * HW;
        const size_t mark = ar.off;
        void* y = alloc_act(M * x.C);
        void* qkv = alloc_act(M * 3 * heads * dh);
        void* a = alloc_act(M * heads * dh);
        qkv_prenorm(L.qkv, L.c0, L.c1, x, qkv);
        if (live()) { chk(launch_linattn(net->sdt, qkv, a, nhyp, HW, heads, dh, s)); x2.overwritten(a); }
        Act aa{a, heads * dh, x.H, x.W, 1};
        Stats cs;
        conv(L.out, aa, nullptr, y, x.H, x.W, nhyp, 1, 1, nullptr, 0, NOPE_F32, &cs);
        gn(L.post, 1, y, 1, out, HW, 0, -1, x.p, 1, cs);
        ar.off = mark;
    }

    // Residual(PreNorm(Attention)), model_utils.py:367-390
    void attn(const Attn& A, const Act& x, void* out) {
        const int HW = x.H * x.W, heads = net->cfg.heads, dh = net->cfg.dim_head;
        const size_t M = (size_t)nhyp * HW;
        const size_t mark = ar.off;
        void* qkv = alloc_act(M * 3 * heads * dh);
        void* a = alloc_act(M * heads * dh);
        qkv_prenorm(A.qkv, A.c0, A.c1, x, qkv);
        if (live()) { chk(launch_attn(net->sdt, qkv, a, nhyp, HW, heads, dh, s)); x2.overwritten(a); }
        Act aa{a, heads * dh, x.H, x.W, 1};
        conv(A.out, aa, nullptr, out, x.H, x.W, nhyp, 1, 1, /*resid=*/x.p, 0, NOPE_F32, nullptr, nullptr, nullptr, /*track_out=*/true);
        ar.off = mark;
    }
};

int run_forward(const nope_unet* net, const float* x, int n_src, int x_rep, const float* pose, int n_hyp, int H, int W,
                void* out, int out_dtype, void* ws, size_t ws_bytes, hipStream_t s, bool dry, size_t* peak) {
    const nope_unet_config& cfg = net->cfg;
    const int L = cfg.n_levels;
    Fwd f;
    f.net = net; f.s = s; f.nhyp = n_hyp; f.es = (size_t)dt_es(net->dt);
    f.ar.base = (unsigned char*)ws; f.ar.cap = ws_bytes; f.ar.dry = dry;
    f.x2.r = &net->x2r; f.x2.s = s; f.x2.on = net->x2 && net->x2r.active() && !dry;
    const int HW = H * W;
    const int* dims = net->dims;

    // ---- persistent buffers ------------------------------------------------------------------
    const int cin_k = net->init_conv.Cin;         // latent channels rounded up to 8
    void* x_in = f.alloc_act((size_t)n_src * HW * cin_k);
    void* x0 = f.alloc_act((size_t)n_src * HW * dims[0]);
    float* c0 = (float*)f.ar.alloc((size_t)n_hyp * net->classes * 4);
    float* c1 = (float*)f.ar.alloc((size_t)n_hyp * net->classes * 4);
    float* emb_all = (float*)f.ar.alloc((size_t)n_hyp * net->emb_total * 4);
    f.gn_partial = (float*)f.ar.alloc((size_t)n_hyp * 16 * (cfg.groups > 1 ? cfg.groups : 1) * 2 * 4);
    f.pn_partial = (float*)f.ar.alloc((size_t)n_hyp * 64 * 2 * 4);
    f.pn_ms = (float*)f.ar.alloc((size_t)n_hyp * 2 * 4);
    f.emb_all = emb_all;
    size_t cur_elems = 0;
    for (int l = 0; l <= L; ++l) {   // every tensor handed from one stage to the next
        const int rd = l < L ? l : L - 1;                 // down-path / bottleneck outputs
        const int ru = l > 0 ? l - 1 : 0;                 // up-path outputs (dims[l] at one level finer)
        const size_t e1 = (size_t)n_hyp * (HW >> (2 * rd)) * dims[l];
        const size_t e2 = l < L ? (size_t)n_hyp * (HW >> (2 * ru)) * dims[l] : 0;
        if (e1 > cur_elems) cur_elems = e1;
        if (e2 > cur_elems) cur_elems = e2;
    }
    void* curbuf[2] = {f.alloc_act(cur_elems), f.alloc_act(cur_elems)};
    void* hbuf[16];
    for (int l = 0; l < L; ++l) {
        const size_t e = (size_t)n_hyp * (HW >> (2 * l)) * dims[l];
        hbuf[2 * l] = f.alloc_act(e);
        hbuf[2 * l + 1] = f.alloc_act(e);
    }
    if (!dry && (!c0 || !c1 || !emb_all || !f.gn_partial || !f.pn_partial || !f.pn_ms)) f.chk(NOPE_ERR_WORKSPACE);
    if (f.err) return f.err;

    // ---- input + pose embedding ----------------------------------------------------------------
    if (f.live()) {
        f.chk(launch_nchw_to_nhwc(net->sdt, x, x_in, n_src, cin_k, HW, s, cfg.channels));
        if (cfg.pose_mlp_layers == 0) f.chk(launch_pos_emb(pose, c0, n_hyp, cfg.pose_dim, net->classes, s));   // u_net.py:73-76
        else f.chk(launch_linear_naive(pose, net->pose_w0, net->pose_b0, c0, n_hyp, net->classes, cfg.pose_dim, 0, net->classes, s));
        const float* c = c0;
        if (cfg.pose_mlp_layers == 2) {
            f.chk(launch_linear_naive(c0, net->pose_w2, net->pose_b2, c1, n_hyp, net->classes, net->classes, 2, net->classes, s));
            c = c1;
        }
        float* sc = (c == c0) ? c1 : c0;
        f.chk(launch_silu_f32(c, sc, (size_t)n_hyp * net->classes, s));
        ConvArgs ea;   // emb_all[n_hyp][emb_total] = SiLU(c) @ W_all^T + b_all, exact-f32 MFMA
        ea.src1 = sc; ea.C1 = net->classes; ea.w = net->emb_w; ea.bias = net->emb_b; ea.out = emb_all;
        ea.Cout = net->emb_total; ea.nhyp = n_hyp;
        f.chk(launch_conv(NOPE_F32, ea, s));
    }
    Act xin{x_in, cin_k, H, W, 1};
    {
        Fwd g = f;   // init_conv runs over the n_src reference samples only
        g.nhyp = n_src;
        g.conv(net->init_conv, xin, nullptr, x0, H, W, n_src, 1, 1);
        f.chk(g.err);
        f.x2 = g.x2;
    }
    Act r0{x0, dims[0], H, W, x_rep};
    Act cur = r0;
    int slot = 0;

    // ---- down path ---------------------------------------------------------------------------------
    for (int l = 0; l < L; ++l) {
        const Level& D = net->downs[l];
        Act h1{hbuf[2 * l], dims[l], cur.H, cur.W, 1};
        Act h2{hbuf[2 * l + 1], dims[l], cur.H, cur.W, 1};
        f.resnet(D.r0, cur, nullptr, true, h1.p);
        {
            const size_t mark = f.ar.off;
            Act t{f.alloc_act((size_t)n_hyp * cur.H * cur.W * dims[l]), dims[l], cur.H, cur.W, 1};
            f.resnet(D.r1, h1, nullptr, true, t.p, /*next_is_attention=*/true);
            f.linattn(D.attn, t, h2.p);
            f.ar.off = mark;
        }
        Act nxt{curbuf[slot], dims[l + 1], cur.H, cur.W, 1};
        if (l < L - 1) { nxt.H = cur.H / 2; nxt.W = cur.W / 2; }
        f.conv(D.resample, h2, nullptr, nxt.p, nxt.H, nxt.W, n_hyp, 1, 1, nullptr, 0, NOPE_F32, nullptr, nullptr, nullptr, /*track_out=*/true);
        cur = nxt;
        slot ^= 1;
    }
    // ---- bottleneck, applied twice with the same weights (u_net.py:177-183) ---------------------------
    for (int it = 0; it < 2; ++it) {
        const size_t mark = f.ar.off;
        const size_t e = (size_t)n_hyp * cur.H * cur.W * cur.C;
        Act a{f.alloc_act(e), cur.C, cur.H, cur.W, 1};
        Act b{f.alloc_act(e), cur.C, cur.H, cur.W, 1};
        f.resnet(net->mid1, cur, nullptr, true, a.p, /*next_is_attention=*/true);
        f.attn(net->mid_attn, a, b.p);
        Act c{curbuf[slot], cur.C, cur.H, cur.W, 1};
        f.resnet(net->mid2, b, nullptr, true, c.p);
        f.ar.off = mark;
        cur = c;
        slot ^= 1;
    }
    // ---- up path --------------------------------------------------------------------------------------
    for (int l = 0; l < L; ++l) {
        const Level& U = net->ups[l];
        const int r = L - 1 - l;
        Act h2{hbuf[2 * r + 1], dims[r], cur.H, cur.W, 1};
        Act h1{hbuf[2 * r], dims[r], cur.H, cur.W, 1};
        const size_t mark = f.ar.off;
        const size_t e = (size_t)n_hyp * cur.H * cur.W * dims[r + 1];
        Act a{f.alloc_act(e), dims[r + 1], cur.H, cur.W, 1};
        Act b{f.alloc_act(e), dims[r + 1], cur.H, cur.W, 1};
        f.resnet(U.r0, cur, &h2, true, a.p);
        f.resnet(U.r1, a, &h1, true, b.p, /*next_is_attention=*/true);
        f.linattn(U.attn, b, a.p);
        Act nxt{curbuf[slot], dims[r], cur.H, cur.W, 1};
        if (l < L - 1) { nxt.H = cur.H * 2; nxt.W = cur.W * 2; }
        f.conv(U.resample, a, nullptr, nxt.p, nxt.H, nxt.W, n_hyp, 1, 1, nullptr, 0, NOPE_F32, nullptr, nullptr, nullptr, /*track_out=*/true);
        f.ar.off = mark;
        cur = nxt;
        slot ^= 1;
    }
    // ---- head --------------------------------------------------------------------------------------------
    {
        const size_t mark = f.ar.off;
        const size_t e = (size_t)n_hyp * HW * cfg.u_net_dim;
        Act a{f.alloc_act(e), cfg.u_net_dim, H, W, 1};
        Act b{f.alloc_act(e), cfg.u_net_dim, H, W, 1};
        f.resnet(net->final_res, cur, &r0, true, a.p);
        // final_conv = ResnetBlock -> Conv2d(dim, out_dim, 1) (u_net.py:154-157,197): the 1x1 conv rides in the block's last GroupNorm pass in the
        // split-precision modes (NOPE_FINAL_FUSED=0: its own launch, as in the other modes)
        if (!f.resnet(net->final_conv0, a, nullptr, false, b.p, false, &net->final_conv1, out, out_dtype))
            f.conv(net->final_conv1, b, nullptr, out, H, W, n_hyp, 1, 1, nullptr, /*out_nchw=*/1, out_dtype);
        f.ar.off = mark;
    }
    if (f.tracking())      // the forward's verdict and, if a layer left its window, NaNs over its output -- device side, no synchronisation (x2_range.h)
        f.chk(f.x2.finish(out, (size_t)n_hyp * cfg.out_dim * HW * (size_t)(out_dtype == NOPE_F32 ? 4 : 2), out_dtype));
    if (peak) *peak = f.ar.peak;
    return f.err;
}

}  // namespace

extern "C" {

int nope_unet_create(const nope_unet_config* cfg, const nope_tensor_desc* tensors, int n_tensors, nope_stream_t stream,
                     nope_unet** out) {
    if (!cfg || !tensors || !out || n_tensors <= 0) return NOPE_ERR_ARG;
    if (cfg->n_levels < 1 || cfg->n_levels > 8 || cfg->groups < 1 || cfg->heads < 1 || cfg->dim_head != 32) return NOPE_ERR_UNSUPPORTED;
    if (!dt_is_compute(cfg->compute_dtype)) return NOPE_ERR_UNSUPPORTED;
    if (cfg->pose_mlp_layers < 0 || cfg->pose_mlp_layers > 2) return NOPE_ERR_UNSUPPORTED;   // 0 = "posEncoding" (no parameters)
    if (cfg->pose_mlp_layers == 0 && (cfg->pose_dim < 1 || (cfg->u_net_dim * 4) % (2 * cfg->pose_dim) || cfg->u_net_dim * 4 / cfg->pose_dim < 4)) return NOPE_ERR_UNSUPPORTED;
    if (cfg->u_net_dim % 8 || cfg->channels < 1 || cfg->out_dim < 1 || cfg->u_net_dim % cfg->groups) return NOPE_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    nope_unet* net = new nope_unet();
    net->cfg = *cfg;
    net->x2 = cfg->compute_dtype == NOPE_F16X2;
    net->dt = dt_base(cfg->compute_dtype);
    net->sdt = dt_storage(net->dt);
    const int L = cfg->n_levels;
    net->dims[0] = cfg->u_net_dim;
    for (int l = 0; l < L; ++l) net->dims[l + 1] = cfg->u_net_dim * cfg->dim_mults[l];
    net->classes = cfg->u_net_dim * 4;
    net->graph_max = getenv("NOPE_UNET_GRAPH") ? atoll(getenv("NOPE_UNET_GRAPH")) : 0;
    const int* dims = net->dims;
    const int HD = cfg->heads * cfg->dim_head;

    Loader ld;
    ld.net = net; ld.s = s;
    for (int i = 0; i < n_tensors; ++i)
        if (tensors[i].name) ld.tab[tensors[i].name] = &tensors[i];
    std::vector<std::pair<std::string, int>> embs;

    if (cfg->pose_mlp_layers >= 1) {
        net->pose_w0 = ld.copy_f32("pose_mlp.0.weight", {net->classes, cfg->pose_dim});
        net->pose_b0 = ld.copy_f32("pose_mlp.0.bias", {net->classes});
    }
    if (cfg->pose_mlp_layers == 2) {
        net->pose_w2 = ld.copy_f32("pose_mlp.2.weight", {net->classes, net->classes});
        net->pose_b2 = ld.copy_f32("pose_mlp.2.bias", {net->classes});
    }
    net->init_conv = ld.conv("init_conv.", cfg->channels, dims[0], 3, NOPE_CONV_PLAIN, true, nullptr, (cfg->channels + 7) / 8 * 8);
    auto linattn = [&](const std::string& p, int C) {
        LinAttn a;
        ld.prenorm_qkv(p, C, 3 * HD, a.pre, a.qkv, a.c0, a.c1);
        a.out = ld.conv(p + "fn.fn.to_out.0.", HD, C, 1, NOPE_CONV_PLAIN, true);
        a.post = ld.norm(p + "fn.fn.to_out.1.", C);
        return a;
    };
    for (int l = 0; l < L; ++l) {
        const std::string p = "downs." + std::to_string(l) + ".";
        Level& D = net->downs[l];
        D.r0 = ld.res(p + "0.", dims[l], dims[l], true, embs);
        D.r1 = ld.res(p + "1.", dims[l], dims[l], true, embs);
        D.attn = linattn(p + "2.", dims[l]);
        if (l < L - 1 && cfg->soft_up_down) D.resample = ld.conv(p + "3.", dims[l], dims[l + 1], 4, NOPE_CONV_STRIDE2, true);   // Conv2d(4, 2, 1)
        else if (l < L - 1) D.resample = ld.conv(p + "3.1.", dims[l], dims[l + 1], 1, NOPE_CONV_DOWN2, true);
        else D.resample = ld.conv(p + "3.", dims[l], dims[l + 1], 3, NOPE_CONV_PLAIN, true);
    }
    net->mid1 = ld.res("mid_block1.", dims[L], dims[L], true, embs);
    ld.prenorm_qkv("mid_attn.", dims[L], 3 * HD, net->mid_attn.pre, net->mid_attn.qkv, net->mid_attn.c0, net->mid_attn.c1);
    net->mid_attn.out = ld.conv("mid_attn.fn.fn.to_out.", HD, dims[L], 1, NOPE_CONV_PLAIN, true);
    net->mid2 = ld.res("mid_block2.", dims[L], dims[L], true, embs);
    for (int l = 0; l < L; ++l) {
        const int r = L - 1 - l;
        const std::string p = "ups." + std::to_string(l) + ".";
        Level& U = net->ups[l];
        U.r0 = ld.res(p + "0.", dims[r + 1] + dims[r], dims[r + 1], true, embs);
        U.r1 = ld.res(p + "1.", dims[r + 1] + dims[r], dims[r + 1], true, embs);
        U.attn = linattn(p + "2.", dims[r + 1]);
        if (l < L - 1 && cfg->soft_up_down) U.resample = ld.conv(p + "3.", dims[r + 1], dims[r], 4, NOPE_CONV_UP2P, true);   // ConvTranspose2d(4, 2, 1)
        else if (l < L - 1) U.resample = ld.conv(p + "3.1.", dims[r + 1], dims[r], 3, NOPE_CONV_UP2P, true);   // 4 phase 2x2 convs
        else U.resample = ld.conv(p + "3.", dims[r + 1], dims[r], 3, NOPE_CONV_PLAIN, true);
    }
    net->final_res = ld.res("final_res_block.", cfg->u_net_dim * 2, cfg->u_net_dim, true, embs);
    net->final_conv0 = ld.res("final_conv.0.", cfg->u_net_dim, cfg->u_net_dim, false, embs);
    net->final_conv1 = ld.conv("final_conv.1.", cfg->u_net_dim, cfg->out_dim, 1, NOPE_CONV_PLAIN, true);
    net->final_w_raw = ld.copy_f32("final_conv.1.weight", {cfg->out_dim, cfg->u_net_dim, 1, 1});
    if (dims[0] != cfg->u_net_dim) ld.fail("init_dim");

    // row-concatenated embedding linears
    net->emb_w = (float*)ld.dmalloc((size_t)net->emb_total * net->classes * 4);
    net->emb_b = (float*)ld.dmalloc((size_t)net->emb_total * 4);
    int off = 0;
    for (auto& e : embs) {
        const nope_tensor_desc* w = ld.get(e.first + "weight", {e.second, net->classes});
        const nope_tensor_desc* b = ld.get(e.first + "bias", {e.second});
        if (w && b && net->emb_w && net->emb_b) {
            ld.copy_d2d(net->emb_w + (size_t)off * net->classes, w->data, (size_t)e.second * net->classes * 4);
            ld.copy_d2d(net->emb_b + off, b->data, (size_t)e.second * 4);
        }
        off += e.second;
    }
    if (ld.err == NOPE_OK) { const int e = net->x2r.init([&](size_t bytes) { return ld.dmalloc(bytes); }, s); if (e) ld.err = e; }
    if (ld.err == NOPE_OK && hipStreamSynchronize(s) != hipSuccess) ld.err = NOPE_ERR_LAUNCH;
    if (ld.err != NOPE_OK) {
        if (!ld.missing.empty()) fprintf(stderr, "nope_unet_create: missing or mis-shaped tensor '%s'\n", ld.missing.c_str());
        nope_unet_destroy(net);
        return ld.err;
    }
    *out = net;
    return NOPE_OK;
}

int nope_unet_profile(nope_unet* net, int enable) {
    if (!net) return NOPE_ERR_ARG;
    for (auto& e : net->evs) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    net->evs.clear();
    net->profile = enable != 0;
    return NOPE_OK;
}

int nope_unet_profile_read(nope_unet* net, int* n_launches, double* total_ms, double* total_flops, double* total_bytes) {
    if (!net || !n_launches || !total_ms || !total_flops || !total_bytes) return NOPE_ERR_ARG;
    double ms = 0.0, fl = 0.0, by = 0.0;
    for (auto& e : net->evs) {
        if (hipEventSynchronize(e.b) != hipSuccess) return NOPE_ERR_LAUNCH;
        float t = 0.f;
        if (hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) return NOPE_ERR_LAUNCH;
        ms += t; fl += e.flops; by += e.bytes;
    }
    *n_launches = (int)net->evs.size(); *total_ms = ms; *total_flops = fl; *total_bytes = by;
    return NOPE_OK;
}

int nope_unet_profile_launches(nope_unet* net, nope_conv_launch_info* out, int max, int* n) {
    if (!net || !n || (max > 0 && !out)) return NOPE_ERR_ARG;
    int i = 0;
    for (auto& e : net->evs) {
        if (i < max) {
            if (hipEventSynchronize(e.b) != hipSuccess) return NOPE_ERR_LAUNCH;
            float t = 0.f;
            if (hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) return NOPE_ERR_LAUNCH;
            out[i] = e.info;
            out[i].ms = t;
        }
        ++i;
    }
    *n = i;
    return NOPE_OK;
}

// ---- NOPE_F16X2 activation ranges ---------------------------------------------------------------------------------------------------------
// The f16 + MX-fp8 tile forms its A operands from a' = a * 2^-t (t per layer, nope_common.h: kX2*): f16(a') (saturates at 65504),
// e4m3(a'_lo * 2^9) and e4m3(a' * 2^-2) (saturates at |a'| = 1792, below 2^-4 it runs out of significant bits).  A launch whose LARGEST |a'|
// lies above 1792 or below 2^-4 computed its cross terms from saturated / subnormal operands: plain-f16 accuracy instead of ~2^-15 per
// product.  Every forward is judged on the device (x2_verdict_kernel) and an out-of-range forward's output is NaN.  The poll reads the verdicts
// that have arrived in mapped host memory since the previous poll -- no synchronisation -- and re-centres t where a layer was out of, or within
// a binade or two of the end of, its window (max |a'| in [256, 512): three binades of headroom, full accuracy down to 2^-13 of the maximum);
// the new shifts travel on `stream`, ordered before whatever is enqueued on it next.
int nope_unet_x2_poll(nope_unet* net, nope_stream_t stream, int* n_out_of_range, int* n_adjusted, float* max_abs) {
    if (n_out_of_range) *n_out_of_range = 0;
    if (n_adjusted) *n_adjusted = 0;
    if (max_abs) *max_abs = 0.f;
    if (!net) return NOPE_ERR_ARG;
    if (!net->x2) return NOPE_OK;
    // (a cached hipGraph replays the same kernels and pointers; the shifts live in device memory: nothing to rebuild)
    return net->x2r.poll((hipStream_t)stream, n_out_of_range, n_adjusted, max_abs);
}

// ... behind a synchronisation of `stream`: the verdicts of every forward issued on it so far.  NOPE_ERR_RANGE: at least one of them was out of
// range (its output is NaN); the shifts are re-centred -- issue it again.
int nope_unet_x2_range_check(nope_unet* net, nope_stream_t stream, int* n_out_of_range, int* n_adjusted, float* max_abs) {
    if (!net) return NOPE_ERR_ARG;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return NOPE_ERR_LAUNCH;
    return nope_unet_x2_poll(net, stream, n_out_of_range, n_adjusted, max_abs);
}

int nope_unet_x2_enable(nope_unet* net, int on) {
    if (!net) return NOPE_ERR_ARG;
    std::lock_guard<std::mutex> lock(net->graph_mu);
    if (net->x2r.off != (on == 0)) {           // the launch plan changes: cached graphs are stale
        for (const UGraph& g : net->graphs) hipGraphExecDestroy(g.exec);
        net->graphs.clear();
    }
    net->x2r.off = on == 0;
    return NOPE_OK;
}

int nope_unet_x2_shifts(const nope_unet* net, int* shifts, int max, int* n) {
    if (!net || !n || (max > 0 && !shifts)) return NOPE_ERR_ARG;
    return net->x2r.shifts(shifts, max, n);
}

int nope_unet_graph_limit(nope_unet* net, long long max_hyp_pixels) {
    if (!net || max_hyp_pixels < 0) return NOPE_ERR_ARG;
    std::lock_guard<std::mutex> lock(net->graph_mu);
    net->graph_max = max_hyp_pixels;
    return NOPE_OK;
}

int nope_unet_graph_replays(const nope_unet* net) {
    if (!net) return NOPE_ERR_ARG;
    std::lock_guard<std::mutex> lock(net->graph_mu);
    return net->graph_replays;
}

void nope_unet_destroy(nope_unet* net) {
    if (!net) return;
    for (const UGraph& g : net->graphs) hipGraphExecDestroy(g.exec);
    for (auto& e : net->evs) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (void* p : net->allocs) hipFree(p);
    net->x2r.destroy();
    delete net;
}

static int check_shape(const nope_unet* net, int n_hyp, int n_src, int x_rep, int H, int W) {
    if (!net || n_hyp <= 0 || n_src <= 0 || x_rep <= 0 || (long long)n_src * x_rep != n_hyp || H <= 0 || W <= 0) return NOPE_ERR_ARG;
    const int f = 1 << (net->cfg.n_levels - 1);
    if (H % f || W % f) return NOPE_ERR_UNSUPPORTED;
    if ((H / f) * (W / f) > 64) return NOPE_ERR_UNSUPPORTED;   // bottleneck attention tile
    return NOPE_OK;
}

// Workspace = [staged x | staged pose | staged output (largest element type) | activation arena]
static size_t unet_stage_bytes(const nope_unet* net, int n_hyp, int n_src, int H, int W, size_t& xb, size_t& pb, size_t& ob) {
    xb = align_up((size_t)n_src * net->cfg.channels * H * W * 4, 256);
    pb = align_up((size_t)n_hyp * net->cfg.pose_dim * 4, 256);
    ob = align_up((size_t)n_hyp * net->cfg.out_dim * H * W * 4, 256);
    return xb + pb + ob;
}

size_t nope_unet_workspace_bytes(const nope_unet* net, int n_hyp, int n_src, int H, int W) {
    if (!net || n_src <= 0 || n_hyp % n_src) return 0;
    if (check_shape(net, n_hyp, n_src, n_hyp / n_src, H, W) != NOPE_OK) return 0;
    size_t peak = 0;
    run_forward(net, nullptr, n_src, n_hyp / n_src, nullptr, n_hyp, H, W, nullptr, NOPE_F32, nullptr, 0, nullptr, true, &peak);
    size_t xb, pb, ob;
    return unet_stage_bytes(net, n_hyp, n_src, H, W, xb, pb, ob) + align_up(peak, 256) + 256;
}

int nope_unet_forward(const nope_unet* net, const float* x, int n_src, int x_rep, const float* pose, int n_hyp, int H, int W,
                      void* out, int out_dtype, void* workspace, size_t workspace_bytes, nope_stream_t stream) {
    int e = check_shape(net, n_hyp, n_src, x_rep, H, W);
    if (e) return e;
    if (!x || !pose || !out || !workspace) return NOPE_ERR_ARG;
    if (out_dtype != NOPE_F32 && out_dtype != NOPE_BF16 && out_dtype != NOPE_F16) return NOPE_ERR_UNSUPPORTED;
    unsigned char* base = (unsigned char*)(((uintptr_t)workspace + 255) / 256 * 256);
    const size_t lost = (size_t)(base - (unsigned char*)workspace);
    if (workspace_bytes < lost) return NOPE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    size_t xb, pb, ob;
    const size_t sb = unet_stage_bytes(net, n_hyp, n_src, H, W, xb, pb, ob);
    const size_t avail = workspace_bytes - lost;
    // Opt-in (net->graph_max > 0): batches of at most that many hypothesis-pixels replay a captured launch sequence; everything else
    // launches directly.
    // (NOPE_F16X2 with range tracking launches directly: the per-forward table of the verdict kernel is a host-to-device copy)
    const bool want_graph = net->graph_max > 0 && net->graphs_ok && !net->profile && avail > sb && (long long)n_hyp * H * W <= net->graph_max &&
                            !(net->x2 && net->x2r.active());
    if (net->x2 && net->x2r.active()) (void)net->x2r.poll((hipStream_t)stream, nullptr, nullptr, nullptr);      // verdicts that have arrived: re-centre first
    if (!want_graph)
        return run_forward(net, x, n_src, x_rep, pose, n_hyp, H, W, out, out_dtype, base, avail, s, false, nullptr);
    std::lock_guard<std::mutex> lock(net->graph_mu);

    float* x_s = (float*)base;
    float* pose_s = (float*)(base + xb);
    void* out_s = base + xb + pb;
    unsigned char* arena = base + sb;
    const size_t arena_bytes = avail - sb;
    const UGraph* hit = nullptr;
    for (const UGraph& g : net->graphs)
        if (g.ws == workspace && g.ws_bytes == workspace_bytes && g.n_hyp == n_hyp && g.n_src == n_src && g.H == H && g.W == W && g.out_dt == out_dtype) { hit = &g; break; }
    if (!hit) {
        {   // dry pass: fail on a too-small arena BEFORE a capture is open
            size_t peak = 0;
            e = run_forward(net, nullptr, n_src, x_rep, nullptr, n_hyp, H, W, nullptr, out_dtype, nullptr, 0, nullptr, true, &peak);
            if (e) return e;
            if (align_up(peak, 256) > arena_bytes) return NOPE_ERR_WORKSPACE;
        }
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            net->graphs_ok = false;
            return run_forward(net, x, n_src, x_rep, pose, n_hyp, H, W, out, out_dtype, base, avail, s, false, nullptr);
        }
        e = run_forward(net, x_s, n_src, x_rep, pose_s, n_hyp, H, W, out_s, out_dtype, arena, arena_bytes, s, false, nullptr);
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(s, &graph);
        hipGraphExec_t exec = nullptr;
        if (e != NOPE_OK || ce != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) hipGraphDestroy(graph);
            (void)hipGetLastError();
            net->graphs_ok = false;
            return e != NOPE_OK ? e : run_forward(net, x, n_src, x_rep, pose, n_hyp, H, W, out, out_dtype, base, avail, s, false, nullptr);
        }
        hipGraphDestroy(graph);
        if (net->graphs.size() >= 16) { hipGraphExecDestroy(net->graphs.front().exec); net->graphs.erase(net->graphs.begin()); }
        net->graphs.push_back(UGraph{workspace, workspace_bytes, n_hyp, n_src, H, W, out_dtype, exec});
        hit = &net->graphs.back();
    }
    const size_t out_bytes = (size_t)n_hyp * net->cfg.out_dim * H * W * (size_t)(out_dtype == NOPE_F32 ? 4 : 2);
    if (hipMemcpyAsync(x_s, x, (size_t)n_src * net->cfg.channels * H * W * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return NOPE_ERR_LAUNCH;
    if (hipMemcpyAsync(pose_s, pose, (size_t)n_hyp * net->cfg.pose_dim * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return NOPE_ERR_LAUNCH;
    if (hipGraphLaunch(hit->exec, s) != hipSuccess) return NOPE_ERR_LAUNCH;
    ++net->graph_replays;
    if (hipMemcpyAsync(out, out_s, out_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return NOPE_ERR_LAUNCH;
    return NOPE_OK;
}

}  // extern "C"
