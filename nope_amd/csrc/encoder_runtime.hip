// Host-side runtime of the template encoder:
//   FeatureExtractor.encode_image, src/model/encoder/template.py:47-53
//   = projector(backbone(image)): ResNet-50 trunk resnet.py:92-152 (conv1 7x7/2 + bn + relu, no max-pool,
//     Bottleneck layers [3,4,6,3] with strides (1,2,2,1), resnet.py:55-90,102-105,179-185) -> /8, 2048 channels ->
//     ReLU, 1x1(2048->256), ReLU, 1x1(256->descriptor_size) (template.py:33-38).
//
// MI355X execution: NHWC activations; every eval-mode BatchNorm is folded into the preceding conv at
// create time (scale into the packed weights, shift as the conv bias); ReLU and the residual add run in the
// conv epilogue, so a Bottleneck is 3 (4 with a projection shortcut) launches of the implicit-GEMM kernel and
// nothing else touches HBM.  conv1 (3 input channels) is a small direct kernel.  With one or two images the deep
// layers have only 8-32 output tiles for 256 CUs, so those launches are split along K (deterministic: f32 partials +
// a fixed-order reduce that also applies bias / residual / ReLU).  Activations live in five ping-pong buffers of a
// caller-provided workspace: no allocation, no host sync, one stream.
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "nope_common.h"

using namespace nope;

namespace {

struct EConv { void* w = nullptr; float* bias = nullptr; int Cin = 0, Cout = 0, ntaps = 1, mode = NOPE_CONV_PLAIN; };
struct Bottleneck { EConv c1, c2, c3, ds; bool has_ds = false; };

constexpr int LAYERS[4] = {3, 4, 6, 3};       // resnet.py:179-185 (resnet50)
constexpr int STRIDES[4] = {1, 2, 2, 1};      // resnet.py:102-105
constexpr int FEATURES = 64;

}  // namespace

// One captured launch sequence: valid for exactly this (workspace, batch, size); image and output are staged through
// the workspace so that the caller's pointers stay out of the graph.
struct EGraph { void* ws; size_t ws_bytes; int n_img, H, W; hipGraphExec_t exec; };

struct nope_encoder {
    nope_encoder_config cfg;
    int dt = NOPE_F32;
    std::vector<void*> allocs;
    mutable std::vector<EGraph> graphs;      // hipGraph cache (the ~85 launches of a pass are 5-15 us each at 1-2 images)
    mutable bool graphs_ok = true;           // cleared when capture is unavailable: direct launches from then on
    mutable std::mutex graph_mu;             // the cache is edited from a const entry point while ctypes callers have released the GIL
    float* stem_w = nullptr;      // [147][64] f32, bn1 scale folded
    float* stem_shift = nullptr;  // [64]
    std::vector<Bottleneck> blocks;
    std::vector<int> block_stride;
    EConv proj0, proj1;
};

namespace {

struct ELoader {
    nope_encoder* enc;
    hipStream_t s;
    std::map<std::string, const nope_tensor_desc*> tab;
    int err = NOPE_OK;
    std::string missing;

    void fail(const std::string& n) { if (err == NOPE_OK) { err = NOPE_ERR_WEIGHT; missing = n; } }
    const nope_tensor_desc* get(const std::string& name, std::initializer_list<int64_t> shape) {
        auto it = tab.find(name);
        if (it == tab.end() || !it->second->data) { fail(name); return nullptr; }
        const nope_tensor_desc* d = it->second;
        if (d->ndim != (int)shape.size()) { fail(name); return nullptr; }
        int i = 0;
        for (int64_t v : shape) if (d->shape[i++] != v) { fail(name); return nullptr; }
        return d;
    }
    void* dmalloc(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { if (err == NOPE_OK) err = NOPE_ERR_ALLOC; return nullptr; }
        enc->allocs.push_back(p);
        return p;
    }
    void chk(int e) { if (e != NOPE_OK && err == NOPE_OK) err = e; }
    // (scale, shift) of the eval-mode BatchNorm2d `pfx` over C channels
    bool bn(const std::string& pfx, int C, float*& scale, float*& shift) {
        const nope_tensor_desc* g = get(pfx + "weight", {C});
        const nope_tensor_desc* b = get(pfx + "bias", {C});
        const nope_tensor_desc* m = get(pfx + "running_mean", {C});
        const nope_tensor_desc* v = get(pfx + "running_var", {C});
        scale = (float*)dmalloc((size_t)C * 4);
        shift = (float*)dmalloc((size_t)C * 4);
        if (!g || !b || !m || !v || !scale || !shift) return false;
        chk(launch_bn_fold(g->data, b->data, m->data, v->data, enc->cfg.bn_eps, scale, shift, C, s));
        return true;
    }
    // conv `wname` (bias-free in the reference) followed by BatchNorm `bnpfx` -> packed weights * scale, bias = shift
    EConv conv_bn(const std::string& wname, const std::string& bnpfx, int Cin, int Cout, int ksz, int stride) {
        EConv c;
        c.Cin = Cin; c.Cout = Cout; c.ntaps = ksz * ksz; c.mode = stride == 2 ? NOPE_CONV_STRIDE2 : NOPE_CONV_PLAIN;
        const nope_tensor_desc* d = get(wname, {Cout, Cin, ksz, ksz});
        float* scale = nullptr;
        if (!bnpfx.empty()) { if (!bn(bnpfx, Cout, scale, c.bias)) return c; }
        if (d) {
            const size_t es = (size_t)dt_es(enc->dt);
            c.w = dmalloc((size_t)Cout * c.ntaps * Cin * es);
            if (c.w) chk(launch_pack_conv_w(enc->dt, d->data, c.w, Cout, Cin, c.ntaps, NOPE_CONV_PLAIN, s, nullptr, scale));
        }
        return c;
    }
};

struct EArena {
    unsigned char* base = nullptr;
    size_t cap = 0, off = 0;
    bool dry = false;
    void* alloc(size_t bytes) {
        const size_t o = align_up(off, 256);
        off = o + bytes;
        if (dry) return (void*)(uintptr_t)(0x1000 + o);
        if (off > cap) return nullptr;
        return base + o;
    }
};

// Buffer sizes (elements per image) of the five activation buffers for an H x W image.
struct Plan { size_t x = 0, t1 = 0, t2 = 0; };
Plan plan_for(int H, int W) {
    Plan p;
    int h = H / 2, w = W / 2, cin = FEATURES;
    p.x = (size_t)h * w * cin;
    for (int l = 0; l < 4; ++l) {
        const int planes = FEATURES << l;
        for (int b = 0; b < LAYERS[l]; ++b) {
            const int st = b == 0 ? STRIDES[l] : 1;
            const size_t e1 = (size_t)h * w * planes;                 // conv1 output at the input resolution
            const int ho = h / st, wo = w / st;
            const size_t e2 = (size_t)ho * wo * planes;
            const size_t e3 = (size_t)ho * wo * planes * 4;
            if (e1 > p.t1) p.t1 = e1;
            if (e2 > p.t2) p.t2 = e2;
            if (e3 > p.x) p.x = e3;
            h = ho; w = wo; cin = planes * 4;
        }
    }
    if ((size_t)h * w * 256 > p.t1) p.t1 = (size_t)h * w * 256;       // projector hidden
    return p;
}

int run_encoder(const nope_encoder* enc, const float* image, int n_img, int H, int W, float* out, EArena& ar, hipStream_t s) {
    const size_t es = (size_t)dt_es(enc->dt);
    const Plan pl = plan_for(H, W);
    void* X[3] = {nullptr, nullptr, nullptr};
    void *T1 = nullptr, *T2 = nullptr, *SK = nullptr;
    size_t sk_bytes = 0;
    bool measure = true;

    // One conv launch; in the measuring pass it only records the split-K scratch the launch would like.
    auto conv = [&](const EConv& c, const void* src, int Hs, int Ws, void* dst, int act, const void* resid, int out_nchw) -> int {
        ConvArgs a;
        a.src1 = src; a.C1 = c.Cin; a.Hs = Hs; a.Ws = Ws;
        a.Ho = c.mode == NOPE_CONV_STRIDE2 ? Hs / 2 : Hs; a.Wo = c.mode == NOPE_CONV_STRIDE2 ? Ws / 2 : Ws;
        a.mode = c.mode; a.ntaps = c.ntaps; a.w = c.w; a.bias = c.bias; a.resid = resid; a.out = dst; a.Cout = c.Cout;
        a.nhyp = n_img; a.act = act; a.out_nchw = out_nchw; a.out_dt = NOPE_F32;
        if (measure) {
            const size_t need = (size_t)conv_splitk_factor(enc->dt, a) * n_img * a.Ho * a.Wo * c.Cout * 4;
            if (need > sk_bytes) sk_bytes = need;
            return NOPE_OK;
        }
        a.splitk_ws = SK; a.splitk_bytes = sk_bytes;
        return launch_conv(enc->dt, a, s);
    };
    auto walk = [&]() -> int {
        int e = NOPE_OK;
        if (!measure && (e = launch_stem_conv(dt_storage(enc->dt), image, enc->stem_w, enc->stem_shift, X[0], n_img, H, W, s))) return e;   // resnet.py:136-138
        int cur = 0, h = H / 2, w = W / 2;
        for (size_t i = 0; i < enc->blocks.size(); ++i) {      // Bottleneck.forward, resnet.py:70-90
            const Bottleneck& b = enc->blocks[i];
            const int st = enc->block_stride[i];
            const int ho = h / st, wo = w / st;
            const void* identity = X[cur];
            const int nxt = (cur + 1) % 3;
            if (b.has_ds) {
                const int dsb = (cur + 2) % 3;
                if ((e = conv(b.ds, X[cur], h, w, X[dsb], 0, nullptr, 0))) return e;
                identity = X[dsb];
            }
            if ((e = conv(b.c1, X[cur], h, w, T1, 1, nullptr, 0))) return e;
            if ((e = conv(b.c2, T1, h, w, T2, 1, nullptr, 0))) return e;
            if ((e = conv(b.c3, T2, ho, wo, X[nxt], 1, identity, 0))) return e;       // relu(bn3(conv3) + identity)
            cur = nxt; h = ho; w = wo;
        }
        // projector: ReLU (a no-op on the ReLU output above), 1x1 -> ReLU -> 1x1, NCHW f32 out (template.py:33-38)
        if ((e = conv(enc->proj0, X[cur], h, w, T1, 1, nullptr, 0))) return e;
        return conv(enc->proj1, T1, h, w, out, 0, nullptr, 1);
    };

    int e = walk();                       // measuring pass: no launches
    if (e) return e;
    for (int i = 0; i < 3; ++i) X[i] = ar.alloc(pl.x * n_img * es);
    T1 = ar.alloc(pl.t1 * n_img * es);
    T2 = ar.alloc(pl.t2 * n_img * es);
    SK = ar.alloc(sk_bytes);
    if (ar.dry) return NOPE_OK;
    if (!X[0] || !X[1] || !X[2] || !T1 || !T2 || !SK) return NOPE_ERR_WORKSPACE;
    measure = false;
    return walk();
}

}  // namespace

extern "C" {

int nope_encoder_create(const nope_encoder_config* cfg, const nope_tensor_desc* tensors, int n_tensors, nope_stream_t stream,
                        nope_encoder** out) {
    if (!cfg || !tensors || !out || n_tensors <= 0) return NOPE_ERR_ARG;
    if (!dt_is_compute(cfg->compute_dtype)) return NOPE_ERR_UNSUPPORTED;
    const int D = cfg->descriptor_size;
    if (D <= 0 || D > 2048) return NOPE_ERR_UNSUPPORTED;
    nope_encoder* enc = new nope_encoder();
    enc->cfg = *cfg;
    if (enc->cfg.bn_eps <= 0.f) enc->cfg.bn_eps = 1e-5f;      // nn.BatchNorm2d default
    enc->dt = dt_base(cfg->compute_dtype);      // (NOPE_F16X2 = NOPE_BF16X3 here: the encoder has no tap-resident launches worth a second weight pack)
    ELoader L{enc, (hipStream_t)stream};
    for (int i = 0; i < n_tensors; ++i) if (tensors[i].name) L.tab[tensors[i].name] = &tensors[i];

    {   // conv1 + bn1 (resnet.py:98-99)
        float* scale = nullptr;
        const nope_tensor_desc* w = L.get("backbone.conv1.weight", {FEATURES, 3, 7, 7});
        if (L.bn("backbone.bn1.", FEATURES, scale, enc->stem_shift) && w) {
            enc->stem_w = (float*)L.dmalloc((size_t)147 * FEATURES * 4);
            if (enc->stem_w) L.chk(launch_stem_pack(w->data, scale, enc->stem_w, L.s));
        }
    }
    int cin = FEATURES;
    for (int l = 0; l < 4; ++l) {
        const int planes = FEATURES << l;
        for (int b = 0; b < LAYERS[l]; ++b) {
            const std::string p = "backbone.layer" + std::to_string(l + 1) + "." + std::to_string(b) + ".";
            const int st = b == 0 ? STRIDES[l] : 1;
            Bottleneck bk;
            bk.c1 = L.conv_bn(p + "conv1.weight", p + "bn1.", cin, planes, 1, 1);
            bk.c2 = L.conv_bn(p + "conv2.weight", p + "bn2.", planes, planes, 3, st);
            bk.c3 = L.conv_bn(p + "conv3.weight", p + "bn3.", planes, planes * 4, 1, 1);
            bk.has_ds = b == 0;                                  // resnet.py:119-128: stride != 1 or inplanes != planes * 4
            if (bk.has_ds) bk.ds = L.conv_bn(p + "downsample.0.weight", p + "downsample.1.", cin, planes * 4, 1, st);
            enc->blocks.push_back(bk);
            enc->block_stride.push_back(st);
            cin = planes * 4;
        }
    }
    enc->proj0 = L.conv_bn("projector.1.weight", "", cin, 256, 1, 1);
    enc->proj1 = L.conv_bn("projector.3.weight", "", 256, D, 1, 1);
    if (hipStreamSynchronize(L.s) != hipSuccess && L.err == NOPE_OK) L.err = NOPE_ERR_LAUNCH;   // the sources may be freed by the caller
    if (L.err != NOPE_OK) {
        if (L.err == NOPE_ERR_WEIGHT) fprintf(stderr, "nope_encoder_create: missing or mis-shaped tensor '%s'\n", L.missing.c_str());
        nope_encoder_destroy(enc);
        return L.err;
    }
    *out = enc;
    return NOPE_OK;
}

void nope_encoder_destroy(nope_encoder* enc) {
    if (!enc) return;
    for (const EGraph& g : enc->graphs) hipGraphExecDestroy(g.exec);
    for (void* p : enc->allocs) hipFree(p);
    delete enc;
}

// Workspace = [staged image | staged output | activation arena].
static size_t stage_bytes(const nope_encoder* enc, int n_img, int H, int W, size_t& img_b, size_t& out_b) {
    img_b = align_up((size_t)n_img * 3 * H * W * 4, 256);
    out_b = align_up((size_t)n_img * enc->cfg.descriptor_size * (H / 8) * (W / 8) * 4, 256);
    return img_b + out_b;
}

size_t nope_encoder_workspace_bytes(const nope_encoder* enc, int n_img, int H, int W) {
    if (!enc || n_img <= 0 || H <= 0 || W <= 0 || H % 8 || W % 8) return 0;
    EArena ar;
    ar.dry = true;
    if (run_encoder(enc, nullptr, n_img, H, W, nullptr, ar, nullptr) != NOPE_OK) return 0;
    size_t ib, ob;
    return stage_bytes(enc, n_img, H, W, ib, ob) + align_up(ar.off, 256);
}

int nope_encoder_forward(const nope_encoder* enc, const float* image, int n_img, int H, int W, float* out, void* workspace,
                         size_t workspace_bytes, nope_stream_t stream) {
    if (!enc || !image || !out || !workspace || n_img <= 0) return NOPE_ERR_ARG;
    if (H <= 0 || W <= 0 || H % 8 || W % 8) return NOPE_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    size_t ib, ob;
    const size_t sb = stage_bytes(enc, n_img, H, W, ib, ob);
    if (workspace_bytes < sb) return NOPE_ERR_WORKSPACE;
    unsigned char* base = (unsigned char*)workspace;
    float* img_s = (float*)base;
    float* out_s = (float*)(base + ib);
    auto direct = [&](const float* src, float* dst) {
        EArena ar;
        ar.base = base + sb; ar.cap = workspace_bytes - sb;
        return run_encoder(enc, src, n_img, H, W, dst, ar, s);
    };
    const bool graphs_wanted = (NOPE_ENV("NOPE_ENC_GRAPH", -1) != 0);      // (the tests compare both paths)
    if (!graphs_wanted) return direct(image, out);
    std::lock_guard<std::mutex> lock(enc->graph_mu);
    if (!enc->graphs_ok) return direct(image, out);

    const size_t in_bytes = (size_t)n_img * 3 * H * W * 4;
    const size_t out_bytes = (size_t)n_img * enc->cfg.descriptor_size * (H / 8) * (W / 8) * 4;
    const EGraph* hit = nullptr;
    for (const EGraph& g : enc->graphs)
        if (g.ws == workspace && g.ws_bytes == workspace_bytes && g.n_img == n_img && g.H == H && g.W == W) { hit = &g; break; }
    if (!hit) {
        // First call for this (workspace, shape): record the launch sequence once.
        {   // dry pass: fail on a too-small arena BEFORE a capture is open
            EArena chk; chk.dry = true;
            int e = run_encoder(enc, nullptr, n_img, H, W, nullptr, chk, nullptr);
            if (e) return e;
            if (align_up(chk.off, 256) > workspace_bytes - sb) return NOPE_ERR_WORKSPACE;
        }
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            enc->graphs_ok = false;
            return direct(image, out);
        }
        const int e = direct(img_s, out_s);
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(s, &graph);
        hipGraphExec_t exec = nullptr;
        if (e != NOPE_OK || ce != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) hipGraphDestroy(graph);
            (void)hipGetLastError();
            enc->graphs_ok = false;
            return e != NOPE_OK ? e : direct(image, out);
        }
        hipGraphDestroy(graph);
        if (enc->graphs.size() >= 16) { hipGraphExecDestroy(enc->graphs.front().exec); enc->graphs.erase(enc->graphs.begin()); }
        enc->graphs.push_back(EGraph{workspace, workspace_bytes, n_img, H, W, exec});
        hit = &enc->graphs.back();
    }
    if (hipMemcpyAsync(img_s, image, in_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return NOPE_ERR_LAUNCH;
    if (hipGraphLaunch(hit->exec, s) != hipSuccess) return NOPE_ERR_LAUNCH;
    if (hipMemcpyAsync(out, out_s, out_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return NOPE_ERR_LAUNCH;
    return NOPE_OK;
}

}  // extern "C"
