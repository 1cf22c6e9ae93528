// extern "C" entry points of libnope_hip.so that are thin argument checks over the kernel
// launchers (the U-Net entry points live in unet_runtime.hip).  See include/nope_hip.h.
#include "nope_common.h"

#include <atomic>
#include <cstdlib>
#include <mutex>

using namespace nope;

namespace nope {
static std::atomic<unsigned> g_tuning_gen{0};
static std::mutex g_env_mu;
unsigned tuning_generation() { return g_tuning_gen.load(std::memory_order_acquire); }
EnvVal env_lookup(EnvCache& c, const char* name) {
    const unsigned g = tuning_generation();
    if (c.gen.load(std::memory_order_acquire) != g) {  // first use of this site, or a reload since: read the variable (rare path, serialised)
        std::lock_guard<std::mutex> lock(g_env_mu);
        const char* v = getenv(name);
        c.set.store(v != nullptr, std::memory_order_relaxed);
        c.val.store(v ? atoll(v) : 0, std::memory_order_relaxed);
        c.gen.store(g, std::memory_order_release);
    }
    return EnvVal{c.set.load(std::memory_order_relaxed), c.val.load(std::memory_order_relaxed)};
}
}  // namespace nope

extern "C" {

int nope_abi_version(void) { return NOPE_ABI_VERSION; }
void nope_tuning_reload(void) { g_tuning_gen.fetch_add(1, std::memory_order_acq_rel); }

const char* nope_strerror(int code) {
    switch (code) {
        case NOPE_OK: return "ok";
        case NOPE_ERR_RANGE: return "f16x2: activations outside a layer's accurate range (shifts adjusted: run again)";
        case NOPE_ERR_RANGE_F16: return "f16x2: non-finite activations (run as bf16x3)";
        case NOPE_ERR_ARG: return "invalid argument";
        case NOPE_ERR_LAUNCH: return "HIP launch/runtime error";
        case NOPE_ERR_WORKSPACE: return "workspace too small";
        case NOPE_ERR_WEIGHT: return "missing or mis-shaped state-dict tensor";
        case NOPE_ERR_ALLOC: return "device allocation failed";
        case NOPE_ERR_UNSUPPORTED: return "unsupported size or dtype";
        default: return "unknown error";
    }
}

int nope_similarity(const float* q, const void* bank, int bank_dtype, float* scores, int B, int N, int C, int H, int W,
                    int64_t bank_stride_b, int score_ld, nope_stream_t stream) {
    if (H <= 0 || W <= 0) return NOPE_ERR_ARG;
    return launch_similarity(q, bank, bank_dtype, scores, B, N, C, H * W, (long long)bank_stride_b, score_ld, (hipStream_t)stream);
}

int nope_topk(const float* scores, int64_t* idx, float* vals, int B, int N, int k, int score_ld, nope_stream_t stream) {
    return launch_topk(scores, (long long*)idx, vals, B, N, k, score_ld, (hipStream_t)stream);
}

int nope_gather_topk(const float* gathered, int n_ranks, int B, int n_total, float* scores, int64_t* idx, float* vals, int k, nope_stream_t stream) {
    return launch_gather_topk(gathered, n_ranks, B, n_total, scores, (long long*)idx, vals, k, (hipStream_t)stream);
}

int nope_topk_merge(const float* cand_vals, const int64_t* cand_idx, int64_t* idx, float* vals, int B, int M, int k, nope_stream_t stream) {
    if (!cand_idx) return NOPE_ERR_ARG;
    return launch_topk(cand_vals, (long long*)idx, vals, B, M, k, M, (hipStream_t)stream, (const long long*)cand_idx);
}

int nope_op_geodesic(const double* poses, int64_t pose_stride_b, int N, const int64_t* idx, const double* gt, const int* symmetry,
                     double* err_rad, int* status, int B, int k, nope_stream_t stream) {
    return launch_geodesic(poses, (long long)pose_stride_b, N, (const long long*)idx, gt, symmetry, err_rad, status, B, k, (hipStream_t)stream);
}

int nope_op_nchw_to_nhwc(int dtype, const float* x, void* y, int n, int C, int HW, nope_stream_t s) {
    return launch_nchw_to_nhwc(dtype, x, y, n, C, HW, (hipStream_t)s);
}
int nope_op_nhwc_to_nchw(int dtype, const void* x, float* y, int n, int C, int HW, nope_stream_t s) {
    return launch_nhwc_to_nchw_f32(dtype, x, y, n, C, HW, (hipStream_t)s);
}
int nope_op_pack_conv_weight(int dtype, const float* w, void* packed, int Cout, int Cin, int ntaps, int mode, nope_stream_t s) {
    return launch_pack_conv_w(dtype, w, packed, Cout, Cin, ntaps, mode, (hipStream_t)s);
}

static void fill_conv_args(ConvArgs& a, const void* src1, int C1, int rep1, const void* src2, int C2, int rep2, int Hs, int Ws, int mode, int ntaps,
                           const void* w_packed, const float* bias, const void* resid, void* out, int Cout, int n_hyp, int out_nchw,
                           int out_dtype, int act_relu) {
    a.src1 = src1; a.C1 = C1; a.rep1 = rep1; a.src2 = src2; a.C2 = C2; a.rep2 = rep2 > 0 ? rep2 : 1;
    a.Hs = Hs; a.Ws = Ws; a.mode = mode; a.ntaps = ntaps;
    const bool up = mode == NOPE_CONV_UP2 || mode == NOPE_CONV_UP2P;
    const bool half = mode == NOPE_CONV_DOWN2 || mode == NOPE_CONV_STRIDE2;
    a.Ho = up ? 2 * Hs : (half ? Hs / 2 : Hs);
    a.Wo = up ? 2 * Ws : (half ? Ws / 2 : Ws);
    a.act = act_relu ? 1 : 0;
    a.w = w_packed; a.bias = bias; a.resid = resid; a.out = out; a.Cout = Cout; a.nhyp = n_hyp;
    a.out_nchw = out_nchw; a.out_dt = out_dtype;
}

int nope_op_conv_ws(int dtype, const void* src1, int C1, int rep1, const void* src2, int C2, int rep2, int Hs, int Ws, int mode,
                    int ntaps, const void* w_packed, const float* bias, const void* resid, void* out, int Cout, int n_hyp,
                    int out_nchw, int out_dtype, int act_relu, void* splitk_ws, size_t splitk_bytes, nope_stream_t s) {
    ConvArgs a;
    fill_conv_args(a, src1, C1, rep1, src2, C2, rep2, Hs, Ws, mode, ntaps, w_packed, bias, resid, out, Cout, n_hyp, out_nchw, out_dtype, act_relu);
    if ((mode == NOPE_CONV_DOWN2 || mode == NOPE_CONV_STRIDE2) && ((Hs | Ws) & 1)) return NOPE_ERR_ARG;
    a.splitk_ws = splitk_ws; a.splitk_bytes = splitk_ws ? splitk_bytes : 0;
    return launch_conv(dtype, a, (hipStream_t)s);
}

int nope_op_conv(int dtype, const void* src1, int C1, int rep1, const void* src2, int C2, int rep2, int Hs, int Ws, int mode,
                 int ntaps, const void* w_packed, const float* bias, const void* resid, void* out, int Cout, int n_hyp,
                 int out_nchw, int out_dtype, int act_relu, nope_stream_t s) {
    return nope_op_conv_ws(dtype, src1, C1, rep1, src2, C2, rep2, Hs, Ws, mode, ntaps, w_packed, bias, resid, out, Cout, n_hyp, out_nchw,
                           out_dtype, act_relu, nullptr, 0, s);
}

size_t nope_op_conv_splitk_bytes(int dtype, int C1, int C2, int rep1, int Hs, int Ws, int mode, int ntaps, int Cout, int n_hyp) {
    ConvArgs a;
    static const int dummy = 0;
    fill_conv_args(a, &dummy, C1, rep1, C2 ? &dummy : nullptr, C2, 1, Hs, Ws, mode, ntaps, &dummy, nullptr, nullptr, (void*)&dummy, Cout, n_hyp, 0, NOPE_F32, 0);
    if (!dt_is_compute(dtype)) return 0;
    const int S = conv_splitk_factor(dtype, a);
    return S > 1 ? (size_t)S * (size_t)n_hyp * a.Ho * a.Wo * Cout * 4 : 0;
}

int nope_op_stem_conv(int dtype, const float* image, const float* w, const float* scale, const float* shift, float* w_scratch,
                      void* out, int n_img, int H, int W, nope_stream_t s) {
    int e = launch_stem_pack(w, scale, w_scratch, (hipStream_t)s);      // w_scratch: 147 * 64 floats
    if (e) return e;
    return launch_stem_conv(dtype, image, w_scratch, shift, out, n_img, H, W, (hipStream_t)s);
}

int nope_op_gn_chunks(int dtype, int HW, int C) { return gn_stats_chunks(HW, C, dtype); }

int nope_op_group_norm(int dtype, const void* x, void* y, float* partial, const float* gamma, const float* beta, int n_hyp, int HW,
                       int C, int G, int act_silu, const float* emb, int emb_stride, const void* resid, nope_stream_t s) {
    const int nch = gn_stats_chunks(HW, C, dtype);
    int e = launch_gn_stats(dtype, x, partial, n_hyp, HW, C, G, nch, (hipStream_t)s);
    if (e) return e;
    GnApplyArgs a;
    a.x = x; a.y = y; a.partial = partial; a.nchunk = nch; a.gamma = gamma; a.beta = beta;
    a.nhyp = n_hyp; a.HW = HW; a.C = C; a.G = G; a.act = act_silu; a.emb = emb; a.emb_stride = emb_stride; a.resid = resid;
    return launch_gn_apply(dtype, a, (hipStream_t)s);
}

int nope_op_linear_attention(int dtype, const void* qkv, void* out, int n_hyp, int HW, int heads, int dim_head, nope_stream_t s) {
    return launch_linattn(dtype, qkv, out, n_hyp, HW, heads, dim_head, (hipStream_t)s);
}
int nope_op_attention(int dtype, const void* qkv, void* out, int n_hyp, int HW, int heads, int dim_head, nope_stream_t s) {
    return launch_attn(dtype, qkv, out, n_hyp, HW, heads, dim_head, (hipStream_t)s);
}
int nope_op_layer_norm(int dtype, const void* x, void* y, const float* gamma, const float* beta, int64_t M, int C, float eps, nope_stream_t s) {
    return launch_layernorm(dtype, x, y, gamma, beta, (long long)M, C, eps, (hipStream_t)s);
}
int nope_op_geglu(int dtype, const void* in, void* out, int64_t M, int D, nope_stream_t s) {
    return launch_geglu(dtype, in, out, (long long)M, D, (hipStream_t)s);
}
int nope_op_token_attention(int dtype, const void* qkv, void* out, int n, int N, int C, int dim_head, nope_stream_t s) {
    return launch_token_attention(dtype, qkv, out, n, N, C, dim_head, (hipStream_t)s);
}
int nope_op_warp_perspective(const void* src, int src_is_u8, int Hs, int Ws, int C, const float* minv9_host, float* dst_chw, int Hd, int Wd,
                             float scale, float shift, nope_stream_t s) {
    return launch_warp_perspective(src, src_is_u8, Hs, Ws, C, minv9_host, dst_chw, Hd, Wd, scale, shift, (hipStream_t)s);
}
int nope_op_linear(const float* in, const float* w, const float* bias, float* out, int M, int N, int K, int act_in, nope_stream_t s) {
    return launch_linear_naive(in, w, bias, out, M, N, K, act_in, N, (hipStream_t)s);
}

}  // extern "C"
