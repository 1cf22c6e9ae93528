/* nope_hip.h -- C ABI of libnope_hip.so: the MI355X (gfx950) implementation of the NOPE
 * inference hot path (template encoder + pose-conditioned U-Net template generation +
 * template-bank scoring).
 *
 * The reference (nv-nguyen/nope) is pure Python on torch ops and has no FFI of its own; the
 * drop-in boundary is its Python operator interface (src/model/model.py, u_net.py), mirrored
 * by `nope_amd/` on top of THIS library.  Each entry point names the reference code whose
 * arithmetic it replaces (paths relative to the reference tree).  See INTEGRATION.md for the
 * ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer unless it says "host";
 *   - the caller owns all buffers; only nope_unet_create / nope_encoder_create allocate device memory
 *     (packed weights), and nope_encoder_forward keeps host-side hipGraph objects in its handle;
 *   - `stream` is a hipStream_t passed as void*; all work is asynchronous on it;
 *   - no global state; calls with different handles, or with one U-Net handle and different workspaces, may
 *     run on different streams; returns 0 or a negative NOPE_ERR_* code;
 *   - nothing here ever falls back to the host: without a GPU the calls fail.
 */
#ifndef NOPE_HIP_H
#define NOPE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { NOPE_F32 = 0, NOPE_BF16 = 1,
       NOPE_F16 = 2,   /* IEEE half: storage type of the template bank (nope_similarity's bank_dtype, nope_unet_forward's out_dtype,
                          BASELINE configs[4]) and a compute mode of the networks: f16 storage + f16 MFMA, f32 accumulate / statistics
                          -- the MFMA rate of NOPE_BF16 with 3 more mantissa bits; stores saturate at +-65504 (no inf; a NaN is stored as -65504) */
       NOPE_BF16X3 = 3,/* compute mode only: f32 storage, every conv / linear as three bf16 MFMA passes over (hi, lo) bf16 splits of
                          both operands (hi*hi + hi*lo + lo*hi, f32 accumulate): ~2^-17 relative per product instead of bf16's 2^-9
                          at 3/16 of the exact-f32 MFMA cost -- meets the 1e-4 score tolerance */
       NOPE_F16X2 = 4  /* compute mode only: NOPE_BF16X3 (f32 storage, same kernels) except that the convolutions the ping-pong kernels run --
                          the tap-resident 3x3 kernel, 9/10 of the U-Net's work, and the per-tap kernel's long 1x1 / space-to-depth / phase
                          convs -- cost TWO pass equivalents instead of three: a_hi w_hi on the f16 MFMA
                          (hi = f16 part) plus ONE MX-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, twice the f16 rate) for both
                          cross terms, K-concatenated: [e4m3(a_lo) | e4m3(a)] x [e4m3(w) ; e4m3(w_lo)], power-of-two pre-scales undone by
                          the instruction's block scale.  The cross terms carry <= 2^-11 of the result, so 4-bit operands leave ~2^-15
                          per product -- the fast mode that meets the 1e-4 score tolerance.  As an element type of nope_op_conv /
                          nope_op_pack_conv_weight it names those kernels and their weight layout (modes PLAIN 1x1 / 3x3, DOWN2, UP2P;
                          Cin % 32 == 0; nope_op_conv refuses a launch whose shape no ping-pong kernel takes).
                          RANGE.  The tile forms its activation operands from a' = a * 2^-t, t a per-layer shift (0 at create time and for
                          nope_op_conv): f16(a') (saturates at 65504), e4m3(a'_lo * 2^9), e4m3(a' * 2^-2) (saturates at |a'| = 1792; four significant
                          bits down to |a'| = 2^-4); the accumulators hold 2^-t x the convolution and the epilogue multiplies by 2^t -- all
                          exact.  Full accuracy while the LARGEST |a| of a launch lies in [2^(t - 4), 1792 * 2^t] (t = 0: 0.06 .. 1792); outside,
                          the cross terms of the saturated / flushed elements are lost: plain-f16 accuracy (~2^-11 per product) there.  The
                          producers of a layer's inputs record max |a|; every forward is judged on the device and an out-of-range forward's
                          output is NaN (nope_unet_x2_poll / _range_check re-centre t and say so; nope_amd's U-Net repeats the call on request),
                          so the mode is never silently outside its accuracy at any magnitude an f32 tensor can hold */ };
/* Tap geometries of nope_op_conv.  UP2P is UP2 (nearest-x2 upsample + 3x3, pad 1) rewritten as four
 * 2x2 convolutions over the un-upsampled input, one per output-pixel parity, with the 3x3 weights that
 * fall on the same source pixel pre-summed at pack time: same function, 4/9 of the multiply-adds. */
enum { NOPE_CONV_PLAIN = 0, NOPE_CONV_UP2 = 1, NOPE_CONV_DOWN2 = 2, NOPE_CONV_UP2P = 3,
       NOPE_CONV_STRIDE2 = 4 /* stride 2: 3x3 pad 1 or 1x1 pad 0 (ResNet Bottleneck, encoder/resnet.py:64-65,122-123), 4x4 pad 1 (Downsample, model_utils.py:129-136) */ };
enum {
    NOPE_OK = 0,
    NOPE_ERR_ARG = -1,        /* bad argument (null pointer, unsupported size/dtype) */
    NOPE_ERR_LAUNCH = -2,     /* HIP reported a launch error */
    NOPE_ERR_WORKSPACE = -3,  /* workspace too small */
    NOPE_ERR_WEIGHT = -4,     /* missing / mis-shaped state-dict entry */
    NOPE_ERR_ALLOC = -5,
    NOPE_ERR_UNSUPPORTED = -6,
    NOPE_ERR_RANGE = -7,      /* nope_unet_x2_range_check: a NOPE_F16X2 launch saw activations outside its layer's window; the shifts were moved -- run the forward again */
    NOPE_ERR_RANGE_F16 = -8   /* ... non-finite activations (inf): no shift repairs that; nope_unet_x2_enable(net, 0) runs the NOPE_BF16X3 kernels, which propagate them as f32 does */
};

typedef void* nope_stream_t;

/* Bumped whenever a struct of this header changes layout or an enum gains a meaning (2: nope_unet_config.soft_up_down;
 * 3: NOPE_F16 / NOPE_BF16X3 compute modes, 4x4 STRIDE2, nope_ldm_config.transformer_depth;
 * 4: nope_op_geodesic, nope_unet_graph_limit -- hipGraph replay became opt-in;
 * 5: NOPE_F16X2, nope_tuning_reload, nope_gather_topk, nope_topk_merge;
 * 6: nope_unet_x2_poll / _x2_range_check / _x2_enable / _x2_shifts, NOPE_ERR_RANGE*).  Callers compare nope_abi_version() against the header they were
 * built with before passing any struct (nope_amd/hip.py does at load time). */
#define NOPE_ABI_VERSION 6
const char* nope_strerror(int code);
int nope_abi_version(void);
/* The library reads its tuning / test switches (NOPE_* environment variables: launch policies, A/B switches, traces) once per call site and
 * caches them.  A caller that changes one of them after its first call into the library calls this to have them read again (the Python
 * binding does so by itself).  No reference counterpart: the reference has no tuning switches. */
void nope_tuning_reload(void);

/* ------------------------------------------------------------------------------------------
 * Template-bank scoring.  Replaces PoseConditional.retrieval's "l2" metric,
 * src/model/model.py:257-262:
 *     score[b,n] = - sum_{h,w} sqrt( sum_c (q[b,c,h,w] - t[b,n,c,h,w])^4 )
 * without materialising the repeated query (model.py:258).
 *   q       (B,C,H,W) f32, contiguous
 *   bank    (B,N,C,H,W) of `bank_dtype` (NOPE_F32 | NOPE_BF16 | NOPE_F16); sample stride `bank_stride_b`
 *           ELEMENTS (0 = one bank shared by every query)
 *   scores  f32, row b at scores + b*score_ld  (score_ld >= N; lets a rank write its
 *           N/G slice of a gathered (B,N) matrix in place)
 * HW*sizeof(elt) must be a multiple of 16 bytes. */
int nope_similarity(const float* q, const void* bank, int bank_dtype, float* scores, int B, int N, int C, int H,
                    int W, int64_t bank_stride_b, int score_ld, nope_stream_t stream);

/* Top-k over each row.  Replaces `similarity.topk(k=5, dim=1)`, model.py:265.
 * Order: descending score; ties -> lowest index (argmax semantics); NaN ranks highest.
 *   idx   (B,k) int64;  vals (B,k) f32 or NULL.   1 <= k <= 16, k <= N. */
int nope_topk(const float* scores, int64_t* idx, float* vals, int B, int N, int k, int score_ld,
              nope_stream_t stream);

/* Template-sharded banks (north_star: "partition [the bank] across the 8 GPUs of one node with an RCCL all-gather of per-shard top-k
 * scores"; the reference has no collective on this path).  Rank r of G holds the contiguous slice [lo_r, hi_r) of the N
 * templates (balanced: the first N % G ranks one more) and scores it into a padded (B, nmax) buffer, nmax = ceil(N / G); the collective
 * itself stays with the caller (torch.distributed / RCCL).  Two ways to finish the step, each ONE launch behind the collective:
 *   nope_gather_topk  gathered (G,B,nmax) f32 = the all-gathered slices -> scores (B,N) f32, the full similarity `retrieval` returns and the
 *                     harness saves (model.py:323,369-375), AND its top-k (model.py:265; k = 0: scores only, idx / vals may be NULL);
 *   nope_topk_merge   cand_vals / cand_idx (B,M) = the all-gathered per-shard top-k lists (M = G k values with their GLOBAL template
 *                     indices, shards in rank order) -> the global top-k.  Same order as nope_topk on the full row: descending score, ties
 *                     -> lowest global index (a shard's list is already in that order and shards are contiguous, so position order =
 *                     index order among equal scores); pad short lists with (-inf, any index).  For callers that do not need the
 *                     full similarity: 12 k bytes per query and rank cross the fabric instead of 4 N / G. */
int nope_gather_topk(const float* gathered, int n_ranks, int B, int n_total, float* scores, int64_t* idx, float* vals, int k,
                     nope_stream_t stream);
int nope_topk_merge(const float* cand_vals, const int64_t* cand_idx, int64_t* idx, float* vals, int B, int M, int k, nope_stream_t stream);

/* Geodesic error of the retrieved poses against the ground truth (row f2).  Replaces `pred_R = template_poses[nearest_idx]`,
 * src/model/model.py:352-354, and GeodesicError's per-element arithmetic, src/model/loss.py:14-75 (so3_relative_angle_with_symmetry:
 * symmetry 0 = none, 1 = 180 degrees about Y, 2 = circular) with pytorch3d's so3_relative_angle(eps = 1e-2) restated from its
 * published formula (acos_linear_extrapolation, bound 1 - 1e-4).  float64 throughout, as loss.py:87,103 casts.
 *   poses     (B, N, 3, 3) f64, sample stride `pose_stride_b` elements (0 = one grid shared by every query)
 *   idx       (B, k) int64 rows of `poses` to score (nope_topk's output), or NULL: score poses[b, 0..k)
 *   gt        (B, 3, 3) f64;  symmetry (B) int32 in {0, 1, 2} or NULL (all 0)
 *   err_rad   (B, k) f64 radians
 *   status    one device int: bit 0 = a trace left [-1 - eps, 3 + eps] (pytorch3d raises ValueError there: the caller must),
 *             bit 1 = an index outside [0, N). */
int nope_op_geodesic(const double* poses, int64_t pose_stride_b, int N, const int64_t* idx, const double* gt, const int* symmetry,
                     double* err_rad, int* status, int B, int k, nope_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Pose-conditioned U-Net.  Replaces UNet.__init__/forward,
 * src/model/u_net/denoising_diffusion_pytorch/u_net.py:27-198 and the blocks of
 * model_utils.py:161-172,198-279,367-418.
 */
typedef struct nope_unet nope_unet;

typedef struct {
    const char* name;      /* host: reference state-dict key, e.g. "downs.0.0.block1.proj.weight" */
    const float* data;     /* device: f32, contiguous, torch layout (Conv2d: [Cout,Cin,kh,kw]) */
    int ndim;
    int64_t shape[4];
} nope_tensor_desc;

typedef struct {
    int u_net_dim;         /* 192  (configs/model/template_base.yaml:4) */
    int channels;          /* encoder.latent_dim, 8 (u_net.py:45) */
    int out_dim;           /* = channels (u_net.py:50) */
    int pose_dim;          /* rot_representation_dim, 6 */
    int n_levels;          /* len(dim_mults) = 4 */
    int dim_mults[8];      /* (1,2,4,8) */
    int groups;            /* resnet_block_groups = 8 */
    int heads, dim_head;   /* 4, 32 (model_utils.py:368,394) */
    int pose_mlp_layers;   /* 1 = "single_layer", 2 = "two_layers" (u_net.py:63-72) */
    int compute_dtype;     /* NOPE_F32: f32 storage + f32-input MFMA (bit-faithful fp32 sums);
                              NOPE_BF16 / NOPE_F16: 16-bit storage + 16-bit MFMA, f32 accumulate / statistics;
                              NOPE_BF16X3 / NOPE_F16X2: f32 storage, split-precision MFMA (see the enum) */
    int soft_up_down;      /* 0: use_hard_up_down = True, the shipped configuration (HardDownsample / HardUpsample, u_net.py:54-56);
                              1: use_hard_up_down = False -- Downsample = Conv2d(4, stride 2, pad 1) at "downs.l.3.weight",
                              Upsample = ConvTranspose2d(4, stride 2, pad 1) at "ups.l.3.weight" (model_utils.py:119-136) */
} nope_unet_config;

/* Validates and repacks the reference state dict for the device (the only allocating call). */
int nope_unet_create(const nope_unet_config* cfg, const nope_tensor_desc* tensors, int n_tensors,
                     nope_stream_t stream, nope_unet** out);
void nope_unet_destroy(nope_unet* net);

size_t nope_unet_workspace_bytes(const nope_unet* net, int n_hyp, int n_src, int H, int W);

/* out[j] = UNet(x[j / x_rep], pose[j])  for j in [0, n_hyp).
 *   x     (n_src, channels, H, W) f32 NCHW, n_src * x_rep == n_hyp.  x_rep = 1 is
 *         UNet.forward(x, pose) (u_net.py:160); x_rep = N evaluates N pose hypotheses per
 *         reference embedding, the batched form of the template loop (model.py:212-222).
 *   pose  (n_hyp, pose_dim) f32
 *   out   (n_hyp, out_dim, H, W) NCHW of `out_dtype` -- i.e. directly a slice of the
 *         (B,N,C,h,w) template bank.
 * H and W must be divisible by 2^(n_levels-1). */
int nope_unet_forward(const nope_unet* net, const float* x, int n_src, int x_rep, const float* pose, int n_hyp,
                      int H, int W, void* out, int out_dtype, void* workspace, size_t workspace_bytes,
                      nope_stream_t stream);

/* hipGraph replay of SMALL forwards, opt-in (no reference counterpart): with max_hyp_pixels > 0, forwards of at most that many
 * n_hyp * H * W replay a captured launch sequence (x, pose and the output are staged through the head of the workspace so the
 * caller's pointers stay out of the graph; one graph per (workspace, shape, out_dtype), cache guarded by a mutex).  0 (the default;
 * NOPE_UNET_GRAPH in the environment at create time overrides) = always launch directly.  Results are bit-identical either way
 * (tests/test_gpu_configs.py::test_unet_graph_replay_matches_direct); a 64-hypothesis pass measured +-0 on MI355X, which is why it is
 * off.  nope_unet_graph_replays: forwards served by a replay since create. */
int nope_unet_graph_limit(nope_unet* net, long long max_hyp_pixels);

/* NOPE_F16X2 activation ranges (no reference counterpart: the reference computes in fp32, model_utils.py:240-252,271-279 see whatever
 * magnitude the residual stream has).  Every nope_unet_forward of a NOPE_F16X2 net ends with a verdict formed ON THE DEVICE: the largest
 * |activation| each two-pass layer read (recorded by the producers of its inputs) against the layer's window.  A forward with a layer outside
 * its window leaves NaNs in `out` -- inaccurate values never pass for accurate ones -- and no host synchronisation is involved.
 *   nope_unet_x2_poll         reads, WITHOUT synchronising, the verdicts that have reached the host since the previous poll, re-centres the
 *                             shifts of the layers that were out of (or within two binades of an end of) their windows -- the new shifts are
 *                             enqueued on `stream` -- and returns NOPE_OK, NOPE_ERR_RANGE (a judged forward was out of range: its output is
 *                             NaN, issue it again) or NOPE_ERR_RANGE_F16 (an activation was infinite: no shift helps; nope_unet_x2_enable(net, 0)
 *                             makes every launch NOPE_BF16X3 -- same weights, the three-pass kernels).  nope_unet_forward polls at entry.
 *   nope_unet_x2_range_check  synchronises `stream` first: the verdict of every forward issued on it so far.
 * n_out_of_range / n_adjusted / max_abs (each may be null): layers out of range in the judged forwards, layers whose shift moved, largest
 * |a| seen.  A net created in another mode: NOPE_OK, nothing to check.  nope_unet_x2_shifts: the current per-layer shifts (creation order). */
int nope_unet_x2_poll(nope_unet* net, nope_stream_t stream, int* n_out_of_range, int* n_adjusted, float* max_abs);
int nope_unet_x2_range_check(nope_unet* net, nope_stream_t stream, int* n_out_of_range, int* n_adjusted, float* max_abs);
int nope_unet_x2_enable(nope_unet* net, int on);
int nope_unet_x2_shifts(const nope_unet* net, int* shifts, int max, int* n);
int nope_unet_graph_replays(const nope_unet* net);

/* Measurement aid (bench.py roofline leg, no reference counterpart): while enabled, every
 * launch of the implicit-GEMM conv kernel made by nope_unet_forward is bracketed by HIP events on
 * the caller's stream; _read synchronises them and returns the launch count, the summed kernel
 * time, the summed executed flops (2*M*Cout*taps*Cin) and the summed algorithmic HBM bytes (each input,
 * weight and output element once).  Adds two event records per launch:
 * keep it off in timed regions. */
int nope_unet_profile(nope_unet* net, int enable);
int nope_unet_profile_read(nope_unet* net, int* n_launches, double* total_ms, double* total_flops, double* total_bytes);
/* ... and launch by launch, in issue order: which kernel took the launch, its shape, its HIP-event time.  `flops` counts the
 * convolution's multiply-adds x 2 as executed (NOPE_BF16X3 issues three MFMA passes per product: mfma_passes = 3).  Writes at
 * most `max` records and the total number of recorded launches to *n. */
enum { NOPE_CONV_KERNEL_GENERIC = 0, NOPE_CONV_KERNEL_DMA128 = 1, NOPE_CONV_KERNEL_PP256 = 2, NOPE_CONV_KERNEL_HALO256 = 3, NOPE_CONV_KERNEL_SMALL = 4, NOPE_CONV_KERNEL_STREAM = 5 };
typedef struct {
    double ms, flops, bytes;
    int kernel;            /* NOPE_CONV_KERNEL_* */
    int mode, ntaps, Cin, Cout, Hs, Ws, n_hyp, mfma_passes, posmajor;
} nope_conv_launch_info;
int nope_unet_profile_launches(nope_unet* net, nope_conv_launch_info* out, int max, int* n);

/* ------------------------------------------------------------------------------------------
 * LDM cross-attention U-Net variant.  Replaces UNetModelPose.__init__/forward,
 * src/model/u_net/ldm/adapt_openaimodel.py:14-158 (over UNetModel, ldm/openaimodel.py:428-760; ResBlock :177-288, Downsample /
 * Upsample :93-174; SpatialTransformer / BasicTransformerBlock / CrossAttention / GEGLU, ldm/attention.py:37-277): the variant
 * whose pose conditioning is cross-attention against context = pose_mlp(pose).  Tensor names are UNetModelPose's own
 * state-dict keys ("input_blocks.1.1.transformer_blocks.0.attn2.to_v.weight", "middle_block.0.in_layers.2.weight", ...).
 * Supported: use_spatial_transformer = true with transformer_depth >= 1 and num_head_channels = 32 (the shipped
 * configs/model/vae_cin_ldm.yaml), conv_resample, ResBlocks with or without use_scale_shift_norm (FiLM, openaimodel.py:277-281),
 * no resblock_updown; pose_mlp "single_layer" / "two_layers"; injecting_condition_twice on or off. */
typedef struct nope_ldm nope_ldm;
typedef struct {
    int in_channels;        /* 4 in vae_cin_ldm.yaml (any count: the input conv's K axis is zero-padded to a multiple of 8 at pack time) */
    int model_channels;     /* 256 */
    int out_channels;       /* 4 */
    int num_res_blocks;     /* 2 */
    int n_levels;           /* len(channel_mult) = 3 */
    int channel_mult[8];    /* (1,2,4) */
    int attn_levels[8];     /* 1 where the level's downsampling factor is in attention_resolutions: (1,1,1) */
    int num_head_channels;  /* 32 */
    int context_dim;        /* 512 */
    int pose_dim;           /* rot_representation_dim, 6 */
    int pose_mlp_layers;    /* 1 = "single_layer", 2 = "two_layers" */
    int injecting_condition_twice;   /* 0: timestep embedding is zeros; 1: emb = pose_mlp_timesteps(pose) */
    int compute_dtype;      /* NOPE_F32 | NOPE_BF16 | NOPE_F16 | NOPE_BF16X3 | NOPE_F16X2 (= NOPE_BF16X3 here: its layers carry no second weight pack), as nope_unet_config */
    int use_scale_shift_norm;        /* 1: ResBlocks apply out_norm(h) * (1 + scale) + shift with (scale, shift) = emb_layers(emb) */
    int transformer_depth;           /* BasicTransformerBlocks per SpatialTransformer (attention.py:232-262); 1 in vae_cin_ldm.yaml; 0 reads as 1 */
} nope_ldm_config;

int nope_ldm_create(const nope_ldm_config* cfg, const nope_tensor_desc* tensors, int n_tensors, nope_stream_t stream, nope_ldm** out);
void nope_ldm_destroy(nope_ldm* net);
size_t nope_ldm_workspace_bytes(const nope_ldm* net, int n_hyp, int n_src, int H, int W);
/* out[j] = UNetModelPose(x[j / x_rep], pose[j]); arguments as nope_unet_forward. */
int nope_ldm_forward(const nope_ldm* net, const float* x, int n_src, int x_rep, const float* pose, int n_hyp, int H, int W,
                     void* out, int out_dtype, void* workspace, size_t workspace_bytes, nope_stream_t stream);
/* NOPE_F16X2 in the LDM variant: the 3x3 convolutions (ResBlock in_layers.2 / out_layers.3, openaimodel.py:205-243, and the nearest-x2
 * up-sampling convs :95-118) on the two-pass tile, everything else as NOPE_BF16X3; activation ranges exactly as nope_unet_x2_poll /
 * nope_unet_x2_range_check / nope_unet_x2_enable above (device-side verdict, NaN output for an out-of-range forward, poll at every forward). */
int nope_ldm_x2_poll(nope_ldm* net, nope_stream_t stream, int* n_out_of_range, int* n_adjusted, float* max_abs);
int nope_ldm_x2_range_check(nope_ldm* net, nope_stream_t stream, int* n_out_of_range, int* n_adjusted, float* max_abs);
int nope_ldm_x2_enable(nope_ldm* net, int on);

/* ------------------------------------------------------------------------------------------
 * Template encoder.  Replaces FeatureExtractor.encode_image, src/model/encoder/template.py:47-53
 * (ResNet-50 trunk src/model/encoder/resnet.py:92-152 with eval-mode BatchNorm, then the
 * ReLU/1x1/ReLU/1x1 projector template.py:33-38).  Tensor names are the FeatureExtractor's own
 * state-dict keys ("backbone.conv1.weight", "backbone.layer1.0.bn1.running_mean", "projector.1.weight", ...).
 */
typedef struct nope_encoder nope_encoder;
typedef struct {
    int descriptor_size;   /* 8 (configs/model/template_base.yaml:10) */
    int compute_dtype;     /* NOPE_F32 | NOPE_BF16 | NOPE_F16 | NOPE_BF16X3 | NOPE_F16X2 (its 3x3 convs on the two-pass tile, the rest as NOPE_BF16X3), as nope_unet_config */
    float bn_eps;          /* BatchNorm2d eps; <= 0 selects the torch default 1e-5 */
} nope_encoder_config;

int nope_encoder_create(const nope_encoder_config* cfg, const nope_tensor_desc* tensors, int n_tensors,
                        nope_stream_t stream, nope_encoder** out);
void nope_encoder_destroy(nope_encoder* enc);
size_t nope_encoder_workspace_bytes(const nope_encoder* enc, int n_img, int H, int W);
/* image (n_img, 3, H, W) f32 NCHW in [-1, 1], H and W multiples of 8 -> out (n_img, descriptor_size, H/8, W/8) f32 NCHW.
 * The launch sequence of a pass (~85 small kernels) is captured into a hipGraph the first time a
 * (workspace, n_img, H, W) combination is seen and replayed afterwards (image / out are staged through the
 * workspace, so they may change from call to call); if stream capture is unavailable the kernels are launched
 * directly.  The graph cache lives in the handle: calls on ONE handle must come from one host thread at a time,
 * and concurrent passes on different streams need different workspaces. */
int nope_encoder_forward(const nope_encoder* enc, const float* image, int n_img, int H, int W, float* out,
                         void* workspace, size_t workspace_bytes, nope_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Operator-level entry points (NHWC activations of `dtype`), exported so that each block of
 * model_utils.py can be parity-tested in isolation.  Weights for nope_op_conv are packed
 * [Cout][ntaps][Cin] by nope_op_pack_conv_weight.
 */
int nope_op_nchw_to_nhwc(int dtype, const float* x_nchw, void* y_nhwc, int n, int C, int HW, nope_stream_t s);
int nope_op_nhwc_to_nchw(int dtype, const void* x_nhwc, float* y_nchw, int n, int C, int HW, nope_stream_t s);
int nope_op_pack_conv_weight(int dtype, const float* w, void* packed, int Cout, int Cin, int ntaps, int mode,
                             nope_stream_t s);
/* conv3x3(pad1) / conv1x1 / nearest-x2+conv3x3 (HardUpsample, model_utils.py:161-165) /
 * space-to-depth+conv1x1 (HardDownsample, :168-172) / stride-2 conv (encoder) over a virtual channel concat
 * (torch.cat((x, skip), dim=1), u_net.py:186,189,194) as one implicit GEMM. */
int nope_op_conv(int dtype, const void* src1, int C1, int rep1, const void* src2, int C2, int rep2, int Hs, int Ws,
                 int mode, int ntaps, const void* w_packed, const float* bias, const void* resid, void* out,
                 int Cout, int n_hyp, int out_nchw, int out_dtype, int act_relu, nope_stream_t s);
/* ... with a scratch for a deterministic split-K launch (f32 partials + fixed-order reduce) when the launcher wants one for the shape:
 * nope_op_conv_splitk_bytes returns the bytes it would use (0: this shape is never split); a smaller or NULL scratch = no split. */
size_t nope_op_conv_splitk_bytes(int dtype, int C1, int C2, int rep1, int Hs, int Ws, int mode, int ntaps, int Cout, int n_hyp);
int nope_op_conv_ws(int dtype, const void* src1, int C1, int rep1, const void* src2, int C2, int rep2, int Hs, int Ws,
                    int mode, int ntaps, const void* w_packed, const float* bias, const void* resid, void* out,
                    int Cout, int n_hyp, int out_nchw, int out_dtype, int act_relu, void* splitk_ws, size_t splitk_bytes, nope_stream_t s);
/* conv1 of the encoder trunk: 7x7 / stride 2 / pad 3 on an NCHW f32 image, + per-channel scale (folded into the
 * weights) and shift + ReLU -> NHWC (n_img, H/2, W/2, 64).  w (64,3,7,7), scale/shift (64). */
int nope_op_stem_conv(int dtype, const float* image, const float* w, const float* scale, const float* shift, float* w_scratch,
                      void* out, int n_img, int H, int W, nope_stream_t s);
/* GroupNorm(G, C) [+ SiLU] [+ emb[hyp, c]] [+ resid]: Block.norm/act (model_utils.py:241-252),
 * the conditioning add (:274-276), PreNorm (:226-234) and Residual (:198-204).
 * `partial` scratch: n_hyp * nope_op_gn_chunks() * G * 2 floats. */
int nope_op_gn_chunks(int dtype, int HW, int C);
int nope_op_group_norm(int dtype, const void* x, void* y, float* partial, const float* gamma, const float* beta,
                       int n_hyp, int HW, int C, int G, int act_silu, const float* emb, int emb_stride,
                       const void* resid, nope_stream_t s);
/* LinearAttention core (model_utils.py:403-416) and Attention core (:376-389) on a fused
 * qkv tensor [n_hyp][HW][3*heads*dim_head]; out [n_hyp][HW][heads*dim_head]. */
int nope_op_linear_attention(int dtype, const void* qkv, void* out, int n_hyp, int HW, int heads, int dim_head,
                             nope_stream_t s);
int nope_op_attention(int dtype, const void* qkv, void* out, int n_hyp, int HW, int heads, int dim_head,
                      nope_stream_t s);
/* out[m, n] = sum_k act(in[m,k]) * w[n,k] + bias[n]  (f32; act_in: 0 none, 1 SiLU, 2 GELU):
 * pose_mlp (u_net.py:63-72) and ResnetBlock.mlp (model_utils.py:261-265). */
int nope_op_linear(const float* in, const float* w, const float* bias, float* out, int M, int N, int K, int act_in,
                   nope_stream_t s);

/* Dataset-side crop (caller of the hot path): cv2.warpPerspective(img, M, (Wd, Hd)) of crop_frame, src/poses/utils.py:262-270
 * (bilinear, zero border), fused with the loader's image transform (dataloader/shapeNet.py:64-69):
 *   dst[c,y,x] = scale * bilinear(src, Minv (x,y,1)) + shift.   src (Hs,Ws,C) uint8 (src_is_u8 = 1, or 2: the interpolated value
 *   is rounded and clamped to [0, 255] first, as cv2's uint8 destination does) or f32 (src_is_u8 = 0), HWC;
 *   minv9_host: HOST pointer to the 3x3 inverse map, row-major; dst (C,Hd,Wd) f32. */
int nope_op_warp_perspective(const void* src, int src_is_u8, int Hs, int Ws, int C, const float* minv9_host, float* dst_chw, int Hd, int Wd,
                             float scale, float shift, nope_stream_t s);
/* Token-space operators of the LDM variant (ldm/attention.py), tokens = NHWC pixels [M][C]:
 * LayerNorm over C (:210-212); GEGLU in [M][2D] -> out [M][D] (:37-44); softmax self-attention over the N tokens of each
 * sample on a fused [n][N][3C] q|k|v tensor, heads of 32 channels (:168-189).  dtype = a storage code; nope_op_token_attention also takes the
 * compute tags NOPE_BF16X3 / NOPE_F16X2 (f32 tensors, every product as three bf16 MFMA passes over (hi, lo) splits: what the LDM runtime
 * launches in those modes; NOPE_F32 = all-f32 VALU arithmetic, the parity mode). */
int nope_op_layer_norm(int dtype, const void* x, void* y, const float* gamma, const float* beta, int64_t M, int C, float eps, nope_stream_t s);
int nope_op_geglu(int dtype, const void* in, void* out, int64_t M, int D, nope_stream_t s);
int nope_op_token_attention(int dtype, const void* qkv, void* out, int n, int N, int C, int dim_head, nope_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* NOPE_HIP_H */
