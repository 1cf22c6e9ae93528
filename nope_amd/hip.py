"""ctypes binding of libnope_hip.so (include/nope_hip.h) for torch tensors.

PyTorch is plumbing here: it owns device memory and the stream; every computation below is
a hand-written gfx950 kernel behind the C ABI.  There is NO fallback: if the library is
missing (not built) the import of anything that computes fails with a clear error.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

F32, BF16, F16, BF16X3 = 0, 1, 2, 3      # BF16X3: compute mode only (f32 storage, three bf16 MFMA passes per product)
F16X2 = 4                                # compute mode only: BF16X3, but the ping-pong launches (tap-resident 3x3, per-tap 1x1 / up / down) run one f16 + one MX-fp8 MFMA pass (include/nope_hip.h)
ABI_VERSION = 6                          # NOPE_ABI_VERSION of include/nope_hip.h these ctypes structs mirror
CONV_PLAIN, CONV_UP2, CONV_DOWN2, CONV_UP2P, CONV_STRIDE2 = 0, 1, 2, 3, 4
ERR_RANGE, ERR_RANGE_F16 = -7, -8        # nope_unet_x2_range_check (include/nope_hip.h)

_HERE = os.path.dirname(os.path.abspath(__file__))
# NOPE_HIP_LIB: load another build of the SAME gfx950 library (A/B timing of kernel variants); default = in-tree build
LIB_PATH = os.environ.get("NOPE_HIP_LIB") or os.path.join(_HERE, "csrc", "libnope_hip.so")

_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", _vp), ("ndim", _i), ("shape", _i64 * 4)]


class UNetConfig(C.Structure):
    _fields_ = [("u_net_dim", _i), ("channels", _i), ("out_dim", _i), ("pose_dim", _i), ("n_levels", _i),
                ("dim_mults", _i * 8), ("groups", _i), ("heads", _i), ("dim_head", _i), ("pose_mlp_layers", _i),
                ("compute_dtype", _i), ("soft_up_down", _i)]


class LdmConfig(C.Structure):
    _fields_ = [("in_channels", _i), ("model_channels", _i), ("out_channels", _i), ("num_res_blocks", _i), ("n_levels", _i),
                ("channel_mult", _i * 8), ("attn_levels", _i * 8), ("num_head_channels", _i), ("context_dim", _i), ("pose_dim", _i),
                ("pose_mlp_layers", _i), ("injecting_condition_twice", _i), ("compute_dtype", _i), ("use_scale_shift_norm", _i),
                ("transformer_depth", _i)]


class ConvLaunchInfo(C.Structure):
    _fields_ = [("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double), ("kernel", _i), ("mode", _i), ("ntaps", _i), ("Cin", _i),
                ("Cout", _i), ("Hs", _i), ("Ws", _i), ("n_hyp", _i), ("mfma_passes", _i), ("posmajor", _i)]


CONV_KERNEL_NAMES = ("conv_gemm_kernel", "conv_gemm_dma_kernel", "conv_gemm_pp_kernel", "conv3x3_halo_kernel", "conv_gemm_small_kernel", "conv1x1_stream_kernel")


class EncoderConfig(C.Structure):
    _fields_ = [("descriptor_size", _i), ("compute_dtype", _i), ("bn_eps", C.c_float)]


_PROTOS = {
    "nope_strerror": (C.c_char_p, [_i]),
    "nope_abi_version": (_i, []),
    "nope_tuning_reload": (None, []),
    "nope_similarity": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i64, _i, _vp]),
    "nope_topk": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "nope_gather_topk": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "nope_topk_merge": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "nope_op_geodesic": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "nope_unet_create": (_i, [C.POINTER(UNetConfig), C.POINTER(TensorDesc), _i, _vp, C.POINTER(_vp)]),
    "nope_unet_destroy": (None, [_vp]),
    "nope_unet_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "nope_unet_forward": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "nope_unet_graph_limit": (_i, [_vp, C.c_longlong]),
    "nope_unet_graph_replays": (_i, [_vp]),
    "nope_unet_x2_range_check": (_i, [_vp, _vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_float)]),
    "nope_unet_x2_poll": (_i, [_vp, _vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_float)]),
    "nope_unet_x2_enable": (_i, [_vp, _i]),
    "nope_unet_x2_shifts": (_i, [_vp, C.POINTER(_i), _i, C.POINTER(_i)]),
    "nope_ldm_x2_range_check": (_i, [_vp, _vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_float)]),
    "nope_ldm_x2_poll": (_i, [_vp, _vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_float)]),
    "nope_ldm_x2_enable": (_i, [_vp, _i]),
    "nope_unet_profile": (_i, [_vp, _i]),
    "nope_unet_profile_read": (_i, [_vp, C.POINTER(_i), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "nope_unet_profile_launches": (_i, [_vp, C.POINTER(ConvLaunchInfo), _i, C.POINTER(_i)]),
    "nope_op_nchw_to_nhwc": (_i, [_i, _vp, _vp, _i, _i, _i, _vp]),
    "nope_op_nhwc_to_nchw": (_i, [_i, _vp, _vp, _i, _i, _i, _vp]),
    "nope_op_pack_conv_weight": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "nope_encoder_create": (_i, [C.POINTER(EncoderConfig), C.POINTER(TensorDesc), _i, _vp, C.POINTER(_vp)]),
    "nope_encoder_destroy": (None, [_vp]),
    "nope_encoder_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "nope_encoder_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "nope_op_conv": (_i, [_i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "nope_op_conv_ws": (_i, [_i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "nope_op_conv_splitk_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "nope_op_stem_conv": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "nope_op_gn_chunks": (_i, [_i, _i, _i]),
    "nope_op_group_norm": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "nope_op_linear_attention": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "nope_op_attention": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "nope_op_linear": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "nope_ldm_create": (_i, [C.POINTER(LdmConfig), C.POINTER(TensorDesc), _i, _vp, C.POINTER(_vp)]),
    "nope_ldm_destroy": (None, [_vp]),
    "nope_ldm_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "nope_ldm_forward": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "nope_op_warp_perspective": (_i, [_vp, _i, _i, _i, _i, C.POINTER(C.c_float), _vp, _i, _i, C.c_float, C.c_float, _vp]),
    "nope_op_layer_norm": (_i, [_i, _vp, _vp, _vp, _vp, _i64, _i, C.c_float, _vp]),
    "nope_op_geglu": (_i, [_i, _vp, _vp, _i64, _i, _vp]),
    "nope_op_token_attention": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)


class NopeError(RuntimeError):
    pass


class NopeLib:
    """A loaded C-ABI library with typed prototypes."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise NopeError(
                f"{path} not found: the gfx950 library is not built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (needs hipcc). nope_amd has no CPU fallback.")
        self.path = path
        self.dll = C.CDLL(path)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(self.dll, name)
            fn.restype = res
            fn.argtypes = args
        got = self.dll.nope_abi_version()
        if got != ABI_VERSION:     # a stale build would read the config structs below past their end
            raise NopeError(f"{path} has ABI version {got}, this binding was written for {ABI_VERSION}: rebuild the library")

    def check(self, code: int, what: str):
        if code != 0:
            raise NopeError(f"{what}: {self.dll.nope_strerror(code).decode()} ({code})")


_lib: Optional[NopeLib] = None
_tuning_seen: Optional[tuple] = None


def lib() -> NopeLib:
    """The loaded library.  Every binding comes through here, so this is also where a change of the NOPE_* tuning variables since the last call
    is noticed: the library caches them per call site (nope_tuning_reload, include/nope_hip.h)."""
    global _lib, _tuning_seen
    if _lib is None:
        _lib = NopeLib(LIB_PATH)
    cur = tuple(sorted(kv for kv in os.environ.items() if kv[0].startswith("NOPE_")))
    if cur != _tuning_seen:
        _tuning_seen = cur
        _lib.dll.nope_tuning_reload()
    return _lib


def _set_library_for_testing(l: Optional[NopeLib]):
    """tests/ only: lets the CPU test-suite run the same host code over tests/hipemu (a build of the same C ABI
    that takes host pointers)."""
    global _lib, _tuning_seen
    if l is not None:
        l.host_pointers = True
    _lib = l
    _tuning_seen = None


def require_device(t: torch.Tensor):
    """The product library takes device pointers only: a host tensor is an error, never a detour through torch CPU ops."""
    l = lib()      # (always: every binding starts here, and lib() is where a changed NOPE_* tuning variable is noticed)
    if not t.is_cuda and not getattr(l, "host_pointers", False):
        raise NopeError("nope_amd computes on the GPU only: pass CUDA (ROCm) tensors; there is no CPU path")


class overlap_stream:
    """`with overlap_stream(t) as ov: y = f(t)` issues f on a second HIP stream ordered after the current one;
    `ov.join(y)` makes the current stream wait for it.  (Host tensors -- the interpreter build of the tests -- have no
    streams: the body simply runs in place.)"""
    _streams: Dict[str, "torch.cuda.Stream"] = {}

    def __init__(self, t: torch.Tensor, wait_current: bool = True):
        """wait_current=False: the side work does NOT wait for what the current stream has queued -- the caller vouches that the inputs of the
        body are complete (PoseConditional.pipeline_encoders: the encoder passes of the next query may then run under the previous query's
        U-Net); the side stream itself stays in order."""
        self.dev = t.device if t.is_cuda else None
        self.wait_current = wait_current

    def __enter__(self):
        if self.dev is None:
            return self
        self.cur = torch.cuda.current_stream(self.dev)
        side = self._streams.get(str(self.dev))
        if side is None:
            side = self._streams[str(self.dev)] = torch.cuda.Stream(device=self.dev)
        self.side = side
        if self.wait_current:
            side.wait_stream(self.cur)              # whatever produced the inputs is ordered before the side work
        self._ctx = torch.cuda.stream(side)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.dev is not None:
            self._ctx.__exit__(*exc)
        return False

    def mark(self):
        """An event on the side stream at this point of the body (for join_at: the current stream then waits only for the work before it)."""
        return self.side.record_event() if self.dev is not None else None

    def join_at(self, event, *tensors):
        if self.dev is not None:
            self.cur.wait_event(event)
            for t in tensors:
                t.record_stream(self.cur)
                t.record_stream(self.side)

    def join(self, *tensors):
        if self.dev is not None:
            self.cur.wait_stream(self.side)
            for t in tensors:          # used on both streams, whichever of them it was allocated on
                t.record_stream(self.cur)
                t.record_stream(self.side)


def _stream(t: torch.Tensor) -> int:
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return 0


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def torch_dtype(dt: int) -> torch.dtype:
    """torch dtype of tensors STORED under dtype code dt (BF16X3 keeps f32 activations)."""
    return {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16, BF16X3: torch.float32, F16X2: torch.float32}[dt]


def storage_code(dt: int) -> int:
    """dtype code the non-conv operators see for tensors of compute mode dt."""
    return F32 if dt in (BF16X3, F16X2) else dt


def dtype_code(dt) -> int:
    if dt in (F32, "f32", "fp32", "float32", torch.float32):
        return F32
    if dt in (BF16, "bf16", "bfloat16", torch.bfloat16):
        return BF16
    if dt in (F16, "f16", "fp16", "float16", "half", torch.float16):
        return F16
    if dt in (BF16X3, "bf16x3"):
        return BF16X3
    if dt in (F16X2, "f16x2"):
        return F16X2
    raise NopeError(f"unsupported dtype {dt!r}")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# --------------------------------------------------------------------------------------------
# scoring
# --------------------------------------------------------------------------------------------
def similarity(q: torch.Tensor, bank: torch.Tensor, out: Optional[torch.Tensor] = None,
               col_offset: int = 0) -> torch.Tensor:
    """score[b,n] of model.py:257-262.  q (B,C,H,W) f32; bank (B|1,N,C,H,W) f32|bf16.
    With `out` (B, Ntotal) given, writes columns [col_offset, col_offset+N)."""
    require_device(q)
    q = _f32c(q)
    B, Cc, H, W = q.shape
    if bank.dim() != 5 or tuple(bank.shape[2:]) != (Cc, H, W) or bank.shape[0] not in (1, B):
        raise NopeError(f"bank shape {tuple(bank.shape)} does not match query {tuple(q.shape)}")
    if bank.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        bank = bank.float()
    bank = bank.contiguous()
    N = bank.shape[1]
    if N == 0:
        return torch.empty((B, 0), dtype=torch.float32, device=q.device) if out is None else out
    stride_b = 0 if (bank.shape[0] == 1 and B > 1) else N * Cc * H * W
    if out is None:
        out = torch.empty((B, N), dtype=torch.float32, device=q.device)
        col_offset = 0
    assert out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] == B
    ld = out.shape[1]
    assert col_offset + N <= ld
    l = lib()
    l.check(l.dll.nope_similarity(_ptr(q), _ptr(bank), dtype_code(bank.dtype), out.data_ptr() + 4 * col_offset, B, N, Cc,
                                  H, W, stride_b, ld, _stream(q)), "nope_similarity")
    return out


def topk(scores: torch.Tensor, k: int = 5) -> Tuple[torch.Tensor, torch.Tensor]:
    require_device(scores)
    scores = _f32c(scores)
    B, N = scores.shape
    idx = torch.empty((B, k), dtype=torch.int64, device=scores.device)
    vals = torch.empty((B, k), dtype=torch.float32, device=scores.device)
    l = lib()
    l.check(l.dll.nope_topk(_ptr(scores), _ptr(idx), _ptr(vals), B, N, k, N, _stream(scores)), "nope_topk")
    return vals, idx


def gather_topk(gathered: torch.Tensor, n_total: int, k: int = 5) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """gathered (G, B, nmax) f32 = the all-gathered padded score slices of a template-sharded step -> (similarity (B, n_total) owned,
    nearest_idx (B, k) or None for k = 0), one launch (nope_gather_topk)."""
    require_device(gathered)
    G, B, nmax = gathered.shape
    assert gathered.is_contiguous() and gathered.dtype == torch.float32 and nmax == (n_total + G - 1) // G
    sim = torch.empty((B, n_total), dtype=torch.float32, device=gathered.device)
    idx = torch.empty((B, k), dtype=torch.int64, device=gathered.device) if k > 0 else None
    l = lib()
    l.check(l.dll.nope_gather_topk(_ptr(gathered), G, B, n_total, _ptr(sim), _ptr(idx), None, k, _stream(gathered)), "nope_gather_topk")
    return sim, idx


def topk_merge(cand_vals: torch.Tensor, cand_idx: torch.Tensor, k: int = 5) -> Tuple[torch.Tensor, torch.Tensor]:
    """cand_vals / cand_idx (B, M): per-shard top-k lists with global template indices, shards in rank order -> global (vals, idx) (B, k)."""
    require_device(cand_vals)
    cand_vals, cand_idx = _f32c(cand_vals), cand_idx.to(torch.int64).contiguous()
    B, M = cand_vals.shape
    idx = torch.empty((B, k), dtype=torch.int64, device=cand_vals.device)
    vals = torch.empty((B, k), dtype=torch.float32, device=cand_vals.device)
    l = lib()
    l.check(l.dll.nope_topk_merge(_ptr(cand_vals), _ptr(cand_idx), _ptr(idx), _ptr(vals), B, M, k, _stream(cand_vals)), "nope_topk_merge")
    return vals, idx


def op_geodesic(poses: torch.Tensor, gt: torch.Tensor, symmetry: Optional[torch.Tensor] = None,
                idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Geodesic error (radians, f64) of poses[b, idx[b, j]] (or poses[b, j] without `idx`) against gt[b], with the reference's
    symmetry handling (loss.py:14-75).  poses (B|1, N, 3, 3), gt (B, 3, 3), symmetry (B,) / (B, 1) in {0, 1, 2}, idx (B, k) int64.
    Raises ValueError where pytorch3d's so3_rotation_angle does (trace outside [-1 - eps, 3 + eps])."""
    require_device(poses)
    # everything follows the PREDICTIONS' device (a ground-truth pose / symmetry flag / index tensor left on the host is moved, not dereferenced)
    poses = poses.to(torch.float64).contiguous()
    gt = gt.to(device=poses.device, dtype=torch.float64).contiguous()
    B = gt.shape[0]
    if poses.dim() != 4 or tuple(poses.shape[2:]) != (3, 3) or poses.shape[0] not in (1, B) or tuple(gt.shape[1:]) != (3, 3):
        raise NopeError(f"poses {tuple(poses.shape)} / gt {tuple(gt.shape)}: expected (B|1, N, 3, 3) and (B, 3, 3)")
    N = poses.shape[1]
    if idx is not None:
        if idx.dim() != 2 or idx.shape[0] != B:
            raise NopeError(f"idx {tuple(idx.shape)}: expected ({B}, k)")
        idx = idx.to(device=gt.device, dtype=torch.int64).contiguous()
        k = idx.shape[1]
    else:
        k = N
    sym = None if symmetry is None else symmetry.reshape(-1).to(device=gt.device, dtype=torch.int32).contiguous()
    if sym is not None and sym.numel() != B:
        raise NopeError(f"symmetry has {sym.numel()} entries for {B} samples")
    err = torch.empty((B, k), dtype=torch.float64, device=gt.device)
    if B == 0 or k == 0:
        return err
    status = torch.empty(1, dtype=torch.int32, device=gt.device)
    stride_b = 0 if (poses.shape[0] == 1 and B > 1) else N * 9
    l = lib()
    l.check(l.dll.nope_op_geodesic(_ptr(poses), stride_b, N, _ptr(idx), _ptr(gt), _ptr(sym), _ptr(err), _ptr(status), B, k, _stream(gt)),
            "nope_op_geodesic")
    st = int(status.item())
    if st & 2:
        raise NopeError("nope_op_geodesic: an index lies outside the pose grid")
    if st & 1:
        raise ValueError("A matrix has trace outside valid range [-1-eps,3+eps].")      # pytorch3d so3_rotation_angle's message
    return err


def _tensor_descs(state_dict: Dict[str, torch.Tensor]):
    keep, dev = [], None
    descs = (TensorDesc * max(1, len(state_dict)))()
    for i, (k, v) in enumerate(state_dict.items()):
        t = _f32c(v.detach())
        keep.append(t)
        dev = t.device
        descs[i].name = k.encode()
        descs[i].data = t.data_ptr()
        descs[i].ndim = t.dim()
        for j, s in enumerate(t.shape[:4]):
            descs[i].shape[j] = s
    return descs, keep, dev


# --------------------------------------------------------------------------------------------
# Template-encoder handle
# --------------------------------------------------------------------------------------------
class EncoderHandle:
    """Owns a `nope_encoder*` built from the FeatureExtractor's state dict (BatchNorm folded at create time)."""

    def __init__(self, descriptor_size: int, state_dict: Dict[str, torch.Tensor], compute_dtype=F32, bn_eps: float = 1e-5):
        l = lib()
        self._l = l
        c = EncoderConfig()
        c.descriptor_size = int(descriptor_size)
        c.compute_dtype = dtype_code(compute_dtype)
        c.bn_eps = float(bn_eps)
        self.descriptor_size, self.compute_dtype = c.descriptor_size, c.compute_dtype
        sd = {k: v for k, v in state_dict.items() if k.startswith(("backbone.", "projector.")) and v.dtype.is_floating_point}
        descs, keep, dev = _tensor_descs(sd)
        self.device = dev
        h = _vp()
        stream = torch.cuda.current_stream(dev).cuda_stream if dev is not None and dev.type == "cuda" else 0
        l.check(l.dll.nope_encoder_create(C.byref(c), descs, len(sd), stream, C.byref(h)), "nope_encoder_create")
        self._h = h
        self._ws: Dict[tuple, torch.Tensor] = {}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._l.dll.nope_encoder_destroy(h)
            self._h = None

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        """image (B,3,H,W) f32 -> (B,descriptor_size,H/8,W/8) f32."""
        require_device(image)
        image = _f32c(image)
        B, Cc, H, W = image.shape
        if Cc != 3:
            raise NopeError(f"encoder expects 3-channel images, got {tuple(image.shape)}")
        need = int(self._l.dll.nope_encoder_workspace_bytes(self._h, B, H, W))
        if need == 0:
            raise NopeError(f"unsupported encoder input size {H}x{W} (must be multiples of 8)")
        # one workspace per stream: two encoder passes may be in flight on different streams (model.generate_and_retrieve)
        key = (str(image.device), _stream(image))
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=image.device)
        out = torch.empty((B, self.descriptor_size, H // 8, W // 8), dtype=torch.float32, device=image.device)
        self._l.check(self._l.dll.nope_encoder_forward(self._h, _ptr(image), B, H, W, _ptr(out), _ptr(ws), need, _stream(image)),
                      "nope_encoder_forward")
        return out


def op_stem_conv(dt: int, image: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """conv1 7x7/2 pad 3 + per-channel affine + ReLU: image (B,3,H,W) f32 NCHW -> NHWC (B,H/2,W/2,64) of dtype dt."""
    image = _f32c(image)
    B, _, H, W = image.shape
    out = torch.empty((B, H // 2, W // 2, 64), dtype=torch_dtype(dt), device=image.device)
    scratch = torch.empty(147 * 64, dtype=torch.float32, device=image.device)
    l = lib()
    l.check(l.dll.nope_op_stem_conv(dt, _ptr(image), _ptr(_f32c(w)), _ptr(_f32c(scale)), _ptr(_f32c(shift)), _ptr(scratch),
                                    _ptr(out), B, H, W, _stream(image)), "nope_op_stem_conv")
    return out


# --------------------------------------------------------------------------------------------
# NOPE_F16X2 activation ranges: shared by the network handles (include/nope_hip.h: nope_unet_x2_poll)
# --------------------------------------------------------------------------------------------
class _X2RangeMixin:
    _x2_prefix = "unet"

    def _x2_init(self):
        # NOPE_F16X2 activation ranges.  The library judges every forward on the device and overwrites the output of one whose layers left
        # their accurate windows with NaNs (include/nope_hip.h: nope_unet_x2_poll) -- no synchronisation in the step.  range_mode:
        #   "poison"            nothing more: the verdicts that have arrived are read at the start of the next forward (shifts re-centred,
        #                       an event recorded, one warning); a caller that finds NaNs in a bank repeats its call;
        #   "repeat"            after a forward (or, deferred, at the end of the caller's step) synchronise, read the verdict and issue
        #                       the forward again until it is inside its windows: never a NaN, one host synchronisation per step;
        #   "auto" (default)    "repeat" until three forwards in a row were inside their windows without any shift moving -- a network's
        #                       first calls settle its shifts without a NaN -- then "poison"; back to "repeat" when a verdict says so;
        #   "off"               do not look (the device still judges and poisons).
        # NOPE_X2_RANGE_CHECK = 0 / 1 / 2 / 3 selects off / auto / repeat / poison.
        self.range_mode = {"0": "off", "1": "auto", "2": "repeat", "3": "poison"}.get(os.environ.get("NOPE_X2_RANGE_CHECK", "1"), "auto") \
            if self.compute_dtype == F16X2 else "off"
        self._settled = 0                            # "auto": forwards in a row that needed nothing
        self.range_events: List[dict] = []           # one record per forward that had to be repeated
        self._pending: List[tuple] = []              # forwards issued with defer_range_check: (re-launch closure, stream)
        self._warned = False
        self.x2_enabled = self.compute_dtype == F16X2

    def _x2_mode(self) -> str:
        if self.range_mode == "auto":
            return "repeat" if self._settled < 3 else "poison"
        return self.range_mode

    def _x2_before_forward(self, stream):
        if self.range_mode != "off" and self.x2_enabled:
            # verdicts of EARLIER forwards that have reached the host (no waiting): their outputs were NaN; the shifts are re-centred now
            code, bad, moved, amax = self.x2_range_check(stream, sync=False)
            if code != 0 or moved:
                self._settled = 0
            if code != 0:
                self.range_events.append({"code": code, "layers_out_of_range": bad, "layers_adjusted": moved, "max_abs": amax, "attempt": -1})
                if self._x2_mode() == "poison" and not self._warned:
                    import warnings
                    self._warned = True
                    warnings.warn(f"nope_amd f16x2: an earlier U-Net forward saw activations up to {amax:.3g}, outside the accurate range of "
                                  f"{bad} layer(s): its output was overwritten with NaNs (never silently inaccurate); the layers' range shifts "
                                  "are re-centred now -- repeat that call (or use range_mode = 'repeat' / NOPE_X2_RANGE_CHECK=2)", RuntimeWarning)
                if code == ERR_RANGE_F16:
                    self.x2_enable(False)

    def x2_range_check(self, stream, sync: bool = True) -> Tuple[int, int, int, float]:
        """(code, layers out of range, layers whose shift moved, largest |activation|) of the forwards judged since the last look; sync:
        synchronise `stream` first (every forward issued on it is judged), else only the verdicts that have already arrived."""
        bad, moved, amax = _i(0), _i(0), C.c_float(0)
        fn = getattr(self._l.dll, f"nope_{self._x2_prefix}_x2_range_check" if sync else f"nope_{self._x2_prefix}_x2_poll")
        code = int(fn(self._h, stream, C.byref(bad), C.byref(moved), C.byref(amax)))
        if code not in (0, ERR_RANGE, ERR_RANGE_F16):
            self._l.check(code, f"nope_{self._x2_prefix}_x2_range_check")
        return code, bad.value, moved.value, float(amax.value)

    def x2_enable(self, on: bool):
        self._l.check(getattr(self._l.dll, f"nope_{self._x2_prefix}_x2_enable")(self._h, int(bool(on))), f"nope_{self._x2_prefix}_x2_enable")
        self.x2_enabled = bool(on) and self.compute_dtype == F16X2

    def finish_range_check(self) -> bool:
        """Check the forwards issued since the last check (nope_unet_x2_range_check: synchronises their stream); every forward whose layers
        left their accurate window is issued again -- same arguments, re-centred shifts -- until it is inside.  Returns True when anything
        was repeated: work the caller derived from the outputs has to be repeated too."""
        pending, self._pending = self._pending, []
        repeated = False
        for attempt in range(8):        # (a repeated forward can move the maxima of layers downstream of the repaired ones: a few rounds at most)
            if not pending or not (self._x2_mode() == "repeat" and self.x2_enabled):
                break
            code, bad, moved, amax = self.x2_range_check(pending[-1][1])
            if code == 0:
                self._settled = 0 if (moved or repeated) else self._settled + 1
                break
            # a two-pass layer saw activations outside its accurate window: those forwards have plain-f16 accuracy there -- repeat them
            self.range_events.append({"code": code, "layers_out_of_range": bad, "layers_adjusted": moved, "max_abs": amax, "attempt": attempt})
            if code == ERR_RANGE_F16 or attempt == 6:
                import warnings
                warnings.warn(f"nope_amd f16x2: activations up to {amax:.3g} " + ("are not finite" if code == ERR_RANGE_F16 else
                              "keep leaving the layers' windows") + ": this U-Net runs as bf16x3 (three MFMA passes) from now on", RuntimeWarning)
                self.x2_enable(False)
            for launch, _ in pending:
                launch()
            repeated = True
        return repeated


# --------------------------------------------------------------------------------------------
# U-Net handle
# --------------------------------------------------------------------------------------------
class UNetHandle(_X2RangeMixin):
    """Owns a `nope_unet*` built from a reference-keyed state dict."""

    def __init__(self, cfg: dict, state_dict: Dict[str, torch.Tensor], compute_dtype=F32):
        l = lib()
        self._l = l
        c = UNetConfig()
        c.u_net_dim = cfg["u_net_dim"]; c.channels = cfg["channels"]; c.out_dim = cfg.get("out_dim", cfg["channels"])
        c.pose_dim = cfg.get("pose_dim", 6)
        mults = tuple(cfg.get("dim_mults", (1, 2, 4, 8)))
        c.n_levels = len(mults)
        for i, m in enumerate(mults):
            c.dim_mults[i] = m
        c.groups = cfg.get("groups", 8); c.heads = 4; c.dim_head = 32
        c.pose_mlp_layers = cfg.get("pose_mlp_layers", 1)
        c.compute_dtype = dtype_code(compute_dtype)
        c.soft_up_down = int(cfg.get("soft_up_down", 0))
        self.cfg = dict(cfg)
        self.compute_dtype = c.compute_dtype
        self.channels, self.out_dim, self.pose_dim = c.channels, c.out_dim, c.pose_dim
        descs, keep, dev = _tensor_descs(state_dict)
        self.device = dev
        h = _vp()
        stream = torch.cuda.current_stream(dev).cuda_stream if dev is not None and dev.type == "cuda" else 0
        l.check(l.dll.nope_unet_create(C.byref(c), descs, len(state_dict), stream, C.byref(h)), "nope_unet_create")
        self._h = h
        self._ws: Dict[tuple, torch.Tensor] = {}     # one arena per (device, stream): forwards on different streams never share one
        self._x2_init()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._l.dll.nope_unet_destroy(h)
            self._h = None

    def x2_shifts(self) -> List[int]:
        n = _i(0)
        self._l.check(self._l.dll.nope_unet_x2_shifts(self._h, None, 0, C.byref(n)), "nope_unet_x2_shifts")
        buf = (_i * max(1, n.value))()
        self._l.check(self._l.dll.nope_unet_x2_shifts(self._h, buf, n.value, C.byref(n)), "nope_unet_x2_shifts")
        return list(buf[:n.value])

    def profile(self, enable: bool):
        self._l.check(self._l.dll.nope_unet_profile(self._h, int(enable)), "nope_unet_profile")

    def graph_limit(self, max_hyp_pixels: int):
        """Opt in to hipGraph replay for forwards of at most this many n_hyp * H * W (0 = off, the default)."""
        self._l.check(self._l.dll.nope_unet_graph_limit(self._h, int(max_hyp_pixels)), "nope_unet_graph_limit")

    def graph_replays(self) -> int:
        return int(self._l.dll.nope_unet_graph_replays(self._h))

    def profile_read(self):
        """(n_launches, total_ms, total_flops, total_bytes) of the conv-GEMM launches since profile(True)."""
        n, ms, fl, by = _i(0), C.c_double(0), C.c_double(0), C.c_double(0)
        self._l.check(self._l.dll.nope_unet_profile_read(self._h, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), "profile_read")
        return n.value, ms.value, fl.value, by.value

    def profile_launches(self):
        """One dict per conv launch since profile(True), in issue order: kernel name, shape, ms, flops, bytes."""
        n = _i(0)
        self._l.check(self._l.dll.nope_unet_profile_launches(self._h, None, 0, C.byref(n)), "profile_launches")
        buf = (ConvLaunchInfo * max(1, n.value))()
        self._l.check(self._l.dll.nope_unet_profile_launches(self._h, buf, n.value, C.byref(n)), "profile_launches")
        out = []
        for r in buf[:n.value]:
            d = {k: getattr(r, k) for k, _ in ConvLaunchInfo._fields_}
            d["kernel"] = CONV_KERNEL_NAMES[r.kernel]
            out.append(d)
        return out

    def workspace_bytes(self, n_hyp: int, n_src: int, H: int, W: int) -> int:
        return int(self._l.dll.nope_unet_workspace_bytes(self._h, n_hyp, n_src, H, W))

    def forward(self, x: torch.Tensor, pose: torch.Tensor, x_rep: int = 1, out: Optional[torch.Tensor] = None,
                out_dtype=F32, defer_range_check: bool = False) -> torch.Tensor:
        """out[j] = UNet(x[j // x_rep], pose[j]); x (n_src,C,H,W) f32, pose (n_src*x_rep, pose_dim).  defer_range_check (NOPE_F16X2): the
        caller will call finish_range_check() itself before it hands results out."""
        require_device(x)
        x = _f32c(x)
        pose = _f32c(pose)
        n_src, Cc, H, W = x.shape
        n_hyp = pose.shape[0]
        if Cc != self.channels or pose.shape[1] != self.pose_dim or n_src * x_rep != n_hyp:
            raise NopeError(f"shape mismatch: x {tuple(x.shape)}, pose {tuple(pose.shape)}, x_rep {x_rep}")
        odt = dtype_code(out_dtype)
        if out is None:
            out = torch.empty((n_hyp, self.out_dim, H, W), dtype=torch_dtype(odt), device=x.device)
        assert out.is_contiguous() and out.numel() == n_hyp * self.out_dim * H * W and out.dtype == torch_dtype(odt)
        need = self.workspace_bytes(n_hyp, n_src, H, W)
        if need == 0:
            raise NopeError(f"unsupported U-Net problem size n_hyp={n_hyp} H={H} W={W}")
        key = (str(x.device), _stream(x))
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            self._ws.pop(key, None)          # release the smaller arena before taking a bigger one
            ws = None
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=x.device)
        def launch():
            self._l.check(self._l.dll.nope_unet_forward(self._h, _ptr(x), n_src, x_rep, _ptr(pose), n_hyp, H, W, _ptr(out), odt,
                                                        _ptr(ws), ws.numel(), _stream(x)), "nope_unet_forward")
        self._x2_before_forward(_stream(x))
        launch()
        if self._x2_mode() == "repeat" and self.x2_enabled:
            # the check needs the forward to have finished: callers that go on issuing work on the stream (PoseConditional: scoring,
            # top-k) call finish_range_check() at the END of their step -- one synchronisation where the results are read anyway --
            # and repeat their own tail when it says the forward was repeated
            self._pending.append((launch, _stream(x)))
            if not defer_range_check:
                self.finish_range_check()
        return out


# --------------------------------------------------------------------------------------------
# LDM cross-attention U-Net handle
# --------------------------------------------------------------------------------------------
class LdmHandle(_X2RangeMixin):
    _x2_prefix = "ldm"

    """Owns a `nope_ldm*` built from a UNetModelPose state dict (reference keys)."""

    def __init__(self, cfg: dict, state_dict: Dict[str, torch.Tensor], compute_dtype=F32):
        l = lib()
        self._l = l
        c = LdmConfig()
        for k in ("in_channels", "model_channels", "out_channels", "num_res_blocks", "num_head_channels", "context_dim", "pose_dim",
                  "pose_mlp_layers", "injecting_condition_twice"):
            setattr(c, k, int(cfg[k]))
        mult = tuple(cfg["channel_mult"])
        c.n_levels = len(mult)
        for i, m in enumerate(mult):
            c.channel_mult[i] = m
            c.attn_levels[i] = int(cfg["attn_levels"][i])
        c.compute_dtype = dtype_code(compute_dtype)
        c.use_scale_shift_norm = int(cfg.get("use_scale_shift_norm", 0))
        c.transformer_depth = int(cfg.get("transformer_depth", 1))
        self.in_channels, self.out_channels, self.pose_dim = c.in_channels, c.out_channels, c.pose_dim
        descs, keep, dev = _tensor_descs(state_dict)
        self.device = dev
        h = _vp()
        stream = torch.cuda.current_stream(dev).cuda_stream if dev is not None and dev.type == "cuda" else 0
        l.check(l.dll.nope_ldm_create(C.byref(c), descs, len(state_dict), stream, C.byref(h)), "nope_ldm_create")
        self._h = h
        self._ws: Dict[tuple, torch.Tensor] = {}
        self.compute_dtype = c.compute_dtype
        self._x2_init()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._l.dll.nope_ldm_destroy(h)
            self._h = None

    def forward(self, x: torch.Tensor, pose: torch.Tensor, x_rep: int = 1, out: Optional[torch.Tensor] = None, out_dtype=F32) -> torch.Tensor:
        require_device(x)
        x, pose = _f32c(x), _f32c(pose)
        n_src, Cc, H, W = x.shape
        n_hyp = pose.shape[0]
        if Cc != self.in_channels or pose.shape[1] != self.pose_dim or n_src * x_rep != n_hyp:
            raise NopeError(f"shape mismatch: x {tuple(x.shape)}, pose {tuple(pose.shape)}, x_rep {x_rep}")
        odt = dtype_code(out_dtype)
        if out is None:
            out = torch.empty((n_hyp, self.out_channels, H, W), dtype=torch_dtype(odt), device=x.device)
        assert out.is_contiguous() and out.numel() == n_hyp * self.out_channels * H * W and out.dtype == torch_dtype(odt)
        need = int(self._l.dll.nope_ldm_workspace_bytes(self._h, n_hyp, n_src, H, W))
        if need == 0:
            raise NopeError(f"unsupported LDM U-Net problem size n_hyp={n_hyp} H={H} W={W}")
        key = (str(x.device), _stream(x))
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            self._ws.pop(key, None)
            ws = None
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=x.device)
        def launch():
            self._l.check(self._l.dll.nope_ldm_forward(self._h, _ptr(x), n_src, x_rep, _ptr(pose), n_hyp, H, W, _ptr(out), odt,
                                                       _ptr(ws), ws.numel(), _stream(x)), "nope_ldm_forward")
        self._x2_before_forward(_stream(x))          # NOPE_F16X2: verdicts of earlier forwards (no waiting), _X2RangeMixin
        launch()
        if self._x2_mode() == "repeat" and self.x2_enabled:
            self._pending.append((launch, _stream(x)))
            self.finish_range_check()
        return out


def op_warp_perspective(img: torch.Tensor, minv, size: int, scale: float = 1.0, shift: float = 0.0, round_u8: bool = False) -> torch.Tensor:
    """img (H,W,C) uint8 or f32 on the device, minv 3x3 (host) mapping output pixels to source pixels -> (C,size,size) f32.
    round_u8 (uint8 sources): round + clamp the interpolated value to [0, 255] before scale / shift (a uint8 destination image)."""
    require_device(img)
    img = img.contiguous()
    if img.dtype not in (torch.uint8, torch.float32):
        img = img.float()
    H, W, Cc = img.shape
    out = torch.empty((Cc, size, size), dtype=torch.float32, device=img.device)
    m = (C.c_float * 9)(*[float(v) for v in list(minv.reshape(-1))])
    l = lib()
    l.check(l.dll.nope_op_warp_perspective(_ptr(img), (2 if round_u8 else 1) if img.dtype == torch.uint8 else 0, H, W, Cc, m, _ptr(out), size, size, scale, shift,
                                           _stream(img)), "nope_op_warp_perspective")
    return out


def op_layer_norm(dt: int, x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """x (..., C) of dtype dt: LayerNorm over the last axis."""
    y = torch.empty_like(x)
    l = lib()
    l.check(l.dll.nope_op_layer_norm(dt, _ptr(x), _ptr(y), _ptr(_f32c(gamma)), _ptr(_f32c(beta)), x.numel() // x.shape[-1], x.shape[-1], eps,
                                     _stream(x)), "nope_op_layer_norm")
    return y


def op_geglu(dt: int, x: torch.Tensor) -> torch.Tensor:
    """x (..., 2D) = [a | gate] -> a * gelu(gate), (..., D)."""
    D = x.shape[-1] // 2
    y = torch.empty((*x.shape[:-1], D), dtype=x.dtype, device=x.device)
    l = lib()
    l.check(l.dll.nope_op_geglu(dt, _ptr(x), _ptr(y), x.numel() // x.shape[-1], D, _stream(x)), "nope_op_geglu")
    return y


def op_token_attention(dt: int, qkv: torch.Tensor, dim_head: int = 32) -> torch.Tensor:
    """qkv (n, N, 3C) -> softmax(q k^T / sqrt(d)) v per head of `dim_head` channels, (n, N, C)."""
    n, N, c3 = qkv.shape
    out = torch.empty((n, N, c3 // 3), dtype=qkv.dtype, device=qkv.device)
    l = lib()
    l.check(l.dll.nope_op_token_attention(dt, _ptr(qkv), _ptr(out), n, N, c3 // 3, dim_head, _stream(qkv)), "nope_op_token_attention")
    return out


# --------------------------------------------------------------------------------------------
# operator-level wrappers (parity tests of single blocks; NCHW f32 in/out at the boundary)
# --------------------------------------------------------------------------------------------
def to_nhwc(x: torch.Tensor, dt: int) -> torch.Tensor:
    x = _f32c(x)
    n, c, h, w = x.shape
    y = torch.empty((n, h, w, c), dtype=torch_dtype(dt), device=x.device)
    l = lib()
    l.check(l.dll.nope_op_nchw_to_nhwc(storage_code(dt), _ptr(x), _ptr(y), n, c, h * w, _stream(x)), "nchw_to_nhwc")
    return y


def to_nchw(y: torch.Tensor, dt: int) -> torch.Tensor:
    n, h, w, c = y.shape
    x = torch.empty((n, c, h, w), dtype=torch.float32, device=y.device)
    l = lib()
    l.check(l.dll.nope_op_nhwc_to_nchw(storage_code(dt), _ptr(y), _ptr(x), n, c, h * w, _stream(y)), "nhwc_to_nchw")
    return x


def pack_conv_weight(w: torch.Tensor, dt: int, mode: int = CONV_PLAIN) -> Tuple[torch.Tensor, int, int]:
    w = _f32c(w)
    cout = w.shape[0]
    if mode == CONV_DOWN2:
        cin, ntaps = w.shape[1] // 4, 4
    elif mode == CONV_UP2P:
        cin, ntaps = w.shape[1], 4
    else:
        cin, ntaps = w.shape[1], w.shape[2] * w.shape[3]
    if dt == F16X2:      # the f16 + MX-fp8 tile's layout (ping-pong kernels): 4 bytes per weight + a 16-byte tail with the layer's block scale
        out = torch.empty((4 if mode == CONV_UP2P else 1) * cout * ntaps * cin + 4, dtype=torch.float32, device=w.device)
    else:
        out = torch.empty((4 if mode == CONV_UP2P else 1, cout, ntaps, cin), dtype=torch_dtype(dt), device=w.device)
    l = lib()
    l.check(l.dll.nope_op_pack_conv_weight(dt, _ptr(w), _ptr(out), cout, cin, ntaps, mode, _stream(w)), "pack_conv_weight")
    return out, cin, ntaps


def op_conv(dt: int, src1: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
            src2: Optional[torch.Tensor] = None, mode: int = CONV_PLAIN, rep1: int = 1, rep2: int = 1,
            resid: Optional[torch.Tensor] = None, n_hyp: Optional[int] = None, out_nchw: bool = False,
            out_dtype: int = F32, act_relu: bool = False, split_k: bool = False, x2_shift: int = 0) -> torch.Tensor:
    """src* NHWC tensors of dtype dt; w torch Conv2d weight (f32).  Returns NHWC (or NCHW).  split_k: hand the launcher the scratch it
    asks for (as the U-Net / encoder runtimes do), so shapes it would split along K (few tiles, long K) are.  x2_shift (F16X2 only): the
    activation range shift t of the packed layer (word 3 of the pack's tail; include/nope_hip.h: full accuracy for 2^(t-4) <= |a| <= 1792 2^t)."""
    pw, cin, ntaps = pack_conv_weight(w, dt, mode)
    if x2_shift:
        assert dt == F16X2
        pw.view(torch.int32)[-1] = int(x2_shift)
    n1, hs, ws, c1 = src1.shape
    c2 = 0 if src2 is None else src2.shape[3]
    assert c1 + c2 == cin
    n_hyp = n_hyp if n_hyp is not None else n1 * rep1
    ho, wo = (2 * hs, 2 * ws) if mode in (CONV_UP2, CONV_UP2P) else ((hs // 2, ws // 2) if mode in (CONV_DOWN2, CONV_STRIDE2) else (hs, ws))
    cout = w.shape[0]
    if out_nchw:
        out = torch.empty((n_hyp, cout, ho, wo), dtype=torch_dtype(out_dtype), device=src1.device)
    else:
        out = torch.empty((n_hyp, ho, wo, cout), dtype=torch_dtype(dt), device=src1.device)
    b = None if bias is None else _f32c(bias)
    l = lib()
    scratch, sk = None, 0
    if split_k:
        sk = int(l.dll.nope_op_conv_splitk_bytes(dt, c1, c2, rep1, hs, ws, mode, ntaps, cout, n_hyp))
        if sk:
            scratch = torch.empty(sk, dtype=torch.uint8, device=src1.device)
    l.check(l.dll.nope_op_conv_ws(dt, _ptr(src1), c1, rep1, _ptr(src2), c2, rep2, hs, ws, mode, ntaps, _ptr(pw), _ptr(b),
                                  _ptr(resid), _ptr(out), cout, n_hyp, int(out_nchw), out_dtype, int(act_relu), _ptr(scratch), sk,
                                  _stream(src1)), "nope_op_conv")
    return out


def op_group_norm(dt: int, x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, act_silu: bool = False,
                  emb: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, h, w, c = x.shape
    l = lib()
    dt = storage_code(dt)
    nch = l.dll.nope_op_gn_chunks(dt, h * w, c)
    partial = torch.empty((n, nch, groups, 2), dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    g, b = _f32c(gamma), _f32c(beta)
    e = None if emb is None else _f32c(emb)
    l.check(l.dll.nope_op_group_norm(dt, _ptr(x), _ptr(y), _ptr(partial), _ptr(g), _ptr(b), n, h * w, c, groups,
                                     int(act_silu), _ptr(e), 0 if e is None else e.shape[1], _ptr(resid), _stream(x)),
            "nope_op_group_norm")
    return y


def op_linear_attention(dt: int, qkv: torch.Tensor, heads: int = 4, dim_head: int = 32, full: bool = False) -> torch.Tensor:
    n, h, w, c3 = qkv.shape
    out = torch.empty((n, h, w, heads * dim_head), dtype=qkv.dtype, device=qkv.device)
    l = lib()
    fn = l.dll.nope_op_attention if full else l.dll.nope_op_linear_attention
    l.check(fn(storage_code(dt), _ptr(qkv), _ptr(out), n, h * w, heads, dim_head, _stream(qkv)), "nope_op_attention")
    return out


def op_linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act_in: int = 0) -> torch.Tensor:
    x, w = _f32c(x), _f32c(w)
    b = None if bias is None else _f32c(bias)
    out = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
    l = lib()
    l.check(l.dll.nope_op_linear(_ptr(x), _ptr(w), _ptr(b), _ptr(out), x.shape[0], w.shape[0], x.shape[1], act_in,
                                 _stream(x)), "nope_op_linear")
    return out
