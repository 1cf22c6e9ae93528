"""Template-axis sharding of the bank across the GPUs of one node (SURVEY.md §8 e).

The reference has no collective on its inference path (its DDP shards *queries*,
configs/machine/trainer/local.yaml:9).  Here every rank holds the full weights and all B
query/reference embeddings, generates and scores its own contiguous slice of the N
templates -- the bank slice (16-32 KB per hypothesis) never leaves local HBM -- and the only
exchange is one all-gather of the (B, N/G) f32 scores (<= 1 MiB in total), after which every
rank holds the full (B, N) similarity that `retrieval` returns and the harness saves
(model.py:323,369-375).  One process per GPU, `torch.distributed` backend "nccl" (= RCCL over
xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


class ShardedBank(torch.Tensor):
    """This rank's slice [lo, hi) of the template axis of a (B, n_total, C, h, w) bank: what `PoseConditional.generate_templates`
    returns under `template_parallel` with more than one rank.  An ordinary tensor of the local slice (same storage: kernels
    write into it, views and `data_ptr()` work) that carries its placement as part of its TYPE -- any torch op on it returns a
    plain `torch.Tensor` (torch-function dispatch is disabled for the subclass), i.e. a copy, a slice or a concatenation is no
    longer a placed shard and `retrieval` refuses it unless told where it sits (`shard=`)."""

    @staticmethod
    def __new__(cls, local: torch.Tensor, lo: int, hi: int, n_total: int):
        if local.dim() != 5 or local.shape[1] != hi - lo or not (0 <= lo <= hi <= n_total):
            raise ValueError(f"shard [{lo}, {hi}) of {n_total} does not describe a local bank of shape {tuple(local.shape)}")
        t = torch.Tensor._make_subclass(cls, local, False)
        t.lo, t.hi, t.n_total = int(lo), int(hi), int(n_total)
        return t

    __torch_function__ = torch._C._disabled_torch_function_impl

    @property
    def shard(self) -> Tuple[int, int, int]:
        return self.lo, self.hi, self.n_total


def world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced split of range(n): first (n % world) ranks get one extra."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


_BUFS = {}


def gather_buffers(B: int, n_total: int, device, group=None):
    """Cached (send (B, nmax), recv (G, B, nmax)) f32 buffers of the score all-gather, nmax = ceil(n_total / G).
    The scoring kernel writes this rank's columns straight into `send` (nope_amd.model.retrieval_from_feat), so a
    step allocates nothing and copies nothing on the way into the collective."""
    _, ws = world(group)
    nmax = (n_total + ws - 1) // ws
    key = (B, nmax, ws, str(device))
    b = _BUFS.get(key)
    if b is None:
        if len(_BUFS) > 8:
            _BUFS.clear()
        b = _BUFS[key] = (torch.zeros((B, nmax), dtype=torch.float32, device=device),
                          torch.empty((ws, B, nmax), dtype=torch.float32, device=device))
    return b


def _gather_padded(send: torch.Tensor, recv: torch.Tensor, group=None):
    """one all_gather_into_tensor of the padded (B, nmax) slices into (G, B, nmax)"""
    ws, B, nmax = recv.shape
    if send.is_cuda and dist.get_backend(group) == "gloo":
        # gloo is a host backend (the multi-rank-on-one-GPU test): stage through host memory.  Production runs use
        # "nccl" (= RCCL), which gathers device buffers directly over xGMI.
        h_send, h_recv = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(h_recv.view(ws * B, nmax), h_send, group=group)
        recv.copy_(h_recv)
    else:
        dist.all_gather_into_tensor(recv.view(ws * B, nmax), send, group=group)


def all_gather_scores_topk(n_local: int, n_total: int, B: int, device, k: int = 5, group=None):
    """The tail of a template-sharded step: this rank's scores already sit in the cached send buffer (gather_buffers; the scoring kernel
    wrote them there).  ONE collective + ONE kernel: all_gather_into_tensor of the padded slices, then nope_gather_topk drops the padding
    into an owned (B, n_total) similarity and ranks it.  Returns (similarity, nearest_idx)."""
    from . import hip
    _, ws = world(group)
    send, recv = gather_buffers(B, n_total, device, group)
    _gather_padded(send, recv, group)
    return hip.gather_topk(recv, n_total, k)


def all_gather_topk_pairs(local_vals: torch.Tensor, local_idx: torch.Tensor, k: int, group=None):
    """north_star's "all-gather of per-shard top-k": local_vals / local_idx (B, k_local <= k) with GLOBAL template indices -> global
    (vals, idx) (B, k) on every rank: one all-gather of (B, k) (value, index) pairs -- 12 k bytes per query and rank instead of the 4 N / G of
    the full score slices -- and one merge kernel (nope_topk_merge: descending score, ties -> lowest global index, model.py:265).  For callers
    that do not keep the full similarity."""
    from . import hip
    rank, ws = world(group)
    B, kl = local_vals.shape
    dev = local_vals.device
    pv = torch.full((B, k), float("-inf"), dtype=torch.float32, device=dev)
    pi = torch.full((B, k), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
    if kl:
        pv[:, :kl], pi[:, :kl] = local_vals, local_idx
    if ws == 1:
        return hip.topk_merge(pv, pi, k)
    # one collective: values and indices travel as one int64 tensor (the f32 bits in the low word of a second int64 column block)
    pack = torch.cat((pv.view(torch.int32).to(torch.int64), pi), 1)              # (B, 2 k)
    out = torch.empty((ws, B, 2 * k), dtype=torch.int64, device=dev)
    if pack.is_cuda and dist.get_backend(group) == "gloo":
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h.view(ws * B, 2 * k), pack.cpu(), group=group)
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out.view(ws * B, 2 * k), pack, group=group)
    out = out.permute(1, 0, 2)                                                   # (B, ws, 2 k): shards in rank order
    cv = out[:, :, :k].to(torch.int32).view(torch.float32).reshape(B, ws * k).contiguous()
    ci = out[:, :, k:].reshape(B, ws * k).contiguous()
    return hip.topk_merge(cv, ci, k)


def all_gather_scores(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """local (B, n_local) f32 slice of this rank (n_local may be 0) -> (B, n_total) on every rank: ONE
    `all_gather_into_tensor` of the padded (B, nmax) slices (RCCL on GPUs, gloo in the CPU tests) and at most two
    strided copies to drop the padding of an uneven split."""
    rank, ws = world(group)
    if ws == 1:
        assert local.shape[1] == n_total
        return local
    B = local.shape[0]
    send, recv = gather_buffers(B, n_total, local.device, group)
    nmax = send.shape[1]
    if not (local.data_ptr() == send.data_ptr() and local.stride() == send.stride()) and local.shape[1] > 0:
        send[:, : local.shape[1]].copy_(local)
    _gather_padded(send, recv, group)
    base, extra = divmod(n_total, ws)
    out = local.new_empty((B, n_total))            # always an OWNED tensor: `recv` is a cached buffer the next gather overwrites
    if extra == 0:                                 # (for B == 1 a reshape of the permuted buffer would be a view of it)
        out.view(B, ws, nmax).copy_(recv.permute(1, 0, 2))
        return out
    cut = extra * (base + 1)                       # the first `extra` ranks hold base + 1 columns each
    out[:, :cut].view(B, extra, base + 1).copy_(recv[:extra].permute(1, 0, 2))
    if base > 0:
        out[:, cut:].view(B, ws - extra, base).copy_(recv[extra:, :, :base].permute(1, 0, 2))
    return out
