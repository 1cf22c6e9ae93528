"""`PoseConditional` -- the task module of the hot path, mirroring the reference's call
surface (src/model/model.py:32-266): `forward`, `sample`, `generate_templates`, `retrieval`
with the same arguments and return tuples, on top of the HIP U-Net and scoring kernels.

Differences in *schedule*, none in arithmetic:
  * `generate_templates` encodes the reference image once (the reference re-encodes it for
    every template, model.py:115 via :219) and evaluates all N pose hypotheses as batched
    U-Net launches writing straight into the (B,N,C,h,w) bank;
  * `retrieval` never materialises the N-fold repeated query (model.py:258);
  * top-k uses "descending score, ties -> lowest index" (torch.topk leaves tie order
    unspecified);
  * with `template_parallel=True` under torch.distributed each rank generates and scores only
    its slice of the template axis and the scores are all-gathered (nope_amd/dist.py).
Lightning-specific members (optimizers, wandb logging, VSD/pyrender evaluation) are out of
scope and are not provided.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import nn

from . import dist as ndist
from . import hip


def _cfg_get(cfg, key, default=None):
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class PoseConditional(nn.Module):
    def __init__(self, u_net, optim_config=None, testing_config=None, save_dir=None, bank_dtype="f32",
                 max_hypotheses_per_launch=512, template_parallel=False, two_stream_below=None, pipeline_encoders=False, **kwargs):
        super().__init__()
        self.u_net = u_net
        self.save_dir = save_dir
        self.lr = _cfg_get(optim_config, "lr", 5e-5)
        self.weight_decay = _cfg_get(optim_config, "weight_decay", 0.0005)
        self.warm_up_steps = _cfg_get(optim_config, "warm_up_steps", 500)
        self.use_inv_deltaR = _cfg_get(optim_config, "use_inv_deltaR", True)
        self.loss_type = _cfg_get(optim_config, "loss_type", "l1")
        self.testing_config = testing_config
        self.similarity_metric = _cfg_get(testing_config, "similarity_metric", "l2")
        self.bank_dtype = bank_dtype
        self.max_hyp = int(max_hypotheses_per_launch)
        self.template_parallel = bool(template_parallel)
        # generate_and_retrieve: both encoder passes on the side stream WITHOUT waiting for what the current stream has queued, so that the
        # (launch-latency-bound, few CUs wide) encoder passes of query k + 1 run in the gaps of query k's U-Net instead of in front of their
        # own.  Only for callers whose `query` / `reference` tensors are complete when the call is made (not still being produced by work
        # queued on the current stream): results are the same bits, consecutive calls overlap on the device.  NOPE_PIPELINE_ENCODERS overrides.
        env = os.environ.get("NOPE_PIPELINE_ENCODERS")
        self.pipeline_encoders = bool(int(env)) if env is not None else bool(pipeline_encoders)
        # Single-query banks of at most this many templates (the reference's 26 / 91-template grids) are generated as two half
        # batches on two HIP streams: with a few dozen hypotheses most launches of a forward have fewer tiles than the chip has
        # workgroup slots, and two independent launch sequences fill each other's idle CUs (NOPE_TWO_STREAM_BELOW overrides; 0 = off).
        env = os.environ.get("NOPE_TWO_STREAM_BELOW")
        self.two_stream_below = int(env) if env is not None else (int(two_stream_below) if two_stream_below is not None else 0)
        self.global_step = 0
        self.global_rank = ndist.world()[0]
        if save_dir is not None:    # model.py:63-66
            os.makedirs(os.path.join(save_dir, "media"), exist_ok=True)
            os.makedirs(os.path.join(save_dir, "predictions"), exist_ok=True)
            self.log_dir = os.path.join(save_dir, "predictions")

    # ---- model.py:96-111 ------------------------------------------------------------------
    def compute_loss(self, pred, gt):
        d = (pred - gt).abs() if self.loss_type == "l1" else (pred - gt) ** 2
        return d.flatten(1).mean(dim=1).mean()

    @torch.no_grad()
    def forward(self, query, reference, relativeR):
        enc = self.u_net.encoder
        query_feat = enc.encode_image(query)
        reference_feat = enc.encode_image(reference, mode="mode")
        pred_query_feat = self.u_net(reference_feat, relativeR)
        return self.compute_loss(pred_query_feat, query_feat)

    # ---- model.py:113-124 -------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, reference, relativeR):
        reference_feat = self.u_net.encoder.encode_image(reference, mode="mode")
        pred_query_feat = self.u_net(reference_feat, relativeR)
        return pred_query_feat, None      # template encoder has no decode_latent (model.py:117-123)

    # ---- model.py:193-252 -----------------------------------------------------------------------
    @torch.no_grad()
    def generate_templates(self, reference, all_relativeR, gt_templates=None, visualize=False):
        """reference (B,3,S,S); all_relativeR (B,N,6) -> (pred_feat_templates (B,N',C,S/8,S/8),
        None, None).  N' = N, or this rank's slice of N under `template_parallel`."""
        reference_feat = self.u_net.encoder.encode_image(reference, mode="mode")   # hoisted: once, not N times
        return self.generate_templates_from_feat(reference_feat, all_relativeR), None, None

    @torch.no_grad()
    def generate_templates_from_feat(self, reference_feat, all_relativeR, defer_range_check=False):
        """defer_range_check (f16x2): the caller finishes the U-Net's range check itself (generate_and_retrieve: after scoring)."""
        out = self._generate_templates_from_feat(reference_feat, all_relativeR)
        if not defer_range_check:
            self.u_net.finish_range_check()        # (repeats the forwards in place when a layer left its accurate window: hip.UNetHandle)
        return out

    def _generate_templates_from_feat(self, reference_feat, all_relativeR):
        B, N = all_relativeR.shape[:2]
        lo, hi, ws = 0, N, 1
        if self.template_parallel:
            rank, ws = ndist.world()
            lo, hi = ndist.shard_range(N, rank, ws)
        poses = all_relativeR[:, lo:hi].contiguous().float()
        n = hi - lo
        C, h, w = self.u_net.channels, reference_feat.shape[2], reference_feat.shape[3]   # latent_dim, model.py:207-210
        bank = torch.empty((B, n, C, h, w), dtype=hip.torch_dtype(hip.dtype_code(self.bank_dtype)),
                           device=reference_feat.device)
        out = bank
        if ws > 1:
            # a sharded bank is a ShardedBank: it carries its placement (lo, hi, N) in its type, so retrieval gathers exactly the banks
            # made here, never a caller-supplied tensor that merely has the same local size
            out = ndist.ShardedBank(bank, lo, hi, N)
        if n == 0:                  # more ranks than templates: this rank still takes part in the all-gather
            return out
        if B == 1 and 2 <= n <= self.two_stream_below and reference_feat.is_cuda:
            # (off by default.  Splitting n changes the GEMM row count and with it the launch plan: the halves agree with the
            #  single-batch result to rounding, not bit for bit -- tests/test_gpu_configs.py::test_two_stream_split_close_to_single_batch)
            n_first = (n + 1) // 2
            with hip.overlap_stream(reference_feat) as side:
                self.u_net.forward_hypotheses(reference_feat, poses[:, n_first:].contiguous(), out=bank[:, n_first:], out_dtype=self.bank_dtype, defer_range_check=True)
            self.u_net.forward_hypotheses(reference_feat, poses[:, :n_first].contiguous(), out=bank[:, :n_first], out_dtype=self.bank_dtype, defer_range_check=True)
            side.join(bank)
            return out
        if n <= self.max_hyp:
            bs = max(1, self.max_hyp // n)
            for b0 in range(0, B, bs):
                self.u_net.forward_hypotheses(reference_feat[b0:b0 + bs], poses[b0:b0 + bs], out=bank[b0:b0 + bs],
                                              out_dtype=self.bank_dtype, defer_range_check=True)
        else:
            for b in range(B):
                for s in range(0, n, self.max_hyp):
                    e = min(n, s + self.max_hyp)
                    self.u_net.forward_hypotheses(reference_feat[b:b + 1], poses[b:b + 1, s:e],
                                                  out=bank[b:b + 1, s:e], out_dtype=self.bank_dtype, defer_range_check=True)
        return out

    # ---- model.py:254-266 ----------------------------------------------------------------------------
    @torch.no_grad()
    def retrieval(self, query, template_feat, shard=None):
        if self.similarity_metric != "l2":
            return None                   # the reference implements only "l2" (model.py:256,266)
        query_feat = self.u_net.encoder.encode_image(query, mode="mode")
        return self.retrieval_from_feat(query_feat, template_feat, shard=shard)

    # ---- model.py:313,323 (the two calls eval_geodesic makes back to back) --------------------------------
    @torch.no_grad()
    def generate_and_retrieve(self, query, reference, all_relativeR):
        """`generate_templates(reference, all_relativeR)` followed by `retrieval(query, bank)` as one call,
        returning (similarity, nearest_idx, bank) -- the same arithmetic in the same order, bit-identical results.  The query does not depend on
        the bank, so its encoder pass (launch-latency bound, a few CUs wide) is issued on a second HIP
        stream and runs underneath the reference encoder and the first U-Net kernels."""
        if self.similarity_metric != "l2":
            return None
        if self.pipeline_encoders and query.is_cuda:
            with hip.overlap_stream(query, wait_current=False) as side:
                reference_feat = self.u_net.encoder.encode_image(reference, mode="mode")
                ref_done = side.mark()
                query_feat = self.u_net.encoder.encode_image(query, mode="mode")
            side.join_at(ref_done, reference_feat)          # the U-Net waits for the reference's pass only
        else:
            with hip.overlap_stream(query) as side:
                query_feat = self.u_net.encoder.encode_image(query, mode="mode")
            reference_feat = self.u_net.encoder.encode_image(reference, mode="mode")
        # (a sharded step finishes the check BEFORE its collective: every rank must enter the score all-gather exactly once, and whether a
        #  forward is repeated is a per-rank fact)
        # (the handle's "repeat" behaviour only -- in its "poison" behaviour no host look is needed at all: an out-of-range bank is NaN, and so are its scores)
        defer = not (self.template_parallel and ndist.world()[1] > 1)
        bank = self.generate_templates_from_feat(reference_feat, all_relativeR, defer_range_check=defer)
        side.join(query_feat)
        similarity, nearest_idx = self.retrieval_from_feat(query_feat, bank)
        # f16x2: the U-Net's activation-range check, at the END of the step -- one stream synchronisation where the results are read anyway.
        # When a layer had left its accurate window the forwards were just repeated into the same bank: score and rank again.
        if defer and self.u_net.finish_range_check():
            similarity, nearest_idx = self.retrieval_from_feat(query_feat, bank)
        return similarity, nearest_idx, bank

    @torch.no_grad()
    def retrieval_topk_from_feat(self, query_feat, template_feat, k=5, shard=None):
        """The top-k only -- (values (B,k), nearest_idx (B,k)) -- for callers that do not keep the full similarity: under `template_parallel` every
        rank ranks its own slice and only the (B, k) (score, global index) pairs are all-gathered and merged (north_star's "all-gather of
        per-shard top-k"; tie rule of model.py:265 as everywhere: lowest global index).  Same indices as `retrieval_from_feat`."""
        sl = (template_feat.shard if isinstance(template_feat, ndist.ShardedBank) else None) if shard is None else (shard or None)
        if not (self.template_parallel and sl is not None):
            if self.template_parallel and shard is None and ndist.world()[1] > 1:
                raise hip.NopeError("template_parallel: this bank is a plain tensor without a shard placement: pass shard=(lo, hi, n_total) or shard=False")
            sim = hip.similarity(query_feat, template_feat)
            return hip.topk(sim, k)
        B, n_local = query_feat.shape[0], template_feat.shape[1]
        kl = min(k, n_local)
        if kl > 0:
            vals, idx = hip.topk(hip.similarity(query_feat, template_feat), kl)
            idx = idx + sl[0]
        else:
            vals = torch.empty((B, 0), dtype=torch.float32, device=query_feat.device)
            idx = torch.empty((B, 0), dtype=torch.int64, device=query_feat.device)
        return ndist.all_gather_topk_pairs(vals, idx, k)

    @torch.no_grad()
    def retrieval_from_feat(self, query_feat, template_feat, k=5, shard=None):
        """`shard` = (lo, hi, N): this rank's slice [lo, hi) of the N templates; `shard=False`: the bank is COMPLETE on this rank
        (e.g. loaded from disk on every rank) and is scored locally without a collective.  Banks made by `generate_templates`
        are `nope_amd.dist.ShardedBank`s and carry their placement themselves; a bank that went through another op since (`.to()`,
        a slice, `torch.cat`, save / load) is a plain tensor again: under `template_parallel` with more than one rank that raises --
        scoring only a local slice while the other ranks wait in the collective would return rank-local indices without an error."""
        sl = (template_feat.shard if isinstance(template_feat, ndist.ShardedBank) else None) if shard is None else (shard or None)
        if self.template_parallel and sl is None and shard is None and ndist.world()[1] > 1:
            raise hip.NopeError("template_parallel: this bank is a plain tensor, not a ShardedBank with a shard placement (it was not made by generate_templates, or was "
                                "copied / sliced since): pass shard=(lo, hi, n_total), or shard=False for a bank that is complete on every rank")
        if self.template_parallel and sl is not None:
            if sl[1] - sl[0] != template_feat.shape[1]:
                raise hip.NopeError(f"shard {tuple(sl)} does not match the bank's {template_feat.shape[1]} local templates")
            B, n_local = query_feat.shape[0], template_feat.shape[1]
            send, _ = ndist.gather_buffers(B, sl[2], query_feat.device)
            if n_local > 0:          # this rank's columns go straight into the collective's send buffer
                hip.similarity(query_feat, template_feat, out=send, col_offset=0)
            if ndist.world()[1] > 1 and k <= sl[2]:
                # scoring -> ONE collective -> ONE kernel (un-pad into an owned (B, N) similarity + top-k): nope_gather_topk
                return ndist.all_gather_scores_topk(n_local, sl[2], B, query_feat.device, k)
            similarity = ndist.all_gather_scores(send[:, :n_local], sl[2])
        else:
            similarity = hip.similarity(query_feat, template_feat)
        _, nearest_idx = hip.topk(similarity, k)
        return similarity, nearest_idx
