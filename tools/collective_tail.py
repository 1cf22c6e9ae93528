"""What a template-sharded step pays AFTER scoring, next to the unsharded top-5: RCCL at world size 1 (the only world a 1-GPU box has).

    python tools/collective_tail.py [--batch 1] [--templates 512]

unsharded tail:  nope_topk on the (B, N) scores
sharded tail:    all_gather_into_tensor (backend "nccl" = RCCL) of the padded (B, N/G) slices + nope_gather_topk
                 (nope_amd.dist.all_gather_scores_topk -- exactly the calls PoseConditional.retrieval_from_feat makes)
pair exchange:   nope_topk on the local slice + all-gather of (B, 5) pairs + nope_topk_merge (retrieval_topk_from_feat)
At world size 1 the collective degenerates to a device copy inside RCCL: the figure bounds the FIXED cost of the tail (launches, the
collective's host path and kernel) -- not the xGMI exchange, which needs a node (never measured).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--templates", type=int, default=512)
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29671")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from nope_amd import dist as ndist
    from nope_amd import hip
    out = []
    for B, N in ((a.batch, a.templates), (32, 512), (32, 4096)):
        sim = torch.randn(B, N, device=dev)
        send, recv = ndist.gather_buffers(B, N, dev)
        send.copy_(sim)
        want_v, want_i = hip.topk(sim, 5)
        got_s, got_i = ndist.all_gather_scores_topk(N, N, B, dev, 5)
        assert torch.equal(got_i, want_i) and torch.equal(got_s, sim)
        pv, pi = ndist.all_gather_topk_pairs(want_v, want_i, 5)
        assert torch.equal(pi, want_i)
        t_plain = timed(lambda: hip.topk(sim, 5))
        t_tail = timed(lambda: ndist.all_gather_scores_topk(N, N, B, dev, 5))
        t_gather = timed(lambda: ndist._gather_padded(send, recv))
        t_kernel = timed(lambda: hip.gather_topk(recv, N, 5))
        t_pairs = timed(lambda: ndist.all_gather_topk_pairs(want_v, want_i, 5))
        rec = {"batch": B, "templates": N, "unsharded_topk_us": t_plain, "sharded_tail_us": t_tail, "of_which_all_gather_us": t_gather,
               "of_which_gather_topk_us": t_kernel, "pair_exchange_tail_us": t_pairs, "sharded_minus_unsharded_us": t_tail - t_plain}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    print("note: wall time per call in a back-to-back loop (host issue + device), RCCL world size 1 on one MI355X; the xGMI exchange of a real "
          "node is NOT in these numbers")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
