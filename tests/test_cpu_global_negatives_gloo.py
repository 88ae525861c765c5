"""Cross-rank in-batch negatives (config 5: "global batch 512 on 8 GPUs") on `gloo`, world size 2 and 3 (CPU).

What is under test is the host logic of ``loss._calc_loss``'s ``distributed_world_size > 1`` branch and ``loss._AllGatherCat``
(padded all-gather of unequal per-rank batches, re-basing of the positive indices by the owning rank's context offset, backward =
all-reduce of the incoming gradient + slice) — the intent of the reference's dead branch dvl/utils.py:121-156 as composed by
train_itm.py:195-222.  The loss FUNCTION is a plain-torch stand-in with the reference's `calc` signature (bi_encoder.py:615-656): the
HIP loss needs a GPU and is covered by tests/test_gpu_global_negatives.py.

Checked against a single-process computation on the concatenated batch: every rank's loss and `correct`, the gradients of every rank's
local q / ctx / caption embeddings (the rank's autograd result equals W x the gradient of the objective (1/W) sum_r L_r — the 1/W is
what train.allreduce_gradients / GradientBucketReducer apply), and the parameter gradients of a small shared encoder after
``allreduce_gradients`` and after ``GradientBucketReducer`` (whose asynchronous bucket all-reduces are issued from autograd hooks
while ``_AllGatherCat.backward``'s blocking all-reduce runs between them)."""
import os
import socket
import sys
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = 24


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class TorchNll:
    """bi_encoder.py:615-656 in plain torch ops (test stand-in with the reference's signature)"""

    def calc(self, q, ctx, cap, positive_idx, hard_neg_idx=None, caption_score_weight=0.1, experiment=None, reduction='mean'):
        scores = q @ ctx.T
        if cap is not None and caption_score_weight != 0:
            scores = (1 - caption_score_weight) * scores + caption_score_weight * (q @ cap.T)
        pos = torch.tensor(positive_idx, dtype=torch.int64)
        loss = F.nll_loss(F.log_softmax(scores, dim=1), pos, reduction=reduction)
        return loss, (scores.argmax(1) == pos).sum(), scores


def make_case(world, sizes, nh, captions, seed=3):
    """inputs of ALL ranks (every rank generates the same tensors and uses its own): raw features -> a shared encoder -> q / ctx / cap"""
    g = torch.Generator().manual_seed(seed)
    feats = []
    for r in range(world):
        n1, n2 = sizes[r], sizes[r] * (1 + nh)
        fq = torch.randn(n1, D, generator=g, dtype=torch.float64)
        fc = torch.cat([fq + 0.3 * torch.randn(n1, D, generator=g, dtype=torch.float64),
                        torch.randn(n2 - n1, D, generator=g, dtype=torch.float64)])       # positives first, hard negatives appended
        fcap = torch.randn(n2, D, generator=g, dtype=torch.float64) if captions else None
        feats.append((fq, fc, fcap))
    return feats


class Enc(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(11)
        self.q = torch.nn.Linear(D, D).double()
        self.c = torch.nn.Linear(D, D).double()
        self.unused = torch.nn.Linear(3, 3).double()      # no gradient on any rank (like bert.pooler)


def single_process(world, feats, w):
    """objective (1/W) sum_r L_r, L_r = mean NLL of rank r's queries against ALL contexts; returns per-rank losses / correct, the
    gradients of every rank's embeddings and of the shared encoder's parameters"""
    enc = Enc()
    qs = [enc.q(f[0]) for f in feats]
    cs = [enc.c(f[1]) for f in feats]
    caps = [enc.c(f[2]) for f in feats] if feats[0][2] is not None else None
    for t in qs + cs + (caps or []):
        t.retain_grad()
    call, capall = torch.cat(cs), (torch.cat(caps) if caps else None)
    losses, corrects, off = [], [], 0
    for r in range(world):
        pos = [off + i for i in range(qs[r].shape[0])]
        l, c, _ = TorchNll().calc(qs[r], call, capall, pos, caption_score_weight=w)
        losses.append(l)
        corrects.append(int(c))
        off += cs[r].shape[0]
    (sum(losses) / world).backward()
    return ([float(l) for l in losses], corrects, [t.grad for t in qs], [t.grad for t in cs],
            [t.grad for t in caps] if caps else None, {n: p.grad for n, p in enc.named_parameters()})


def _run(rank, world, port, sizes, nh, captions, exchange):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lightningdot_amd.loss import _calc_loss
    from lightningdot_amd.train import GradientBucketReducer, allreduce_gradients
    w = 0.1 if captions else 0.0
    feats = make_case(world, sizes, nh, captions)
    ref_loss, ref_correct, ref_dq, ref_dc, ref_dcap, ref_dp = single_process(world, feats, w)
    enc = Enc()
    fq, fc, fcap = feats[rank]
    q, c = enc.q(fq), enc.c(fc)
    cap = enc.c(fcap) if captions else None
    for t in (q, c) + ((cap,) if captions else ()):
        t.retain_grad()
    args = types.SimpleNamespace(distributed_world_size=world, caption_score_weight=w)
    loss, correct, scores = _calc_loss(args, TorchNll(), q, c, cap, list(range(sizes[rank])), None)
    assert scores.shape == (sizes[rank], sum(sizes) * (1 + nh))             # every rank scores against the GLOBAL contexts
    assert abs(float(loss) - ref_loss[rank]) < 1e-12 and int(correct) == ref_correct[rank]
    red = None
    if exchange == 'reducer':
        red = GradientBucketReducer(enc.parameters(), bucket_bytes=1500)    # several buckets: hooks fire around the in-backward all-reduce
        assert len(red.buckets) >= 2
        red.arm()
    loss.backward()
    # the rank's embedding gradients are W x the gradients of the averaged objective
    assert torch.allclose(q.grad / world, ref_dq[rank], rtol=1e-10, atol=1e-13)
    assert torch.allclose(c.grad / world, ref_dc[rank], rtol=1e-10, atol=1e-13)
    if captions:
        assert torch.allclose(cap.grad / world, ref_dcap[rank], rtol=1e-10, atol=1e-13)
    if red is not None:
        red.finish()
        red.remove()
    else:
        allreduce_gradients(enc.parameters())
    for n, p in enc.named_parameters():
        if ref_dp[n] is None:
            assert p.grad is None, n
        else:
            assert torch.allclose(p.grad, ref_dp[n], rtol=1e-10, atol=1e-13), n
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,sizes,nh,captions,exchange', [
    (2, (5, 3), 0, False, 'onepass'),
    (2, (4, 6), 2, True, 'reducer'),
    (3, (3, 5, 2), 0, True, 'reducer'),
    (3, (4, 1, 4), 2, False, 'onepass'),
])
def test_calc_loss_global_negatives_equals_single_process(world, sizes, nh, captions, exchange):
    mp.spawn(_run, args=(world, _free_port(), sizes, nh, captions, exchange), nprocs=world, join=True)
