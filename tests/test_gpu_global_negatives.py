"""Cross-rank in-batch negatives on the HIP loss: two ranks share cuda:0 (one-GPU test box; the collectives run on gloo, which accepts
CUDA tensors), 2 x 256 rows = the 512-row global batch of BASELINE configs[4].

    loss._calc_loss (distributed branch)  -> loss._AllGatherCat (padded all-gather; backward = all-reduce + slice)
                                           -> BiEncoderNllLoss.calc on csrc/loss.hip  (256 x 512 x 768 per rank, captions mixed in one case)
    gradient exchange                      -> train.allreduce_gradients, and train.GradientBucketReducer armed (its asynchronous bucket
                                              all-reduces are issued from autograd hooks around _AllGatherCat.backward's blocking one)

Checker: the fp64 oracle (oracle.biencoder_nll_loss / biencoder_nll_grads, restating bi_encoder.py:615-656) on the concatenated batch:
rank r's loss / correct / scores, the gradients of its local q / ctx / caption embeddings (W x the gradient of (1/W) sum_r L_r) and the
parameter gradients of a small shared encoder after the exchange.  Reference intent: dvl/utils.py:121-156, train_itm.py:195-222."""
import os
import socket
import sys
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F_IN, D = 48, 768


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(world, sizes, nh, captions, seed):
    """features of every rank + the shared encoder's weights, as float64 numpy (every rank and the checker generate the same)"""
    rng = np.random.default_rng(seed)
    feats = []
    for r in range(world):
        n1, n2 = sizes[r], sizes[r] * (1 + nh)
        fq = rng.standard_normal((n1, F_IN))
        fc = np.concatenate([fq + 0.5 * rng.standard_normal((n1, F_IN)), rng.standard_normal((n2 - n1, F_IN))])
        fcap = rng.standard_normal((n2, F_IN)) if captions else None
        feats.append((fq, fc, fcap))
    wq, wc = rng.uniform(-1, 1, (D, F_IN)) * 0.03, rng.uniform(-1, 1, (D, F_IN)) * 0.03
    bq, bc = rng.uniform(-1, 1, D) * 0.01, rng.uniform(-1, 1, D) * 0.01
    return feats, (wq, bq, wc, bc)


def _oracle(world, feats, weights, w, rank):
    """fp64: rank `rank`'s forward results, its embedding gradients as autograd leaves them (sum over ranks of the slices for ctx / cap),
    and the parameter gradients of the averaged objective"""
    from oracle import oracle_np as O
    wq, bq, wc, bc = weights
    qs = [f[0] @ wq.T + bq for f in feats]
    cs = [f[1] @ wc.T + bc for f in feats]
    caps = [f[2] @ wc.T + bc for f in feats] if feats[0][2] is not None else None
    call, capall = np.concatenate(cs), (np.concatenate(caps) if caps else None)
    offs = np.cumsum([0] + [c.shape[0] for c in cs])
    dq, dctx, dcap, fwd = [], np.zeros_like(call), (np.zeros_like(call) if caps else None), None
    for r in range(world):
        pos = [int(offs[r]) + i for i in range(qs[r].shape[0])]
        if r == rank:
            fwd = O.biencoder_nll_loss(qs[r], call, capall, pos, w, 'mean', dtype=np.float64)
        g = O.biencoder_nll_grads(qs[r], call, capall, pos, w, 'mean')
        dq.append(g[0])
        dctx += g[1]
        if caps and g[2] is not None:
            dcap += g[2]
    sl = slice(int(offs[rank]), int(offs[rank + 1]))
    # parameters: y = x W^T + b  ->  dW = dy^T x, db = sum dy; averaged over the ranks
    dwq = sum(dq[r].T @ feats[r][0] for r in range(world)) / world
    dbq = sum(dq[r].sum(0) for r in range(world)) / world
    dwc = sum(dctx[offs[r]:offs[r + 1]].T @ feats[r][1] for r in range(world))
    dbc = dctx.sum(0)
    if caps:
        dwc = dwc + sum(dcap[offs[r]:offs[r + 1]].T @ feats[r][2] for r in range(world))
        dbc = dbc + dcap.sum(0)
    return fwd, dq[rank], dctx[sl], (dcap[sl] if caps else None), (dwq, dbq, dwc / world, dbc / world)


def _worker(rank, world, port, sizes, nh, captions, exchange, seed):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lightningdot_amd.loss import BiEncoderNllLoss, _calc_loss
    from lightningdot_amd.train import GradientBucketReducer, allreduce_gradients
    w = 0.1 if captions else 0.0
    feats, weights = _case(world, sizes, nh, captions, seed)
    (ref_loss, ref_correct, ref_scores), ref_dq, ref_dc, ref_dcap, ref_dp = _oracle(world, feats, weights, w, rank)

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q, self.c = torch.nn.Linear(F_IN, D), torch.nn.Linear(F_IN, D)
            self.unused = torch.nn.Linear(4, 4)                   # no gradient on any rank
            with torch.no_grad():
                for lin, (wt, b) in ((self.q, weights[:2]), (self.c, weights[2:])):
                    lin.weight.copy_(torch.from_numpy(wt))
                    lin.bias.copy_(torch.from_numpy(b))
    enc = Enc().cuda()
    t = lambda a: torch.from_numpy(a).float().cuda()
    fq, fc, fcap = feats[rank]
    q, c = enc.q(t(fq)), enc.c(t(fc))
    cap = enc.c(t(fcap)) if captions else None
    for v in (q, c) + ((cap,) if captions else ()):
        v.retain_grad()
    args = types.SimpleNamespace(distributed_world_size=world, caption_score_weight=w)
    loss, correct, scores = _calc_loss(args, BiEncoderNllLoss(), q, c, cap, list(range(sizes[rank])), None)
    assert tuple(scores.shape) == (sizes[rank], sum(sizes) * (1 + nh)) and scores.dtype == torch.float32
    np.testing.assert_allclose(scores.detach().cpu().numpy(), ref_scores, rtol=0, atol=5e-5)
    assert abs(float(loss.detach()) - float(ref_loss)) < 5e-5 and int(correct) == ref_correct, (float(loss.detach()), float(ref_loss))
    red = None
    if exchange == 'reducer':
        red = GradientBucketReducer(enc.parameters(), bucket_bytes=100 << 10)      # q and c towers in different buckets
        assert len(red.buckets) >= 2
        red.arm()
    loss.backward()
    # (db_c is exactly zero in exact arithmetic — the rows of softmax - onehot sum to zero —, hence the absolute floor)
    close = lambda got, want, what: np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=2e-4,
                                                               atol=2e-4 * max(float(np.abs(want).max()), 1e-3), err_msg=what)
    close(q.grad, ref_dq, 'dq')
    close(c.grad, ref_dc, 'dctx (sum over ranks of this rank\'s slice)')
    if captions:
        close(cap.grad, ref_dcap, 'dcap')
    if red is not None:
        red.finish()
        red.remove()
    else:
        allreduce_gradients(enc.parameters())
    close(enc.q.weight.grad, ref_dp[0], 'dW_q')
    close(enc.q.bias.grad, ref_dp[1], 'db_q')
    close(enc.c.weight.grad, ref_dp[2], 'dW_c')
    close(enc.c.bias.grad, ref_dp[3], 'db_c')
    assert enc.unused.weight.grad is None
    flat = torch.cat([p.grad.reshape(-1) for p in enc.parameters() if p.grad is not None])
    others = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(others, flat)
    assert all(torch.equal(o, others[0]) for o in others)                      # every rank holds the same reduced gradients
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('sizes,nh,captions,exchange', [
    ((256, 256), 0, False, 'onepass'),         # config 5's global batch: 2 x 256 queries against 512 contexts
    ((256, 256), 0, False, 'reducer'),
    ((256, 256), 2, True, 'reducer'),          # 256 x 1536 per rank, caption mix
    ((200, 57), 1, True, 'onepass'),           # unequal batches: the padded all-gather, offsets that are no tile multiple
])
def test_global_negatives_two_ranks_hip_loss_vs_fp64_oracle(sizes, nh, captions, exchange):
    import torch.multiprocessing as mp
    from lightningdot_amd import _lib
    _lib.require_gpu()
    mp.spawn(_worker, args=(2, _free_port(), sizes, nh, captions, exchange, 17), nprocs=2, join=True)
