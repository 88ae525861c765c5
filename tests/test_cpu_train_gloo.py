"""world_size-2 `gloo` test (CPU) of the fine-tuning loop's plumbing (lightningdot_amd.train_itm.TRAIN, mirror of train_itm.py:176-358):
rank-0 broadcast, gradient all-reduce, per-epoch evaluation + best/last/<epoch> checkpoints in the CheckpointState layout, hard-negative
re-mining feeding new_epoch, resume from `last`.  The HIP loss / retrieval / mining are replaced by plain-PyTorch stand-ins injected
through TRAIN's hooks (test infrastructure — the defaults are the HIP implementations and need a GPU)."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torch_loss(args, txt, img, cap, batch):
    """plain PyTorch restatement of the step's loss composition (train_itm.py:195-222; bi_encoder.py:615-656)"""
    import torch.nn.functional as F
    bs = batch['sample_size']
    pos = torch.arange(bs)

    def nll(q, ctx):
        scores = q @ ctx.T
        loss = F.nll_loss(F.log_softmax(scores, dim=1), pos, reduction='mean')
        return loss, (scores.argmax(1) == pos).sum(), scores
    if args.num_hard_negatives > 0:
        lt, ct, st = nll(img[:bs], txt)
        li, ci, si = nll(txt[:bs], img)
    else:
        lt, ct, st = nll(img, txt)
        li, ci, si = nll(txt, img)
    return 0.5 * lt + 0.5 * li, (ct.item() + ci.item()) / 2, 0.5 * st + 0.5 * si, (lt, li)


def _encode(bi_encoder, loader):
    tid, iid, tq, iq = [], [], [], []
    with torch.no_grad():
        for b in loader:
            t, i, _ = bi_encoder(b)
            tid += b['txt_index']
            iid += b['img_fname']
            tq.append(t)
            iq.append(i)
    return tid, iid, torch.cat(tq), torch.cat(iq)


def _torch_eval(bi_encoder, loader, args, img2txt=None, num_tops=100):
    tid, iid, tq, iq = _encode(bi_encoder, loader)
    imgs = list(dict.fromkeys(iid))
    xi = torch.stack([iq[len(iid) - 1 - iid[::-1].index(n)] for n in imgs])
    top = (tq @ xi.T).topk(min(10, len(imgs)), dim=1).indices
    want = torch.tensor([imgs.index(n) for n in iid])
    r = {t: float((top[:, :t] == want[:, None]).any(1).float().mean()) for t in (1, 5, 10)}
    return 0.0, 0.0, (None, None), (r, dict(r)), (None, None)


def _torch_mine(loaders, args, bi_encoder, img2txt, txt2img):
    nh = args.num_hard_negatives
    hn_txt, hn_img = {}, {}
    for loader in loaders:
        tid, iid, tq, iq = _encode(bi_encoder, loader)
        imgs = list(dict.fromkeys(iid))
        xi = torch.stack([iq[iid.index(n)] for n in imgs])
        s = tq @ xi.T
        for j, t in enumerate(tid):
            order = [imgs[k] for k in s[j].argsort(descending=True).tolist() if imgs[k] != txt2img[t]]
            hn_img[t] = order[:nh]
        for k, n in enumerate(imgs):
            order = [tid[j] for j in s[:, k].argsort(descending=True).tolist() if tid[j] not in img2txt[n]]
            hn_txt[n] = order[:nh]
    return hn_txt, hn_img


class _TorchNll:
    """bi_encoder.py:615-656 in plain torch ops behind the reference's `calc` signature (stand-in for the HIP BiEncoderNllLoss)"""

    def calc(self, q, ctx, cap, positive_idx, hard_neg_idx=None, caption_score_weight=0.1, experiment=None, reduction='mean'):
        import torch.nn.functional as F
        scores = q @ ctx.T
        pos = torch.tensor(positive_idx)
        return F.nll_loss(F.log_softmax(scores, dim=1), pos, reduction=reduction), (scores.argmax(1) == pos).sum(), scores


def _global_negatives_loss(args, txt, img, cap, batch):
    """the product's own composition (loss.train_step_loss -> loss._calc_loss: with args.distributed_world_size > 1 the contexts of
    all ranks are all-gathered) around the plain-torch calc"""
    from lightningdot_amd.loss import train_step_loss
    return train_step_loss(args, txt, img, cap, batch, loss_function=_TorchNll())


def _tiny_setup(nh):
    from lightningdot_amd.data import itm_fast_collate
    from lightningdot_amd.synthetic import SyntheticItmDataset
    from lightningdot_amd.towers import BiEncoder, TowerConfig
    cfg = TowerConfig(vocab_size=29000, hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                      max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)

    def model(seed):
        torch.manual_seed(seed)
        return BiEncoder(types.SimpleNamespace(img_model_type='uniter-base', txt_model_type='bert-base'), project_dim=16,
                         txt_config=cfg, img_config=cfg).double()

    # float64 throughout: AdamW turns rounding noise on near-zero gradients (the last projection's bias) into steps of the size of
    # the learning rate, which would drown the comparison in fp32
    def f64(b):
        return {k: {kk: (vv.double() if torch.is_tensor(vv) and vv.is_floating_point() else vv) for kk, vv in v.items()}
                if isinstance(v, dict) else (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}

    train_ds = SyntheticItmDataset(12, caps_per_img=2, txt_len=8, num_bb=4, img_dim=2048, num_hard_negatives=nh, seed=1)
    val_ds = SyntheticItmDataset(6, caps_per_img=2, txt_len=8, num_bb=4, img_dim=2048, seed=2)
    loader_of = lambda ds: [f64(itm_fast_collate([ds[i] for i in range(b, min(b + 6, len(ds)))])) for b in range(0, len(ds), 6)]

    def train_loader(ds, args, epoch, device):
        from lightningdot_amd.train_itm import default_train_loader
        for b in default_train_loader(ds, args, epoch, device):
            yield f64(b)
    train_loader.steps_per_epoch = lambda ds, args: len(ds) // (args.train_batch_size * args.distributed_world_size)

    def mining_loaders():
        saved = (train_ds.neg_imgs, train_ds.neg_txts)
        train_ds.new_epoch()
        out = loader_of(train_ds)
        train_ds.neg_imgs, train_ds.neg_txts = saved
        return [out]
    return model, train_ds, val_ds, loader_of(val_ds), mining_loaders, train_loader


def _args_global(out_dir, world, bs, nh):
    return types.SimpleNamespace(output_dir=out_dir, learning_rate=2e-3, num_train_epochs=2, train_batch_size=bs,
                                 gradient_accumulation_steps=1, max_grad_norm=2.0, num_hard_negatives=nh,
                                 sample_init_hard_negatives=nh > 0, hard_negatives_sampling='none', save_all_epochs=False, seed=5,
                                 distributed_world_size=world, caption_score_weight=0.0, log_result_step=100)


def _run_global(rank, world, port, out_dir, nh):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lightningdot_amd.train_itm import TRAIN
    model, train_ds, val_ds, val_loader, mining_loaders, train_loader = _tiny_setup(nh)
    m = model(100 + rank)                          # different initial weights per rank: rank 0's are broadcast
    hist = TRAIN(_args_global(os.path.join(out_dir, 'w'), world, 4, nh), m, train_ds, val_loader, val_ds.img2txts,
                 train_img2txt=train_ds.img2txts, train_txt2img=train_ds.txt2img, mining_loaders=mining_loaders,
                 make_train_loader=train_loader, loss_fn=_global_negatives_loss, evaluate=_torch_eval, mine=_torch_mine, device=torch.device('cpu'))
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    others = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(others, flat)
    assert all(torch.equal(o, others[0]) for o in others)
    losses = torch.tensor([h['loss'] for h in hist], dtype=torch.float64)     # a rank logs the mean loss of ITS queries: average the ranks
    dist.all_reduce(losses)
    if rank == 0:
        torch.save(dict(flat=flat, loss=(losses / world).tolist()), os.path.join(out_dir, 'world.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('nh', [0, 1])
def test_train_loop_global_negatives_two_ranks_equal_one_process_on_the_global_batch(tmp_path, nh):
    """TRAIN at world 2 with the loss hook going through loss._calc_loss's distributed branch (cross-rank in-batch negatives: 2 x 4
    rows per step) ends with the weights of ONE process training on the global batch of 8 — the same items (rank r takes the r-th
    slice of 4 of every 8 drawn), the same objective (mean of the ranks' mean losses = the global mean at equal sizes; with hard
    negatives the gathered context order differs from the single-process one, which the softmax does not see), the same clipping,
    schedule and re-mining."""
    sys.path.insert(0, ROOT)
    from lightningdot_amd.train_itm import TRAIN
    world = 2
    mp.spawn(_run_global, args=(world, _free_port(), str(tmp_path), nh), nprocs=world, join=True)
    got = torch.load(os.path.join(str(tmp_path), 'world.pt'))
    model, train_ds, val_ds, val_loader, mining_loaders, train_loader = _tiny_setup(nh)
    m = model(100)
    hist = TRAIN(_args_global(os.path.join(str(tmp_path), 's'), 1, 8, nh), m, train_ds, val_loader, val_ds.img2txts,
                 train_img2txt=train_ds.img2txts, train_txt2img=train_ds.txt2img, mining_loaders=mining_loaders,
                 make_train_loader=train_loader, loss_fn=_global_negatives_loss, evaluate=_torch_eval, mine=_torch_mine, device=torch.device('cpu'))
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    assert np.allclose(got['loss'], [h['loss'] for h in hist], rtol=0, atol=1e-9), (got['loss'], [h['loss'] for h in hist])
    assert torch.allclose(got['flat'], flat, rtol=0, atol=1e-7), float((got['flat'] - flat).abs().max())


def _run(rank, world, port, out_dir, nh):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lightningdot_amd.data import itm_fast_collate
    from lightningdot_amd.synthetic import SyntheticItmDataset
    from lightningdot_amd.towers import BiEncoder, CheckpointState, TowerConfig
    from lightningdot_amd.train_itm import TRAIN
    cfg = TowerConfig(vocab_size=29000, hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                      max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)

    def model(seed):
        torch.manual_seed(seed)                # DIFFERENT initial weights per rank: the broadcast must make them equal
        return BiEncoder(types.SimpleNamespace(img_model_type='uniter-base', txt_model_type='bert-base'), project_dim=16,
                         txt_config=cfg, img_config=cfg)

    def args_for(tag, epochs):
        return types.SimpleNamespace(output_dir=os.path.join(out_dir, tag), learning_rate=2e-3, num_train_epochs=epochs,
                                     train_batch_size=4, gradient_accumulation_steps=1, max_grad_norm=2.0, num_hard_negatives=nh,
                                     sample_init_hard_negatives=nh > 0, hard_negatives_sampling='none', save_all_epochs=True, seed=5,
                                     distributed_world_size=1, caption_score_weight=0.0, log_result_step=100)

    train_ds = SyntheticItmDataset(12, caps_per_img=2, txt_len=8, num_bb=4, img_dim=2048, num_hard_negatives=nh, seed=1)
    val_ds = SyntheticItmDataset(6, caps_per_img=2, txt_len=8, num_bb=4, img_dim=2048, seed=2)
    loader_of = lambda ds: [itm_fast_collate([ds[i] for i in range(b, min(b + 6, len(ds)))]) for b in range(0, len(ds), 6)]
    val_loader = loader_of(val_ds)

    def mining_loaders():
        saved = (train_ds.neg_imgs, train_ds.neg_txts)
        train_ds.new_epoch()
        out = loader_of(train_ds)
        train_ds.neg_imgs, train_ds.neg_txts = saved
        return [out]

    kw = dict(train_img2txt=train_ds.img2txts, train_txt2img=train_ds.txt2img, mining_loaders=mining_loaders, loss_fn=_torch_loss,
              evaluate=_torch_eval, mine=_torch_mine, device=torch.device('cpu'))
    # run A: 2 epochs straight through (24 captions / (4 per rank x 2 ranks) = 3 steps per epoch)
    a = model(100 + rank)
    hist = TRAIN(args_for('a', 2), a, train_ds, val_loader, val_ds.img2txts, **kw)
    assert [h['epoch'] for h in hist] == [0, 1] and all(np.isfinite(h['loss']) for h in hist)
    assert hist[1]['hard_negatives'] == (nh > 0)
    flat = torch.cat([p.detach().reshape(-1) for p in a.parameters()])
    others = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(others, flat)
    assert all(torch.equal(o, others[0]) for o in others)                      # identical parameters on both ranks
    dist.barrier()
    if rank == 0:
        for name in ('biencoder.best.pt', 'biencoder.last.pt', 'biencoder.0.pt', 'biencoder.1.pt'):
            st = torch.load(os.path.join(out_dir, 'a', name), map_location='cpu')
            assert set(st) == set(CheckpointState._fields) and st['offset'] == 0
        assert torch.load(os.path.join(out_dir, 'a', 'biencoder.last.pt'))['epoch'] == 1
    # run B: resume from the checkpoint run A wrote after its FIRST epoch (as `last` of a fresh output dir) and train the second
    # epoch only: same schedule position, same re-mined negatives (mined from the restored weights), same shuffling -> same weights
    import shutil
    os.makedirs(os.path.join(out_dir, 'b'), exist_ok=True)
    if rank == 0:
        shutil.copy(os.path.join(out_dir, 'a', 'biencoder.0.pt'), os.path.join(out_dir, 'b', 'biencoder.last.pt'))
    dist.barrier()
    c = model(300 + rank)
    hist_c = TRAIN(args_for('b', 2), c, train_ds, val_loader, val_ds.img2txts, resume_from=os.path.join(out_dir, 'b', 'biencoder.last.pt'),
                   **kw)
    assert [h['epoch'] for h in hist_c] == [1]
    flat_c = torch.cat([p.detach().reshape(-1) for p in c.parameters()])
    assert torch.allclose(flat_c, flat, rtol=0, atol=1e-6), float((flat_c - flat).abs().max())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('nh', [0, 1])
def test_train_loop_two_ranks_gloo(tmp_path, nh):
    world, port = 2, _free_port()
    mp.spawn(_run, args=(world, port, str(tmp_path), nh), nprocs=world, join=True)


def _run_reducer(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lightningdot_amd.train import GradientBucketReducer, allreduce_gradients
    import torch.nn as nn

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Linear(8, 16)
            self.b = nn.Linear(16, 16)
            self.unused = nn.Linear(4, 4)           # never produces a gradient on any rank (like bert.pooler)
            self.sometimes = nn.Linear(16, 3)       # used on rank 0 only
            self.c = nn.Linear(16, 2)

            self.early = nn.Linear(16, 5)           # used by the first micro-steps only, never by the armed (last) one

        def forward(self, x, use_extra, use_early=False):
            h = torch.tanh(self.b(torch.tanh(self.a(x))))
            out = self.c(h).sum()
            if use_early:
                out = out + self.early(h).sum()
            return out + self.sometimes(h).sum() if use_extra else out

    def fresh():
        torch.manual_seed(0)
        return Net()

    torch.manual_seed(10 + rank)
    xs = [torch.randn(5, 8) for _ in range(3)]
    # reference: gradients accumulated over 3 micro-steps, exchanged after backward
    # `early` takes part in micro-step 0 on both ranks and in micro-step 1 on rank 1 only: its hook does not fire in the armed backward,
    # its accumulated gradient must be exchanged all the same (allreduce_gradients decides by `p.grad is None`)
    early = lambda i: i == 0 or (i == 1 and rank == 1)
    ref = fresh()
    for i, x in enumerate(xs):
        ref(x, rank == 0, early(i)).backward()
    allreduce_gradients(ref.parameters())
    assert ref.early.weight.grad is not None
    for dtype, tol in ((None, 0.0), (torch.bfloat16, 2e-2)):
        net = fresh()
        red = GradientBucketReducer(net.parameters(), bucket_bytes=200, reduce_dtype=dtype)      # several small buckets
        assert len(red.buckets) > 2
        for step in range(2):                      # the second step runs on the re-laid buckets (never-used parameters moved last)
            for i, x in enumerate(xs):
                if i == len(xs) - 1:
                    red.arm()                      # only the last micro-step's backward exchanges (the accumulated gradients)
                net(x, rank == 0, early(i)).backward()
            red.finish()
            for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
                assert (p.grad is None) == (q.grad is None), n
                if p.grad is not None:
                    if tol == 0.0:
                        assert torch.equal(p.grad, q.grad), n     # same sums, same order of ranks: bit-identical to the one-pass exchange
                    else:
                        assert torch.allclose(p.grad, q.grad, rtol=tol, atol=tol), n
            assert net.unused.weight.grad is None and net.sometimes.weight.grad is not None
            assert red._relaid and red.buckets[-1]['params'][-1] in (net.unused.weight, net.unused.bias)
            if step == 0:
                net.zero_grad(set_to_none=True)
        # every rank holds the same reduced gradients
        flat = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.grad is not None])
        others = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(others, flat)
        assert all(torch.equal(o, others[0]) for o in others)
        red.remove()
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_bucket_reducer_equals_one_pass_exchange():
    """GradientBucketReducer (all-reduce launched bucket by bucket from autograd hooks, overlapping backward) gives the gradients of the
    one-pass allreduce_gradients: same values (bit-identical in the gradients' dtype, within bf16 rounding when reduced in bf16), None
    for a parameter no rank used, zeros contributed for a parameter only some ranks used, gradient accumulation respected."""
    world, port = 2, _free_port()
    mp.spawn(_run_reducer, args=(world, port), nprocs=world, join=True)
