"""GPU parity tests of the in-batch contrastive loss (HIP fp32-MFMA kernels) vs golden vectors captured from the
reference (G1/G2) and vs the fp64 CPU oracle at the config-5 shape (S3)."""
import json
import os
import types

import numpy as np
import pytest

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


def _cuda(a, grad=False):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.requires_grad_() if grad else t


def test_g1_loss_forward_backward(golden_dir):
    import torch
    from lightningdot_amd.loss import BiEncoderNllLoss
    g = np.load(os.path.join(golden_dir, 'g1_loss.npz'))
    cases = json.load(open(os.path.join(golden_dir, 'g1_loss_cases.json')))
    for ci, c in enumerate(cases):
        p = f'c{ci}_'
        q, ctx = _cuda(g[p + 'q'], True), _cuda(g[p + 'ctx'], True)
        cap = _cuda(g[p + 'cap'], True) if c['has_cap'] else None
        loss, correct, scores = BiEncoderNllLoss().calc(q, ctx, cap, g[p + 'pos'].tolist(), None, float(g[p + 'w']),
                                                        None, c['reduction'])
        assert scores.dtype == torch.float32 and tuple(scores.shape) == (c['n1'], c['n2'])
        np.testing.assert_allclose(scores.detach().cpu().numpy(), g[p + 'scores'], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(loss.detach().cpu().numpy(), g[p + 'loss'], rtol=2e-5, atol=2e-5)
        assert int(correct.item()) == int(g[p + 'correct']), (ci, c)
        torch.autograd.backward([loss, scores], [_cuda(g[p + 'gl']).reshape(loss.shape), _cuda(g[p + 'gs'])])
        np.testing.assert_allclose(q.grad.cpu().numpy(), g[p + 'dq'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ctx.grad.cpu().numpy(), g[p + 'dctx'], rtol=1e-4, atol=1e-5)
        if (p + 'dcap') in g.files:
            np.testing.assert_allclose(cap.grad.cpu().numpy(), g[p + 'dcap'], rtol=1e-4, atol=1e-5)
        elif c['has_cap']:
            assert cap.grad is None or float(cap.grad.abs().max()) == 0.0


def test_g2_train_step(golden_dir):
    from lightningdot_amd.loss import train_step_loss
    g = np.load(os.path.join(golden_dir, 'g2_train_step.npz'))
    cases = json.load(open(os.path.join(golden_dir, 'g2_train_step_cases.json')))
    for ci, c in enumerate(cases):
        p = f'c{ci}_'
        bs, nh = c['bs'], c['nh']
        args = types.SimpleNamespace(caption_score_weight=c['w'], num_hard_negatives=nh)
        batch = dict(sample_size=bs, pos_ctx_indices=list(range(bs)),
                     neg_ctx_indices=[[bs + i * nh + j for j in range(nh)] for i in range(bs)])
        loss, is_correct, scores, (lt, li) = train_step_loss(args, _cuda(g[p + 'txt']), _cuda(g[p + 'img']), None, batch)
        np.testing.assert_allclose(loss.cpu().numpy(), g[p + 'loss'], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(lt.cpu().numpy(), g[p + 'loss_txt'], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(li.cpu().numpy(), g[p + 'loss_img'], rtol=2e-5, atol=2e-5)
        assert is_correct == float(g[p + 'is_correct'])
        np.testing.assert_allclose(scores.cpu().numpy(), g[p + 'scores'], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('n1,n2,nh,w', [(512, 512, 0, 0.0), (512, 1536, 2, 0.0), (96, 96, 0, 0.1), (7, 13, 0, 0.3)])
def test_s3_config5_shape_vs_fp64_oracle(n1, n2, nh, w):
    """txt, img in R^{512 x 768} (+ hard negatives), seed 99; loss / grad parity vs the fp64 oracle."""
    import torch
    from lightningdot_amd.loss import BiEncoderNllLoss, dot_product_scores
    rng = np.random.default_rng(99)
    d = 768
    q = (rng.standard_normal((n1, d)) * 0.2).astype(np.float32)
    ctx = (rng.standard_normal((n2, d)) * 0.2).astype(np.float32)
    ctx[:min(n1, n2)] += q[:min(n1, n2)]
    cap = (rng.standard_normal((n2, d)) * 0.2).astype(np.float32) if w else None
    pos = [i % n2 for i in range(n1)]
    tq, tc = _cuda(q, True), _cuda(ctx, True)
    tcap = _cuda(cap, True) if w else None
    loss, correct, scores = BiEncoderNllLoss().calc(tq, tc, tcap, pos, None, w if w else 0.1)
    l64, c64, s64 = O.biencoder_nll_loss(q, ctx, cap, pos, w if w else 0.1, 'mean', dtype=np.float64)
    np.testing.assert_allclose(scores.detach().cpu().numpy(), s64, rtol=0, atol=1e-3 * 0.05)   # << the 1e-3 budget
    assert abs(float(loss.item()) - float(l64)) < 1e-4
    assert int(correct.item()) == c64
    loss.backward()
    dq, dctx, dcap = O.biencoder_nll_grads(q, ctx, cap, pos, w if w else 0.1, 'mean')
    np.testing.assert_allclose(tq.grad.cpu().numpy(), dq, rtol=1e-4, atol=2e-7)
    np.testing.assert_allclose(tc.grad.cpu().numpy(), dctx, rtol=1e-4, atol=2e-7)
    if w:
        np.testing.assert_allclose(tcap.grad.cpu().numpy(), dcap, rtol=1e-4, atol=2e-7)
    # plain score matrix + its autograd
    tq2, tc2 = _cuda(q, True), _cuda(ctx, True)
    r = dot_product_scores(tq2, tc2)
    np.testing.assert_allclose(r.detach().cpu().numpy(), q.astype(np.float64) @ ctx.astype(np.float64).T, atol=5e-5)
    gr = rng.standard_normal((n1, n2)).astype(np.float32)
    r.backward(_cuda(gr))
    np.testing.assert_allclose(tq2.grad.cpu().numpy(), gr.astype(np.float64) @ ctx, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(tc2.grad.cpu().numpy(), gr.astype(np.float64).T @ q, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('n1,n2,d,w', [(1, 1, 1, 0.0), (31, 5, 17, 0.0), (33, 100, 100, 0.25), (70, 64, 32, 0.0), (64, 97, 96, 0.1),
                                      (129, 1000, 770, 0.0), (5, 1537, 768, 0.1)])
def test_direct_kernel_edge_shapes_vs_fp64_oracle(n1, n2, d, w):
    """The 32 x 32-tile direct GEMM + fused NLL forward (csrc/loss.hip: sgemm_direct_kernel) at shapes that are not multiples of its
    tile, of its K chunk (32) or of a float4, with and without the caption mix; ties on the row maximum resolve to the first column."""
    import torch
    from lightningdot_amd.loss import BiEncoderNllLoss
    rng = np.random.default_rng(1000 * n1 + n2 + d)
    q = (rng.standard_normal((n1, d)) * 0.3).astype(np.float32)
    ctx = (rng.standard_normal((n2, d)) * 0.3).astype(np.float32)
    if n2 > 40:                      # two identical context rows: equal scores in columns 3 and 37 (different column tiles)
        ctx[37] = ctx[3]
    cap = (rng.standard_normal((n2, d)) * 0.3).astype(np.float32) if w else None
    if cap is not None and n2 > 40:
        cap[37] = cap[3]
    pos = [int(v) for v in rng.integers(0, n2, n1)]
    for reduction in ('mean', 'none'):
        tq, tc = _cuda(q, True), _cuda(ctx, True)
        tcap = _cuda(cap, True) if w else None
        loss, correct, scores = BiEncoderNllLoss().calc(tq, tc, tcap, pos, None, w if w else 0.1, None, reduction)
        l64, c64, s64 = O.biencoder_nll_loss(q, ctx, cap, pos, w if w else 0.1, reduction, dtype=np.float64)
        np.testing.assert_allclose(scores.detach().cpu().numpy(), s64, rtol=0, atol=5e-5)
        np.testing.assert_allclose(loss.detach().cpu().numpy(), l64, rtol=1e-5, atol=1e-4)
        # the oracle's arg-max is taken on fp64 scores: count with the SAME first-maximum rule on the returned fp32 scores
        s32 = scores.detach().cpu().numpy()
        assert int(correct.item()) == int((s32.argmax(axis=1) == np.asarray(pos)).sum())
        gl = rng.standard_normal(loss.shape).astype(np.float32)
        loss.backward(_cuda(gl).reshape(loss.shape))
        dq, dctx, dcap = O.biencoder_nll_grads(q, ctx, cap, pos, w if w else 0.1, reduction, grad_loss=gl)
        np.testing.assert_allclose(tq.grad.cpu().numpy(), dq, rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(tc.grad.cpu().numpy(), dctx, rtol=2e-4, atol=2e-6)
        if w:
            np.testing.assert_allclose(tcap.grad.cpu().numpy(), dcap, rtol=2e-4, atol=2e-6)


def _bidir_oracle(txt, img, bs, pos, g_nce=1.0, g_txt=0.0, g_img=0.0, g_avg=None):
    """fp64: the two calc calls of train_itm.py:195-222 and the gradient of  g_nce * loss_nce + g_txt * loss_txt + g_img * loss_img
    + <g_avg, scores_avg>  w.r.t. txt and img"""
    l1, c1, s1 = O.biencoder_nll_loss(img[:bs], txt, None, pos, 0.0, 'mean', dtype=np.float64)
    l2, c2, s2 = O.biencoder_nll_loss(txt[:bs], img, None, pos, 0.0, 'mean', dtype=np.float64)
    ga = None if g_avg is None else 0.5 * np.asarray(g_avg, np.float64)
    dq1, dc1, _ = O.biencoder_nll_grads(img[:bs], txt, None, pos, 0.0, 'mean', grad_loss=0.5 * g_nce + g_txt, grad_scores=ga)
    dq2, dc2, _ = O.biencoder_nll_grads(txt[:bs], img, None, pos, 0.0, 'mean', grad_loss=0.5 * g_nce + g_img, grad_scores=ga)
    dtxt, dimg = dc1.copy(), dc2.copy()
    dimg[:bs] += dq1
    dtxt[:bs] += dq2
    return (l1, l2, 0.5 * l1 + 0.5 * l2, (c1 + c2) / 2, 0.5 * s1 + 0.5 * s2), dtxt, dimg


@pytest.mark.parametrize('bs,nh,d', [(512, 0, 768), (512, 2, 768), (64, 0, 768), (33, 0, 96), (70, 1, 64), (100, 3, 160), (1, 0, 32)])
def test_bidirectional_step_vs_fp64_oracle_and_the_two_call_composition(bs, nh, d):
    """loss.train_step_loss on the one-pass path (ldot_inbatch_nll_bidir_fwd / _bwd: one score GEMM for the shared bs x bs block, row and
    column statistics from the same epilogue) — SURVEY S3 shapes (512 x 512, 512 x 1536) and shapes that are no multiple of a tile:
    losses, is_correct, averaged scores and the gradients of txt / img vs the fp64 oracle of train_itm.py:195-222, with upstream
    gradients into loss_nce, into the two direction losses and into the averaged scores (the KD branch, :224-241); and against the
    two-call composition on the HIP BiEncoderNllLoss."""
    import torch
    from lightningdot_amd.loss import BiEncoderNllLoss, train_step_loss
    rng = np.random.default_rng(7 * bs + nh)
    n = bs * (1 + nh)
    txt = (rng.standard_normal((n, d)) * 0.2).astype(np.float32)
    img = (rng.standard_normal((n, d)) * 0.2).astype(np.float32)
    img[:bs] += txt[:bs]
    if bs > 40:
        txt[37] = txt[3]              # equal scores in different column tiles: first arg-max in both directions
    pos = list(range(bs))
    args = types.SimpleNamespace(caption_score_weight=0.0, num_hard_negatives=nh)
    batch = dict(sample_size=bs, pos_ctx_indices=pos, neg_ctx_indices=list(range(bs, n)))
    g_avg = rng.standard_normal((bs, n)).astype(np.float32) * 1e-3
    for case in ('nce', 'all'):
        tt, ti = _cuda(txt, True), _cuda(img, True)
        loss, is_correct, scores, (lt, li) = train_step_loss(args, tt, ti, None, batch)
        assert torch.is_tensor(is_correct) and is_correct.is_cuda           # no host round trip in front of backward()
        gn, gt, gi, ga = (1.0, 0.0, 0.0, None) if case == 'nce' else (0.7, 0.3, -0.2, g_avg)
        (l1, l2, lnce, ic, savg), dtxt, dimg = _bidir_oracle(txt.astype(np.float64), img.astype(np.float64), bs, pos, gn, gt, gi, ga)
        if case == 'nce':
            loss.backward()
        else:
            (gn * loss + gt * lt + gi * li + (scores * _cuda(g_avg)).sum()).backward()
        assert abs(float(lt) - l1) < 1e-4 and abs(float(li) - l2) < 1e-4 and abs(float(loss) - lnce) < 1e-4
        np.testing.assert_allclose(scores.detach().cpu().numpy(), savg, rtol=0, atol=5e-5)
        # the oracle's arg-max is taken on fp64 scores: count with the same first-maximum rule on fp32 scores of the HIP calc
        r1 = BiEncoderNllLoss().calc(_cuda(img[:bs]), _cuda(txt), None, pos)
        r2 = BiEncoderNllLoss().calc(_cuda(txt[:bs]), _cuda(img), None, pos)
        assert float(is_correct) == (int(r1[1]) + int(r2[1])) / 2
        scale = max(float(np.abs(dtxt).max()), float(np.abs(dimg).max()))
        np.testing.assert_allclose(tt.grad.cpu().numpy(), dtxt, rtol=2e-4, atol=2e-4 * scale)
        np.testing.assert_allclose(ti.grad.cpu().numpy(), dimg, rtol=2e-4, atol=2e-4 * scale)
        if case == 'nce':
            # the two-call composition (the reference's own structure) on the same inputs
            t2, i2 = _cuda(txt, True), _cuda(img, True)
            loss2, ic2, scores2, (lt2, li2) = train_step_loss(args, t2, i2, None, batch, loss_function=BiEncoderNllLoss())
            loss2.backward()
            assert abs(float(loss2) - float(loss)) < 2e-6 and ic2 == float(is_correct)
            assert abs(float(lt2) - float(lt)) < 2e-6 and abs(float(li2) - float(li)) < 2e-6
            np.testing.assert_allclose(scores2.detach().cpu().numpy(), scores.detach().cpu().numpy(), rtol=0, atol=1e-5)
            np.testing.assert_allclose(t2.grad.cpu().numpy(), tt.grad.cpu().numpy(), rtol=1e-4, atol=1e-4 * scale)
            np.testing.assert_allclose(i2.grad.cpu().numpy(), ti.grad.cpu().numpy(), rtol=1e-4, atol=1e-4 * scale)


def test_bidirectional_step_large_and_ragged_k_take_the_staged_kernels():
    """shapes the direct kernel does not take (K not a multiple of 32; more than 256 tiles of 128 x 128): the same entry point on the
    staged kernels"""
    import torch
    from lightningdot_amd.loss import train_step_loss
    for bs, nh, d in ((48, 1, 50), (2304, 0, 64)):
        rng = np.random.default_rng(bs)
        n = bs * (1 + nh)
        txt = (rng.standard_normal((n, d)) * 0.2).astype(np.float32)
        img = (rng.standard_normal((n, d)) * 0.2).astype(np.float32)
        img[:bs] += txt[:bs]
        pos = list(range(bs))
        args = types.SimpleNamespace(caption_score_weight=0.0, num_hard_negatives=nh)
        batch = dict(sample_size=bs, pos_ctx_indices=pos, neg_ctx_indices=list(range(bs, n)))
        tt, ti = _cuda(txt, True), _cuda(img, True)
        loss, is_correct, scores, (lt, li) = train_step_loss(args, tt, ti, None, batch)
        loss.backward()
        (l1, l2, lnce, ic, savg), dtxt, dimg = _bidir_oracle(txt.astype(np.float64), img.astype(np.float64), bs, pos)
        assert abs(float(loss) - lnce) < 1e-4 and abs(float(lt) - l1) < 1e-4 and abs(float(li) - l2) < 1e-4
        np.testing.assert_allclose(scores.detach().cpu().numpy(), savg, rtol=0, atol=5e-5)
        scale = max(float(np.abs(dtxt).max()), float(np.abs(dimg).max()))
        np.testing.assert_allclose(tt.grad.cpu().numpy(), dtxt, rtol=2e-4, atol=2e-4 * scale)
        np.testing.assert_allclose(ti.grad.cpu().numpy(), dimg, rtol=2e-4, atol=2e-4 * scale)


def test_loss_rejects_cpu_tensors():
    import torch
    from lightningdot_amd import LdotError
    from lightningdot_amd.loss import BiEncoderNllLoss
    with pytest.raises(LdotError):
        BiEncoderNllLoss().calc(torch.zeros(2, 4), torch.zeros(2, 4), None, [0, 1])


def test_g3_recall_harness_on_gpu(golden_dir):
    """End-to-end eval_model_on_dataloader (fake towers) on the HIP path vs the reference's own outputs."""
    import torch
    from lightningdot_amd.harness import eval_model_on_dataloader
    from tests.test_oracle_golden import _stream_from_kw

    class Fake:
        def eval(self):
            return self

        def __call__(self, batch):
            return batch['_q'], batch['_ctx'], batch.get('_cap')

    allg = json.load(open(os.path.join(golden_dir, 'g3_recall.json')))
    for name, g in allg.items():
        batches, img2txt = _stream_from_kw(g['kw'])
        rb = []
        for b in batches:
            e = dict(txt_index=b['txt_index'], img_fname=b['img_fname'],
                     txts={'input_ids': torch.zeros(len(b['txt_index']), 4, dtype=torch.long)},
                     _q=_cuda(b['q']), _ctx=_cuda(b['ctx']))
            if 'cap' in b:
                e['_cap'] = _cuda(b['cap'])
            rb.append(e)
        args = types.SimpleNamespace(hnsw_index=False, vector_size=g['kw']['d'], caption_score_weight=g['w'])
        loss, acc, _, (r_txt, r_img), (rank_txt, rank_img) = eval_model_on_dataloader(Fake(), rb, args, img2txt,
                                                                                      g['num_tops'])
        assert abs(loss - g['loss']) < 1e-4
        assert abs(acc - g['acc']) < 1e-12
        assert {str(k): v for k, v in r_txt.items()} == g['recall_txt']
        assert {str(k): v for k, v in r_img.items()} == g['recall_img']
        assert rank_txt == g['rank_txt'], name
        assert rank_img == g['rank_img'], name
