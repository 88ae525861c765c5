"""In-batch contrastive loss — host-side mirror of the reference's loss surface, backed by the HIP kernels of
csrc/loss.hip (fp32-input MFMA, fused logsumexp / NLL / arg-max rows, hand-written backward).

    dot_product_scores      <- dvl/models/bi_encoder.py:54-68
    BiEncoderNllLoss.calc   <- dvl/models/bi_encoder.py:615-656
    _calc_loss              <- dvl/utils.py:114-169  (world-size-1 path :158-167; the cross-rank branch :121-156 is
                               dead code in the reference — here it is live when torch.distributed is initialised
                               and ``args.distributed_world_size > 1``: autograd-aware all-gather of the embeddings)
    train_step_loss         <- the composition at train_itm.py:195-222

Same signatures and return values as the reference: ``(loss, correct_predictions_count, scores)`` with ``scores`` in
fp32 and everything differentiable w.r.t. q / ctx / caption vectors.  No CPU fallback: CPU tensors raise.
"""
import ctypes
from typing import List, Optional

import torch
from torch import Tensor as T

from . import _lib as L


def _ptr(t: Optional[T]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _prep(t: T) -> T:
    if not t.is_cuda:
        raise L.LdotError(-2, 'lightningdot_amd loss kernels need CUDA(HIP) tensors: there is no CPU fallback')
    return t.detach().float().contiguous()


class _InBatchNll(torch.autograd.Function):
    """scores = (1-w) q.ctx^T + w q.cap^T ; row_loss_i = logsumexp_j scores_ij - scores_i,pos_i."""

    @staticmethod
    def forward(ctx, q, c, cap, pos, w: float):
        lib = L.load_library()
        qf, cf = _prep(q), _prep(c)
        capf = _prep(cap) if cap is not None else None
        n1, d = qf.shape
        n2 = cf.shape[0]
        if cf.shape[1] != d or (capf is not None and capf.shape != cf.shape):
            raise ValueError('shape mismatch between q / ctx / caption vectors')
        dev = qf.device
        scores = torch.empty((n1, n2), dtype=torch.float32, device=dev)
        # one allocation for the small outputs: row_loss [n1] | lse [n1] | loss_sum [1] | correct [1] (int32 view)
        small = torch.empty((2 * n1 + 2,), dtype=torch.float32, device=dev)
        row_loss, lse, loss_sum = small[:n1], small[n1:2 * n1], small[2 * n1:2 * n1 + 1]
        correct = small[2 * n1 + 1:].view(torch.int32)
        L.check(lib.ldot_inbatch_nll_fwd(_ptr(qf), _ptr(cf), _ptr(capf), float(w), _ptr(pos), n1, n2, d,
                                         _ptr(scores), _ptr(row_loss), _ptr(lse), _ptr(correct), _ptr(loss_sum),
                                         _stream()))
        ctx.save_for_backward(qf, cf, capf if capf is not None else torch.empty(0, device=dev), pos, scores, lse)
        ctx.w = float(w)
        ctx.has_cap = capf is not None
        ctx.in_dtypes = (q.dtype, c.dtype, cap.dtype if cap is not None else None)
        ctx.mark_non_differentiable(correct)
        return row_loss, scores, correct, loss_sum

    @staticmethod
    def backward(ctx, g_row, g_scores, _gc, g_sum):
        lib = L.load_library()
        qf, cf, capf, pos, scores, lse = ctx.saved_tensors
        capf = capf if ctx.has_cap else None
        n1, d = qf.shape
        n2 = cf.shape[0]
        dev = qf.device
        # loss_sum = sum(row_loss) (the kernel's deterministic sum): its gradient reaches every row
        if g_sum is not None:
            g_sum = g_sum.float().reshape(1).expand(n1)
            g_row = g_sum.contiguous() if g_row is None else g_row.float() + g_sum
        g_row = torch.zeros((n1,), dtype=torch.float32, device=dev) if g_row is None else g_row.float().contiguous()
        g_scores = None if g_scores is None else g_scores.float().contiguous()
        need_q, need_c, need_cap = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        mix = capf is not None and ctx.w != 0.0
        dq = torch.empty_like(qf) if need_q else None
        dc = torch.empty_like(cf) if need_c else None
        dcap = torch.empty_like(cf) if (need_cap and mix) else None
        ds = torch.empty((n1, n2), dtype=torch.float32, device=dev)
        L.check(lib.ldot_inbatch_nll_bwd(_ptr(qf), _ptr(cf), _ptr(capf), ctx.w, _ptr(pos), n1, n2, d, _ptr(scores),
                                         _ptr(lse), _ptr(g_row), _ptr(g_scores), _ptr(ds), _ptr(dq), _ptr(dc),
                                         _ptr(dcap), _stream()))
        if need_cap and capf is not None and dcap is None:
            dcap = torch.zeros_like(cf)          # w == 0: captions do not influence the loss (bi_encoder.py:625)
        tq, tc, tcap = ctx.in_dtypes
        return (dq.to(tq) if dq is not None else None, dc.to(tc) if dc is not None else None,
                dcap.to(tcap) if dcap is not None else None, None, None)


class _DotScores(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, c):
        lib = L.load_library()
        qf, cf = _prep(q), _prep(c)
        n1, d = qf.shape
        n2 = cf.shape[0]
        out = torch.empty((n1, n2), dtype=torch.float32, device=qf.device)
        L.check(lib.ldot_dot_product_scores(_ptr(qf), _ptr(cf), n1, n2, d, _ptr(out), _stream()))
        ctx.save_for_backward(qf, cf)
        ctx.in_dtypes = (q.dtype, c.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = L.load_library()
        qf, cf = ctx.saved_tensors
        n1, d = qf.shape
        n2 = cf.shape[0]
        dev = qf.device
        g = g.float().contiguous()
        # reuse the loss backward with g_row = 0: dS = g exactly (exp(0 - 1e30) = 0)
        zeros_row = torch.zeros((n1,), dtype=torch.float32, device=dev)
        big = torch.full((n1,), 1e30, dtype=torch.float32, device=dev)
        pos = torch.zeros((n1,), dtype=torch.int32, device=dev)
        sc = torch.zeros((n1, n2), dtype=torch.float32, device=dev)
        ds = torch.empty((n1, n2), dtype=torch.float32, device=dev)
        dq, dc = torch.empty_like(qf), torch.empty_like(cf)
        L.check(lib.ldot_inbatch_nll_bwd(_ptr(qf), _ptr(cf), _ptr(None), 0.0, _ptr(pos), n1, n2, d, _ptr(sc), _ptr(big),
                                         _ptr(zeros_row), _ptr(g), _ptr(ds), _ptr(dq), _ptr(dc), _ptr(None),
                                         _stream()))
        return dq.to(ctx.in_dtypes[0]), dc.to(ctx.in_dtypes[1])


def dot_product_scores(q_vectors: T, ctx_vectors: T, cosine=False) -> T:
    """bi_encoder.py:54-68 — q_vector: n1 x D, ctx_vectors: n2 x D, result n1 x n2 (fp32)."""
    r = _DotScores.apply(q_vectors, ctx_vectors)
    if cosine:
        n1 = torch.norm(q_vectors.float(), dim=-1)
        n2 = torch.norm(ctx_vectors.float(), dim=-1)
        return r / torch.ger(n1, n2)
    return r


_range_lists = {}      # n -> list(range(n))  (the collate's pos_ctx_indices, dvl/data/itm.py:283)
_pos_cache = {}        # (n, device) -> int32 device tensor arange(n)


def _positive_tensor(positive_idx: list, n2: int, device) -> T:
    """int32 device tensor of the positives.  The reference uploads the list on every call (torch.tensor(list).to(device),
    bi_encoder.py:651,655: a pageable host-to-device copy, i.e. a synchronisation); the in-batch positives are always range(bs)
    (dvl/data/itm.py:189,283), so that tensor is built once per (length, device).  Other lists are validated and uploaded."""
    if not isinstance(positive_idx, list):          # (tensors / arrays / ranges: the reference's torch.tensor(...) takes them too)
        positive_idx = [int(v) for v in positive_idx]
    n = len(positive_idx)
    rl = _range_lists.get(n)
    if rl is None:
        rl = _range_lists[n] = list(range(n))
    if positive_idx == rl:
        if n > n2:
            raise IndexError('Target out of bounds')
        key = (n, device)
        t = _pos_cache.get(key)
        if t is None:
            t = _pos_cache[key] = torch.arange(n, dtype=torch.int32, device=device)
        return t
    if n and (min(positive_idx) < 0 or max(positive_idx) >= n2):
        raise IndexError('Target out of bounds')      # what F.nll_loss raises in the reference
    return torch.tensor(positive_idx, dtype=torch.int32).to(device)


class BiEncoderNllLoss(object):
    """bi_encoder.py:613-665"""

    def calc(self, q_vectors: T, ctx_vectors: T, caption_vectors: T, positive_idx_per_question: list,
             hard_negatice_idx_per_question: list = None, caption_score_weight: float = 0.1,
             experiment=None, reduction='mean'):
        use_cap = caption_vectors is not None and caption_score_weight != 0
        if len(q_vectors.size()) == 1:
            q_vectors = q_vectors.view(1, -1)
        n2 = ctx_vectors.shape[0]
        if len(positive_idx_per_question) != q_vectors.shape[0]:
            raise ValueError('one positive index per question is required')
        pos = _positive_tensor(positive_idx_per_question, n2, q_vectors.device)
        row_loss, scores, correct, loss_sum = _InBatchNll.apply(
            q_vectors, ctx_vectors, caption_vectors if use_cap else None, pos,
            float(caption_score_weight) if use_cap else 0.0)
        if experiment is not None:                      # same metrics as :631-643
            d = torch.diag(scores)
            experiment.log_metric('score_diag_mean', d.mean().item())
            experiment.log_metric('score_offdiag_mean', (scores.sum() - d.sum()) / (torch.numel(scores) - len(d)))
        # (mean / sum come from the forward kernel's own deterministic fp64-tree sum of the row losses, not from another reduction kernel)
        if reduction == 'mean':
            loss = loss_sum.reshape(()) / row_loss.shape[0]
        elif reduction == 'sum':
            loss = loss_sum.reshape(())
        elif reduction == 'none':
            loss = row_loss
        else:
            raise ValueError(f'{reduction} is not a valid value for reduction')
        correct_predictions_count = correct[0].to(torch.int64)
        return loss, correct_predictions_count, scores

    @staticmethod
    def get_scores(q_vector: T, ctx_vectors: T) -> T:
        f = BiEncoderNllLoss.get_similarity_function()
        return f(q_vector, ctx_vectors)

    @staticmethod
    def get_similarity_function():
        return dot_product_scores


def _f32c(t: T) -> T:
    """detached fp32 contiguous view / copy of a CUDA tensor (one op when it already is one)"""
    if not t.is_cuda:
        raise L.LdotError(-2, 'lightningdot_amd loss kernels need CUDA(HIP) tensors: there is no CPU fallback')
    if t.dtype == torch.float32 and t.is_contiguous():
        return t.detach()
    return t.detach().float().contiguous()


class _BidirNll(torch.autograd.Function):
    """The two _calc_loss calls of a fine-tuning step (train_itm.py:195-222) as ONE forward and ONE backward call into the library:
    S_txt = img[:bs].txt^T and S_img = txt[:bs].img^T share their bs x bs block (transposed), so one GEMM tile pass leaves both with
    row and column softmax statistics (ldot_inbatch_nll_bidir_fwd); nothing is read back by the host, the upstream gradients are
    read by the backward kernel from device memory.  The step's cost is the HOST's (a handful of launches around ~60 us of kernels), so
    the Python below avoids tensor views (buffers are addressed by pointer arithmetic) and conversions that would be no-ops."""

    @staticmethod
    def forward(ctx, txt, img, pos, bs: int, want_scores: bool):
        lib = L.load_library()
        tf, mf = _f32c(txt), _f32c(img)
        n, d = tf.shape
        if mf.shape != tf.shape:
            raise ValueError('txt and img vectors of a step have the same shape (bs + bs * num_hard_negatives rows)')
        dev = tf.device
        # ONE allocation: S_txt | S_img | scores_avg | dS work of the backward (bs * (2 n - bs) floats) | lse [2][bs] | row_loss [2][bs] | out [8]
        # out = loss_txt, loss_img, loss_nce, is_correct, #correct_txt, #correct_img
        o_work, o_small = 3 * bs * n, 3 * bs * n + bs * (2 * n - bs)
        buf = torch.empty((o_small + 4 * bs + 8,), dtype=torch.float32, device=dev)
        pb, sz = buf.data_ptr(), 4 * bs * n
        ps = pb + 4 * o_small
        L.check(lib.ldot_inbatch_nll_bidir_fwd(mf.data_ptr(), tf.data_ptr(), pos.data_ptr(), bs, n, d, pb, pb + sz,
                                               pb + 2 * sz if want_scores else None, ps, ps + 8 * bs, ps + 16 * bs,
                                               torch.cuda.current_stream().cuda_stream))
        ctx.save_for_backward(tf, mf, pos, buf)
        ctx.bs = bs
        ctx.in_dtypes = (txt.dtype, img.dtype)
        ctx.set_materialize_grads(False)
        loss_txt, loss_img, loss_nce, is_correct = buf[o_small + 4 * bs:o_small + 4 * bs + 4].unbind(0)
        is_correct = is_correct.clone()      # (not a view: a caller that keeps it — a history list — must not pin the whole workspace)
        ctx.mark_non_differentiable(is_correct)
        return loss_nce, loss_txt, loss_img, is_correct, (buf[2 * bs * n:3 * bs * n].view(bs, n) if want_scores else None)

    @staticmethod
    def backward(ctx, g_nce, g_txt, g_img, _g_ic, g_scores):
        lib = L.load_library()
        tf, mf, pos, buf = ctx.saved_tensors
        bs = ctx.bs
        n, d = tf.shape
        need_t, need_i = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dt = torch.empty_like(tf) if need_t else None
        di = torch.empty_like(mf) if need_i else None
        g = [None if x is None else _f32c(x) for x in (g_nce, g_txt, g_img, g_scores)]     # (kept alive across the call)
        gp = [None if x is None else x.data_ptr() for x in g]
        pb, sz = buf.data_ptr(), 4 * bs * n
        L.check(lib.ldot_inbatch_nll_bidir_bwd(mf.data_ptr(), tf.data_ptr(), pos.data_ptr(), bs, n, d, pb, pb + sz,
                                               pb + 4 * (3 * bs * n + bs * (2 * n - bs)), gp[0], gp[1], gp[2], gp[3], pb + 3 * sz,
                                               di.data_ptr() if need_i else None, dt.data_ptr() if need_t else None,
                                               torch.cuda.current_stream().cuda_stream))
        tt, ti = ctx.in_dtypes
        if need_t and tt != torch.float32:
            dt = dt.to(tt)
        if need_i and ti != torch.float32:
            di = di.to(ti)
        return dt, di, None, None, None


class _AllGatherCat(torch.autograd.Function):
    """Autograd-aware all-gather + concat along dim 0 (RCCL all_gather forward; the backward returns this rank's
    slice of the gradient — with the loss averaged over ranks by the gradient all-reduce, that is the exact
    gradient of the global-batch loss w.r.t. the local embeddings contributed to OTHER ranks' scores being
    accounted for by those ranks' own backward passes via reduce-scatter)."""

    @staticmethod
    def forward(ctx, x):
        import torch.distributed as dist
        ws, rank = dist.get_world_size(), dist.get_rank()
        sizes = [torch.zeros(1, dtype=torch.int64, device=x.device) for _ in range(ws)]
        dist.all_gather(sizes, torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device))
        sizes = [int(s.item()) for s in sizes]
        mx = max(sizes)
        pad = x if x.shape[0] == mx else torch.cat([x, x.new_zeros(mx - x.shape[0], *x.shape[1:])], 0)
        bufs = [torch.empty_like(pad) for _ in range(ws)]
        dist.all_gather(bufs, pad.contiguous())
        ctx.sizes, ctx.rank = sizes, rank
        return torch.cat([b[:n] for b, n in zip(bufs, sizes)], 0)

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        # every rank holds the gradient of ITS loss w.r.t. ALL gathered rows; the gradient of the summed loss
        # w.r.t. the local rows is the sum over ranks of the corresponding slice -> all-reduce, then slice.
        g = g.contiguous().clone()      # autograd may share the incoming buffer (retain_graph, hooks): never reduce in place
        dist.all_reduce(g)
        start = sum(ctx.sizes[:ctx.rank])
        return g[start:start + ctx.sizes[ctx.rank]]


def _calc_loss(args, loss_function, local_q_vector, local_ctx_vectors, local_caption_vectors, local_positive_idxs,
               local_hard_negatives_idxs: list = None, experiment=None):
    """dvl/utils.py:114-169.  World size 1 (the only live path of the reference, :158-167) passes straight through.
    With ``args.distributed_world_size > 1`` and an initialised process group the embeddings of all ranks are
    all-gathered (RCCL) so every rank scores its queries against the GLOBAL batch of contexts — the intent of the
    reference's dead branch (:121-156), with positives re-based by the context offset of the owning rank."""
    ws = int(getattr(args, 'distributed_world_size', 1) or 1)
    if ws > 1:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('distributed_world_size > 1 needs an initialised torch.distributed process group')
        rank = dist.get_rank()
        global_q_vector = local_q_vector
        global_ctxs_vector = _AllGatherCat.apply(local_ctx_vectors)
        global_caption_vector = (_AllGatherCat.apply(local_caption_vectors)
                                 if local_caption_vectors is not None else None)
        counts = [torch.zeros(1, dtype=torch.int64, device=local_ctx_vectors.device) for _ in range(ws)]
        dist.all_gather(counts, torch.tensor([local_ctx_vectors.shape[0]], dtype=torch.int64,
                                             device=local_ctx_vectors.device))
        offset = sum(int(c.item()) for c in counts[:rank])
        positive_idx_per_question = [v + offset for v in local_positive_idxs]
        # the collate hands over a FLAT list (dvl/data/itm.py:284: list(range(bs, n))); the reference's dead branch iterates nested
        # per-question lists (dvl/utils.py:147,152, the DPR layout) and would raise on it — both are re-based here
        hard_negatives_per_question = (None if local_hard_negatives_idxs is None else
                                       [[v + offset for v in l] if isinstance(l, (list, tuple)) else l + offset
                                        for l in local_hard_negatives_idxs])
    else:
        global_q_vector = local_q_vector
        global_ctxs_vector = local_ctx_vectors
        global_caption_vector = local_caption_vectors
        positive_idx_per_question = local_positive_idxs
        hard_negatives_per_question = local_hard_negatives_idxs

    loss, is_correct, scores = loss_function.calc(global_q_vector, global_ctxs_vector, global_caption_vector,
                                                  positive_idx_per_question, hard_negatives_per_question,
                                                  args.caption_score_weight, experiment)
    return loss, is_correct, scores


def train_step_loss(args, txt_vector: T, img_vectors: T, caption_vectors: Optional[T], batch: dict, experiment=None,
                    loss_function=None):
    """The loss composition of one fine-tuning step — train_itm.py:195-222 (both directions, averaged).
    Returns (loss_nce, is_correct, scores, (loss_nce_txt, loss_nce_img)); ``is_correct`` is a Python float on the two-call path (the
    reference's .item() arithmetic, train_itm.py:211) and a detached 0-dim DEVICE tensor on the one-call fast path (world size 1, no
    caption mixing, no experiment): ``float(is_correct)`` gives the reference's number on both.  ``loss_function`` (an object with the reference's
    ``calc``) defaults to the HIP ``BiEncoderNllLoss``, as train_itm.py:193 constructs it."""
    bs = batch['sample_size']
    ws = int(getattr(args, 'distributed_world_size', 1) or 1)
    w = getattr(args, 'caption_score_weight', 0.0)
    if (loss_function is None and ws == 1 and experiment is None and (caption_vectors is None or w == 0) and txt_vector.is_cuda
            and txt_vector.dim() == 2 and txt_vector.shape == img_vectors.shape and 0 < bs <= txt_vector.shape[0]
            and len(batch['pos_ctx_indices']) == bs
            and (args.num_hard_negatives > 0 or bs == txt_vector.shape[0])):
        # both directions in one forward and one backward call (one score GEMM for the shared bs x bs block); same values as the
        # two-call composition below.  `is_correct` is a 0-dim DEVICE tensor (the reference's .item() calls, train_itm.py:211, would
        # stall the host in front of backward()); float(is_correct) gives the reference's number.
        pos = _positive_tensor(batch['pos_ctx_indices'], txt_vector.shape[0], txt_vector.device)
        loss_nce, loss_nce_txt, loss_nce_img, is_correct, scores = _BidirNll.apply(txt_vector, img_vectors, pos, bs, True)
        return loss_nce, is_correct, scores, (loss_nce_txt, loss_nce_img)
    loss_function = loss_function or BiEncoderNllLoss()
    if args.num_hard_negatives > 0:
        loss_nce_txt, is_correct_txt, scores_txt = _calc_loss(args, loss_function, img_vectors[:bs], txt_vector,
                                                              caption_vectors, batch['pos_ctx_indices'],
                                                              batch['neg_ctx_indices'], experiment)
        loss_nce_img, is_correct_img, scores_img = _calc_loss(args, loss_function, txt_vector[:bs], img_vectors,
                                                              caption_vectors, batch['pos_ctx_indices'],
                                                              batch['neg_ctx_indices'], experiment)
    else:
        loss_nce_txt, is_correct_txt, scores_txt = _calc_loss(args, loss_function, img_vectors, txt_vector,
                                                              caption_vectors, batch['pos_ctx_indices'],
                                                              batch['neg_ctx_indices'], experiment)
        loss_nce_img, is_correct_img, scores_img = _calc_loss(args, loss_function, txt_vector, img_vectors,
                                                              caption_vectors, batch['pos_ctx_indices'],
                                                              batch['neg_ctx_indices'], experiment)
    is_correct = (is_correct_txt.sum().item() + is_correct_img.sum().item()) / 2
    loss_nce = 0.5 * loss_nce_txt + 0.5 * loss_nce_img
    scores = scores_txt * 0.5 + scores_img * 0.5
    return loss_nce, is_correct, scores, (loss_nce_txt, loss_nce_img)
