"""Fine-tuning step — host-side mirror of train_itm.py:191-289 on the MI355X loss path, plus the data-parallel pieces the
reference only has in pretrain.py: initial parameter broadcast (pretrain.py:346), flat-buffer gradient
all-reduce (pretrain.py:441-451; uniter_model/utils/distributed.py:15-42), global in-batch negatives through the
autograd-aware embedding all-gather (loss._calc_loss).  One process per GPU, torch.distributed "nccl" = RCCL over xGMI.

    get_optimizer          <- dvl/models/bi_encoder.py:566-576   (AdamW, no weight decay on bias / LayerNorm.weight)
    get_schedule_linear    <- dvl/models/bi_encoder.py:668-680
    train_step             <- train_itm.py:191-289  (loss composition :195-222, clip :262, step :286-289)

xGMI note: a ring all-reduce of the 0.9 GB fp32 gradient is bound by ONE 153 GB/s link; the gradients are
therefore reduced in a few large flat buckets (default 256 MiB) so RCCL can use its direct algorithms on all 7 links,
and averaged (the reference's pretrain loop sums, rescale_denom = 1; with per-rank mean losses the average is the
gradient of the global-batch mean loss)."""
from typing import Iterable, Optional

import torch
import torch.nn as nn
from torch.optim.lr_scheduler import LambdaLR

from .loss import train_step_loss


def get_optimizer(model: nn.Module, learning_rate: float = 1e-5, adam_eps: float = 1e-8, weight_decay: float = 0.0):
    no_decay = ['bias', 'LayerNorm.weight']
    groups = [
        {'params': [p for n, p in model.named_parameters() if not any(nd in n for nd in no_decay)],
         'weight_decay': weight_decay},
        {'params': [p for n, p in model.named_parameters() if any(nd in n for nd in no_decay)], 'weight_decay': 0.0}]
    return torch.optim.AdamW(groups, lr=learning_rate, eps=adam_eps)


def get_schedule_linear(optimizer, warmup_steps, training_steps, last_epoch=-1):
    def lr_lambda(current_step):
        if current_step < warmup_steps:
            return float(current_step) / float(max(1, warmup_steps))
        return max(0.0, float(training_steps - current_step) / float(max(1, training_steps - warmup_steps)))
    return LambdaLR(optimizer, lr_lambda, last_epoch)


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None


def broadcast_parameters(model: nn.Module, src: int = 0):
    """rank-0 parameters (and buffers) -> all ranks, one flat message per dtype (pretrain.py:346 / C2)."""
    dist = _dist()
    if dist is None:
        return
    tensors = [p.data for p in model.parameters()] + [b.data for b in model.buffers()]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for ts in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src)
        o = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[o:o + n].view_as(t))
            o += n


def allreduce_gradients(params: Iterable[nn.Parameter], bucket_bytes: int = 256 << 20, average: bool = True):
    """Flat-bucket gradient all-reduce (C1).  Parameters without a gradient contribute zeros so that every rank issues
    the same collectives; a parameter that had no gradient on ANY rank (e.g. the unused bert.pooler) keeps ``grad = None``
    afterwards, so the optimizer skips it exactly as in a single-process run (no weight decay on untouched parameters)."""
    dist = _dist()
    if dist is None:
        return
    ws = dist.get_world_size()
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        # one extra element per parameter: "some rank had a gradient" (summed with the payload in the same collective)
        had = torch.tensor([0.0 if p.grad is None else 1.0 for p in bucket], dtype=flat.dtype, device=flat.device)
        both = torch.cat([flat, had])
        dist.all_reduce(both)
        flat, had = both[:flat.numel()], both[flat.numel():].tolist()
        if average:
            flat.div_(ws)
        o = 0
        for p, h in zip(bucket, had):
            n = p.numel()
            if h > 0:
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(flat[o:o + n].view_as(p))
            o += n
        bucket, size = [], 0

    for p in params:
        if not p.requires_grad:
            continue
        bucket.append(p)
        size += p.numel() * p.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


def train_step(bi_encoder, batch, args, optimizer, scheduler=None, accumulate: bool = False, autocast_bf16: bool = False):
    """One optimisation step (train_itm.py:191-289 without the KD branch): forward both towers, bidirectional in-batch
    NLL with appended hard negatives, backward, gradient all-reduce, clip (max_grad_norm, default 2.0), AdamW step.
    Returns (loss value, is_correct)."""
    bi_encoder.train()
    dev_type = 'cuda' if next(bi_encoder.parameters()).is_cuda else 'cpu'
    with torch.autocast(dev_type, dtype=torch.bfloat16, enabled=autocast_bf16):
        txt_vector, img_vectors, caption_vectors = bi_encoder(batch)
    loss, is_correct, _scores, _ = train_step_loss(args, txt_vector.float(), img_vectors.float(),
                                                   caption_vectors.float() if caption_vectors is not None else None,
                                                   batch)
    gas = int(getattr(args, 'gradient_accumulation_steps', 1) or 1)
    (loss / gas if gas > 1 else loss).backward()
    if not accumulate:
        allreduce_gradients(bi_encoder.parameters())
        mg = float(getattr(args, 'max_grad_norm', 2.0) or 0.0)
        if mg > 0:
            torch.nn.utils.clip_grad_norm_(bi_encoder.parameters(), mg)
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
        bi_encoder.zero_grad(set_to_none=True)
    return float(loss.item()), is_correct
