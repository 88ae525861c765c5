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


class GradientBucketReducer:
    """Gradient all-reduce OVERLAPPED with backward (C1; the reference relies on horovod's DistributedOptimizer hooks for the same
    effect, pretrain.py:441-451 / uniter_model/utils/distributed.py:15-42 do it after backward).

    The parameters are laid out once in flat buckets, in REVERSE registration order (roughly the order in which backward produces
    gradients: the last layers first).  A post-accumulate-grad hook copies every finished gradient into its bucket slot; the moment a
    bucket's last gradient arrives its all-reduce is launched asynchronously (RCCL runs it on its own stream while backward continues
    with the earlier layers), so that when ``finish()`` is called after ``backward()`` only the first layers' bucket is still in flight.
    ``finish()`` launches what never filled (parameters without a gradient contribute zeros, so every rank issues the same collectives
    in the same order), waits, averages and writes the results back into ``p.grad``; a parameter that had no gradient on ANY rank keeps
    ``grad = None`` (the optimizer skips it exactly as in a single-process run).

    xGMI: buckets of 64 MiB keep every link busy with direct algorithms; ``reduce_dtype=torch.bfloat16`` halves the
    bytes (0.9 GB of fp32 gradients -> 0.45 GB) at bf16 summation precision — every rank still receives the SAME reduced values, so the
    replicas stay bit-identical; the default reduces in the gradients' own dtype.

    ``arm()`` before the backward whose gradients are to be exchanged (not before the earlier micro-steps of a gradient accumulation);
    without ``arm()`` the hooks do nothing."""

    def __init__(self, params: Iterable[nn.Parameter], bucket_bytes: int = 64 << 20, reduce_dtype: Optional[torch.dtype] = None,
                 average: bool = True, group=None):
        self.group, self.average, self.reduce_dtype, self.bucket_bytes = group, average, reduce_dtype, bucket_bytes
        self.params = [p for p in params if p.requires_grad]
        self._layout(list(reversed(self.params)))
        self.armed = False
        self._relaid = False
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    def _layout(self, ordered):
        """flat buckets over `ordered` (the order in which backward is expected to produce the gradients)"""
        self.buckets = []            # dict(params, offsets, flat, flags, pending, fired, handle)
        self.where = {}              # id(param) -> (bucket index, slot)
        cur, size = [], 0
        for p in ordered:
            cur.append(p)
            size += p.numel() * (torch.empty((), dtype=self.reduce_dtype or p.dtype).element_size())
            if size >= self.bucket_bytes:
                self._close(cur)
                cur, size = [], 0
        if cur:
            self._close(cur)

    def _close(self, plist):
        dt = self.reduce_dtype or plist[0].dtype
        offs, o = [], 0
        for p in plist:
            offs.append(o)
            o += p.numel()
        # payload + one "some rank had a gradient" element per parameter, reduced in the same collective
        dev = plist[0].device
        flat = torch.zeros(o + len(plist), dtype=dt, device=dev)
        # the flags are staged in pinned host memory so that the hook's copy does not stall the host behind backward
        flags = torch.zeros(len(plist), dtype=dt, pin_memory=dev.type == 'cuda')
        b = dict(params=plist, offsets=offs, n=o, flat=flat, flags=flags, fired=[False] * len(plist), pending=len(plist), handle=None)
        for i, p in enumerate(plist):
            self.where[id(p)] = (len(self.buckets), i)
        self.buckets.append(b)

    def arm(self):
        for b in self.buckets:
            b['fired'] = [False] * len(b['params'])
            b['pending'] = len(b['params'])
            b['handle'] = None
        self.armed = True
        self._next = 0               # buckets are launched in index order on every rank (a bucket that fills early waits for its turn)

    def _launch_ready(self):
        dist = _dist()
        while self._next < len(self.buckets) and self.buckets[self._next]['pending'] == 0:
            b = self.buckets[self._next]
            # the "had a gradient" flags travel behind the payload (one small asynchronous copy per bucket, not one fill per parameter)
            b['flags'].copy_(torch.tensor([1.0 if f else 0.0 for f in b['fired']], dtype=b['flags'].dtype))
            b['flat'][b['n']:].copy_(b['flags'], non_blocking=True)
            b['handle'] = dist.all_reduce(b['flat'], group=self.group, async_op=True) if dist is not None else None
            self._next += 1

    def _take(self, b, i, p):
        o, n = b['offsets'][i], p.numel()
        b['flat'][o:o + n].copy_(p.grad.reshape(-1))
        b['fired'][i] = True
        b['pending'] -= 1

    def _hook(self, p):
        if not self.armed:
            return
        bi, i = self.where[id(p)]
        b = self.buckets[bi]
        if b['fired'][i]:
            return
        self._take(b, i, p)
        if b['pending'] == 0:
            self._launch_ready()

    def finish(self):
        """after backward(): exchange whatever is still missing, wait, and leave the averaged gradients in p.grad"""
        if not self.armed:
            return
        self.armed = False
        dist = _dist()
        for b in self.buckets:
            if b['handle'] is not None:
                continue             # launched: every slot was filled by a hook
            for i, p in enumerate(b['params']):
                if b['fired'][i]:
                    continue
                if p.grad is not None:
                    # no hook in the armed backward, but a gradient accumulated by the earlier micro-steps (gradient accumulation
                    # with a parameter the last micro-step did not use): it takes part like every other gradient, exactly as
                    # allreduce_gradients decides by `p.grad is None`
                    self._take(b, i, p)
                else:                # zeros (and a zero "had" flag) from this rank
                    o, n = b['offsets'][i], p.numel()
                    b['flat'][o:o + n].zero_()
            b['pending'] = 0
        self._launch_ready()
        ws = dist.get_world_size(self.group) if dist is not None else 1
        for b in self.buckets:
            if b['handle'] is not None:
                b['handle'].wait()
        # ONE device -> host read for the flags of all buckets
        had_all = torch.cat([b['flat'][b['n']:].float() for b in self.buckets]).tolist()
        k, never = 0, set()
        for b in self.buckets:
            for i, p in enumerate(b['params']):
                if had_all[k + i] > 0:
                    o, n = b['offsets'][i], p.numel()
                    g = b['flat'][o:o + n].view_as(p)
                    if p.grad is None:
                        p.grad = torch.empty_like(p)
                    p.grad.copy_(g)
                    if self.average and ws > 1:
                        p.grad.div_(ws)
                else:                # no rank produced a gradient: p.grad stays None
                    never.add(id(p))
            k += len(b['params'])
        if never and not self._relaid:
            # parameters without a gradient on ANY rank (an unused pooler) sit somewhere in the bucket order and keep their bucket —
            # and, because buckets launch in order, every later one — from starting during backward.  The flags are reduced values,
            # identical on every rank, so all ranks move the same parameters behind the last bucket, once.
            self._relaid = True
            order = [p for p in reversed(self.params) if id(p) not in never] + [p for p in reversed(self.params) if id(p) in never]
            self._layout(order)

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def train_step(bi_encoder, batch, args, optimizer, scheduler=None, accumulate: bool = False, autocast_bf16: bool = False,
               reducer: Optional[GradientBucketReducer] = None):
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
    if reducer is not None and not accumulate:
        reducer.arm()                            # this backward's gradients are exchanged while it runs
    (loss / gas if gas > 1 else loss).backward()
    if not accumulate:
        if reducer is not None:
            reducer.finish()
        else:
            allreduce_gradients(bi_encoder.parameters())
        mg = float(getattr(args, 'max_grad_norm', 2.0) or 0.0)
        if mg > 0:
            torch.nn.utils.clip_grad_norm_(bi_encoder.parameters(), mg)
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
        bi_encoder.zero_grad(set_to_none=True)
    return float(loss.item()), is_correct
