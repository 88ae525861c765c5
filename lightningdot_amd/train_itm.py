"""Fine-tuning loop — host-side mirror of train_itm.py:176-358 on the MI355X path (BASELINE configs[4]).

    TRAIN   <- train_itm.py:176-358   epoch loop :178-293, per-epoch evaluation :303-341, checkpoints :343-349, per-epoch
                                       hard-negative re-mining :351-358, KD branch :224-241 (teacher as a callable hook)

Same control flow and the same artefacts as the reference — ``biencoder.best.pt`` / ``biencoder.last.pt`` / ``biencoder.<epoch>.pt`` in
``args.output_dir`` in the CheckpointState layout (dvl/trainer.py:44-63), written by rank 0 — on top of: the HIP in-batch loss
(loss.train_step_loss), the HIP retrieval harness for the per-epoch Recall@k (harness.eval_model_on_dataloader), device-side
hard-negative mining (hn.sampled_hard_negatives) and, for N > 1, torch.distributed over RCCL instead of horovod: rank-0 parameter
broadcast, flat-bucket gradient all-reduce, global in-batch negatives (train.py, loss._calc_loss).

Reference quirk kept by default: ``best_eval_metric`` is initialised to 0.0 and never raised (train_itm.py:176,343), so "best" is
rewritten after every epoch with a positive metric; ``track_best=True`` keeps the best epoch instead.

``loss_fn`` / ``evaluate`` / ``mine`` are injectable (the CPU world-size-2 test of the loop's plumbing runs without the HIP library;
the defaults are the HIP implementations and fail loudly without a GPU).
"""
import logging
import os
from typing import Callable, Dict, Iterable, Optional

import numpy as np
import torch

from .train import GradientBucketReducer, allreduce_gradients, broadcast_parameters, get_optimizer, get_schedule_linear
from .data import EvalLoader          # noqa: F401  (re-exported: the evaluation-style loader of the CLI below)
from .towers import CheckpointState, save_checkpoint

logger = logging.getLogger(__name__)


def _is_main() -> bool:
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def _loss_dtype(t):
    """the loss path computes in fp32 (bf16 / fp16 tower outputs are widened; float64 towers — CPU tests with an injected loss — pass)"""
    return t if t is None or t.dtype == torch.float64 else t.float()


def _checkpoint_path(args, epoch: int, offset: int = 0, cp_name: Optional[str] = None) -> str:
    """dvl/trainer.py:44-50"""
    if cp_name is None:
        return os.path.join(args.output_dir, 'biencoder.' + str(epoch) + ('.' + str(offset) if offset > 0 else '') + '.pt')
    return os.path.join(args.output_dir, 'biencoder.' + cp_name + '.pt')


def default_train_loader(train_dataset, args, epoch: int, device):
    """shuffled (seeded per epoch: a resumed run sees the batches an uninterrupted run would), collated with itm_fast_collate, moved
    to the device; every rank draws the same permutation and takes its strided share of the batches' items"""
    from .data import batch_to_device, itm_fast_collate
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    g = torch.Generator().manual_seed(int(getattr(args, 'seed', 42)) + epoch)
    order = torch.randperm(len(train_dataset), generator=g).tolist()
    bs = int(args.train_batch_size)
    for b0 in range(0, len(order) - bs * world + 1, bs * world):
        mine = order[b0 + rank * bs:b0 + (rank + 1) * bs]
        yield batch_to_device(itm_fast_collate([train_dataset[i] for i in mine]), device)


def _default_steps_per_epoch(train_dataset, args) -> int:
    import torch.distributed as dist
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return len(train_dataset) // (int(args.train_batch_size) * world)


default_train_loader.steps_per_epoch = _default_steps_per_epoch


def TRAIN(args, bi_encoder, train_dataset, val_dataloader, val_img2txt: Dict, *, train_img2txt: Optional[Dict] = None,
          train_txt2img: Optional[Dict] = None, mining_loaders: Optional[Callable[[], Iterable]] = None,
          make_train_loader: Optional[Callable] = None, kd_teacher: Optional[Callable] = None, loss_fn: Optional[Callable] = None,
          evaluate: Optional[Callable] = None, mine: Optional[Callable] = None, resume_from: Optional[str] = None,
          track_best: bool = False, autocast_bf16: bool = False, device=None):
    """Runs ``args.num_train_epochs`` epochs; returns the per-epoch history (list of dicts).

    kd_teacher(batch) -> teacher score matrix [N_teacher, n2] (train_itm.py:224-241): adds ``kd_loss_weight * T^2 *
    KLDiv(log_softmax(scores[:N] / T), softmax(teacher / T))`` to the contrastive loss.
    mining_loaders() -> iterable of evaluation-style loaders over the training sets (dvl/hn.py:46-50), used when
    ``args.num_hard_negatives > 0`` (initially if ``args.sample_init_hard_negatives``, then after every epoch)."""
    import torch.nn.functional as F
    if loss_fn is None:
        from .loss import train_step_loss as loss_fn
    if evaluate is None:
        from .harness import eval_model_on_dataloader as evaluate
    if mine is None:
        from .hn import sampled_hard_negatives as mine
    device = device or next(bi_encoder.parameters()).device
    os.makedirs(args.output_dir, exist_ok=True)
    make_train_loader = make_train_loader or default_train_loader
    nh = int(getattr(args, 'num_hard_negatives', 0) or 0)
    gas = int(getattr(args, 'gradient_accumulation_steps', 1) or 1)
    datasets = getattr(train_dataset, 'datasets', [train_dataset])

    optimizer = get_optimizer(bi_encoder, args.learning_rate)
    broadcast_parameters(bi_encoder)                                        # C2: every rank starts from rank 0's weights
    # C1: gradients are exchanged bucket by bucket WHILE backward runs (the reference relies on horovod's optimizer hooks);
    # args.grad_reduce_dtype = 'bf16' halves the bytes on xGMI, args.overlap_grad_reduce = False falls back to one pass after backward
    reducer = None
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1 \
            and getattr(args, 'overlap_grad_reduce', True):
        rd = getattr(args, 'grad_reduce_dtype', None)
        reducer = GradientBucketReducer(bi_encoder.parameters(),
                                        reduce_dtype=torch.bfloat16 if rd in ('bf16', torch.bfloat16) else None)
    # (counted, not iterated: a dry pass over the loader would read and collate every item and image feature once more)
    if hasattr(make_train_loader, 'steps_per_epoch'):
        steps_per_epoch = int(make_train_loader.steps_per_epoch(train_dataset, args))
    else:
        steps_per_epoch = sum(1 for _ in make_train_loader(train_dataset, args, 0, torch.device('cpu')))
    updates_per_epoch = steps_per_epoch // gas
    total_updates = updates_per_epoch * int(args.num_train_epochs)
    scheduler = get_schedule_linear(optimizer, int(0.1 * total_updates), total_updates)     # :172-175
    start_epoch = 0
    if resume_from:                                                         # dvl/trainer.py:66-84 (load_saved_state)
        state = CheckpointState(**torch.load(resume_from, map_location='cpu'))
        bi_encoder.load_state_dict(state.model_dict)
        if state.optimizer_dict:
            optimizer.load_state_dict(state.optimizer_dict)
        if state.scheduler_dict:
            scheduler.load_state_dict(state.scheduler_dict)
        start_epoch = state.epoch + (1 if state.offset == 0 else 0)
        logger.info('resumed from %s at epoch %d', resume_from, start_epoch)

    hard_neg_txt = hard_neg_img = None
    if nh > 0 and (getattr(args, 'sample_init_hard_negatives', False) or start_epoch > 0):
        hard_neg_txt, hard_neg_img = mine(mining_loaders(), args, bi_encoder, train_img2txt, train_txt2img)     # :150-152
    elif nh > 0 and not getattr(args, 'sample_init_hard_negatives', False) and start_epoch == 0:
        raise NotImplementedError('random init hard negatives not impelmented yet')                              # :155-156

    best_eval_metric, history, seed_grad = 0.0, [], None
    for epoch in range(start_epoch, int(args.num_train_epochs)):
        epoch_loss, epoch_correct, n_steps = 0.0, 0.0, 0
        bi_encoder.train()
        for dset in datasets:
            dset.new_epoch(hard_neg_img, hard_neg_txt)                                                           # :188-190
        for step, batch in enumerate(make_train_loader(train_dataset, args, epoch, device)):
            dev_type = 'cuda' if device.type == 'cuda' else 'cpu'
            with torch.autocast(dev_type, dtype=torch.bfloat16, enabled=autocast_bf16):
                txt_vector, img_vectors, caption_vectors = bi_encoder(batch)
            loss_nce, is_correct, scores, _ = loss_fn(args, _loss_dtype(txt_vector), _loss_dtype(img_vectors),
                                                      _loss_dtype(caption_vectors), batch)
            loss = loss_nce
            if kd_teacher is not None:                                                                           # :224-241
                with torch.no_grad():
                    teacher_scores = kd_teacher(batch)
                n_t, T = teacher_scores.shape[0], float(args.T)
                loss_kd = torch.nn.KLDivLoss()(F.log_softmax(scores[:n_t] / T, dim=1), F.softmax(teacher_scores / T, dim=1)) * T * T
                loss = loss_nce + float(args.kd_loss_weight) * loss_kd
            if gas > 1:
                loss = loss / gas
            # (accumulated on the device: a float() / .item() here would stall the host in front of backward(), train_itm.py:211,243)
            epoch_correct = epoch_correct + (is_correct.detach() if torch.is_tensor(is_correct) else float(is_correct))
            epoch_loss = epoch_loss + loss.detach()
            n_steps += 1
            last_micro = (step + 1) % gas == 0
            if reducer is not None and last_micro:
                reducer.arm()
            if seed_grad is None or seed_grad.dtype != loss.dtype:
                seed_grad = torch.ones((), dtype=loss.dtype, device=loss.device)     # (loss.backward() alone fills a fresh one per step)
            loss.backward(seed_grad)
            if last_micro:
                if reducer is not None:
                    reducer.finish()                                        # C1: only the first layers' bucket is still in flight here
                else:
                    allreduce_gradients(bi_encoder.parameters())
                mg = float(getattr(args, 'max_grad_norm', 2.0) or 0.0)
                if mg > 0:
                    torch.nn.utils.clip_grad_norm_(bi_encoder.parameters(), mg)                                  # :262
                optimizer.step()
                scheduler.step()
                bi_encoder.zero_grad(set_to_none=True)
            if (step + 1) % int(getattr(args, 'log_result_step', 100) or 100) == 0 and _is_main():
                logger.info('Epoch: %d: Step: %d/%d, loss=%f, lr=%f', epoch, step, steps_per_epoch, loss.item(),
                            optimizer.param_groups[0]['lr'])
        epoch_loss = float(epoch_loss) / n_steps if n_steps else 0.0
        epoch_correct = float(epoch_correct)
        correct_ratio = epoch_correct / max(n_steps * int(args.train_batch_size), 1)

        # eval and save (:303-349)
        bi_encoder.eval()
        loss_val, correct_ratio_val, _indexers, recall_both, _ = evaluate(bi_encoder, val_dataloader, args, img2txt=val_img2txt)
        recall_val = {t: (recall_both[0][t] + recall_both[1][t]) / 2 for t in recall_both[0]}
        current_eval_metric = float(np.mean(list(recall_val.values())))
        if current_eval_metric > best_eval_metric and _is_main():
            save_checkpoint(bi_encoder, optimizer, scheduler, epoch, 0, _checkpoint_path(args, epoch, cp_name='best'))
        if track_best:
            best_eval_metric = max(best_eval_metric, current_eval_metric)
        if _is_main():
            save_checkpoint(bi_encoder, optimizer, scheduler, epoch, 0, _checkpoint_path(args, epoch, cp_name='last'))
            if getattr(args, 'save_all_epochs', False):
                save_checkpoint(bi_encoder, optimizer, scheduler, epoch, 0, _checkpoint_path(args, epoch))
        history.append(dict(epoch=epoch, loss=epoch_loss, correct_ratio=correct_ratio, val_loss=float(loss_val),
                            val_correct_ratio=float(correct_ratio_val), recall=recall_val, metric=current_eval_metric,
                            hard_negatives=hard_neg_img is not None))
        if _is_main():
            logger.info('epoch %d: loss %.4f, val loss %.4f, recall %s', epoch, epoch_loss, loss_val, recall_val)

        # sample hard negative in here (:351-358)
        if nh > 0:
            hard_neg_txt, hard_neg_img = mine(mining_loaders(), args, bi_encoder, train_img2txt, train_txt2img)
        else:
            hard_neg_txt, hard_neg_img = None, None
            assert getattr(args, 'hard_negatives_sampling', 'none') == 'none', \
                f'sampleing method {args.hard_negatives_sampling} is not none'
    return history


def main(argv=None):
    """python -m lightningdot_amd.train_itm --config CONFIG.json [--synthetic N_IMAGES]   (one process per GPU under
    torch.distributed.run for N > 1).  With --synthetic the databases are replaced by synthetic.SyntheticItmDataset; otherwise
    the train / val text and image DBs named in the config are read with lightningdot_amd.data (converted FlatDb containers)."""
    import sys
    import torch.distributed as dist
    from .data import DetectFeatDb, ItmFastDataset, TxtTokDb
    from .options import build_parser, parse_with_config
    from .synthetic import SyntheticItmDataset
    from .towers import BiEncoder, load_biencoder_checkpoint
    argv = list(sys.argv[1:] if argv is None else argv)
    syn = 0
    if '--synthetic' in argv:
        i = argv.index('--synthetic')
        syn = int(argv[i + 1])
        del argv[i:i + 2]
    args = parse_with_config(build_parser(), argv)
    if not torch.cuda.is_available():
        raise RuntimeError('train_itm needs an MI355X: the loss / retrieval path has no CPU fallback')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    args.device = torch.device('cuda', local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=args.device)
    args.distributed_world_size = world
    args.vector_size = args.project_dim if args.project_dim > 0 else 768
    bi_encoder = BiEncoder(args, args.fix_img_encoder, args.fix_txt_encoder, args.project_dim)
    if args.biencoder_checkpoint and os.path.exists(args.biencoder_checkpoint):
        load_biencoder_checkpoint(bi_encoder, args.biencoder_checkpoint)
    bi_encoder.to(args.device)

    def eval_loader(ds):
        return EvalLoader(ds, args.valid_batch_size, args.device)

    if syn:
        train_ds = SyntheticItmDataset(syn, num_hard_negatives=args.num_hard_negatives, seed=args.seed)
        val_ds = SyntheticItmDataset(max(syn // 4, 8), seed=args.seed + 1)
        train_img2txt, train_txt2img, val_img2txt = train_ds.img2txts, train_ds.txt2img, val_ds.img2txts
    else:
        img_db = lambda p: DetectFeatDb(p, args.conf_th, args.max_bb, args.min_bb, args.num_bb, args.compressed_db)
        first = lambda v: v[0] if isinstance(v, (list, tuple)) else v
        txt = TxtTokDb(first(args.train_txt_dbs), args.max_txt_len)
        train_ds = ItmFastDataset(txt, img_db(first(args.train_img_dbs)), args.num_hard_negatives)
        vtxt = TxtTokDb(args.val_txt_db, -1)
        val_ds = ItmFastDataset(vtxt, img_db(args.val_img_db))
        train_img2txt, train_txt2img, val_img2txt = txt.img2txts, txt.txt2img, vtxt.img2txts
    val_ds.new_epoch()

    def mining_loaders():
        # evaluation-style items (no negatives appended) over the training set (dvl/hn.py:46-50), then the epoch's bindings back
        # (the loader is lazy: the evaluation-style bindings must be in force WHILE it is consumed, the epoch's afterwards)
        def loader():
            saved = (train_ds.neg_imgs, train_ds.neg_txts)
            train_ds.new_epoch()
            try:
                yield from eval_loader(train_ds)
            finally:
                train_ds.neg_imgs, train_ds.neg_txts = saved
        return [loader()]

    hist = TRAIN(args, bi_encoder, train_ds, eval_loader(val_ds), val_img2txt, train_img2txt=train_img2txt,
                 train_txt2img=train_txt2img, mining_loaders=mining_loaders, autocast_bf16=bool(args.fp16))
    if _is_main():
        for h in hist:
            print(h)
    if world > 1:
        dist.destroy_process_group()
    return hist


if __name__ == '__main__':
    main()
