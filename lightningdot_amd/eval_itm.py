"""Evaluation entry point — host-side mirror of eval_itm.py (EVAL_MODEL, :40-152) on the MI355X path.

    python -m lightningdot_amd.eval_itm CONFIG.json CHECKPOINT.pt [--synthetic N_IMAGES]

Same config / checkpoint surface as the reference: argparse groups + JSON (options.parse_with_config), hyper-parameters
parsed from the checkpoint's directory name with the zero fallback (:55-65, incl. caption_score_weight = 0), the
sequence-length guard (:68-71), ``inf_minibatch_size = 400``, ``vector_size = project_dim`` (:79-80), strict checkpoint load
with the pre-training prefix fallback (:97-107), per-partition evaluation and the same prints (:130-152) — including the
reference's naming swap: what is printed as "image retrieval recall" is text-query -> image retrieval.

The partitions' text / image DBs are read with lightningdot_amd.data (``db_dataloader`` below = load_dataset(is_train=False) +
build_dataloader of dvl/trainer.py:29-41,193-209 over the converted FlatDb containers); a different
``dataloader_factory(args, txt_db, img_db) -> (dataloader, img2txt)`` can be injected, and ``--synthetic`` feeds random batches in
the reference collate layout (smoke / throughput runs; the recalls are then meaningless)."""
import os
import sys
import time

import numpy as np
import torch

from .harness import eval_model_on_dataloader
from .options import build_parser, parse_with_config
from .towers import BiEncoder, load_biencoder_checkpoint


def _hparams_from_dirname(args):
    parsed = os.path.basename(os.path.dirname(args.biencoder_checkpoint or '')).split('_')
    try:
        (args.learning_rate, args.train_batch_size, args.num_hard_negatives, args.hard_negatives_sampling,
         args.caption_score_weight) = parsed[1:-1]
        args.caption_score_weight = float(args.caption_score_weight)
    except ValueError:
        (args.learning_rate, args.train_batch_size, args.num_hard_negatives, args.hard_negatives_sampling,
         args.caption_score_weight) = 0, 0, 0, 0, 0
    if len(parsed) >= 4:
        args.hard_negatives_sampling = parsed[3]


def db_dataloader(args, txt_db, img_db):
    """eval_itm.py:137-142 -> dvl/trainer.py:203-207 (``TxtTokLmdb(txt_db, -1)``, ``ItmFastDataset(..., args.inf_minibatch_size, ...)``),
    ``dataset.new_epoch()``, ``build_dataloader(dataset, itm_fast_collate, False, args)`` and the partition's img2txts.json"""
    from .data import DetectFeatDb, EvalLoader, ItmFastDataset, TxtTokDb
    img = DetectFeatDb(img_db, args.conf_th, args.max_bb, args.min_bb, args.num_bb, bool(getattr(args, 'compressed_db', False)))
    txt = TxtTokDb(txt_db, -1)
    dataset = ItmFastDataset(txt, img, args.inf_minibatch_size, getattr(args, 'img_meta', None), getattr(args, 'tokenizer', None))
    dataset.new_epoch()
    return EvalLoader(dataset, args.valid_batch_size, args.device), txt.img2txts


def EVAL_MODEL(config: str, checkpoint: str, dataloader_factory=None, synthetic_images: int = 0, cmds=None):
    cmds = list(cmds or []) + ['--config', config, '--biencoder_checkpoint', checkpoint]
    args = parse_with_config(build_parser(), cmds)
    _hparams_from_dirname(args)
    if args.conf_th == -1:
        assert args.max_bb + args.max_txt_len + 2 <= 512
    else:
        assert args.num_bb + args.max_txt_len + 2 <= 512
    if not torch.cuda.is_available():
        raise RuntimeError('eval_itm needs an MI355X: the retrieval path has no CPU fallback')
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    args.device = torch.device('cuda', local_rank)
    torch.cuda.set_device(local_rank)
    args.n_gpu = int(os.environ.get('WORLD_SIZE', '1'))
    args.inf_minibatch_size = 400
    args.vector_size = args.project_dim
    bi_encoder = BiEncoder(args, args.fix_img_encoder, args.fix_txt_encoder, project_dim=args.project_dim)
    if checkpoint and os.path.exists(checkpoint):
        load_biencoder_checkpoint(bi_encoder, checkpoint)
    elif not synthetic_images:
        raise FileNotFoundError(checkpoint)
    n_img = sum(p.numel() for p in bi_encoder.img_model.parameters())
    n_txt = sum(p.numel() for p in bi_encoder.txt_model.parameters())
    print(f'total #params in img model = {n_img}, in txt model = {n_txt}')
    bi_encoder.to(args.device).eval()
    results = {}
    for partition, txt_db, img_db in zip(['dev', 'test'], [args.val_txt_db, args.test_txt_db],
                                         [args.val_img_db, args.test_img_db]):
        if txt_db is None:
            continue
        print('*' * 100)
        print('for set', partition)
        if synthetic_images:
            from .synthetic import synthetic_itm_batches
            dataloader, img2txt = synthetic_itm_batches(synthetic_images, batch_size=args.valid_batch_size,
                                                        txt_len=min(args.max_txt_len, 30), num_bb=args.num_bb,
                                                        device=args.device, seed=args.seed)
        else:
            dataloader, img2txt = (dataloader_factory or db_dataloader)(args, txt_db, img_db)
        start_time = time.time()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bool(args.fp16)):
            loss_val, correct_ratio_val, (indexer_img, indexer_txt), (recall_img, recall_txt), _ = \
                eval_model_on_dataloader(bi_encoder, dataloader, args, img2txt=img2txt)
        print(f'time cost = {time.time() - start_time}s')
        print(f'average loss = {loss_val}, accuracy = {correct_ratio_val}')
        print('indexed ', len(indexer_img.index_id_to_db_id), 'data')
        print('image retrieval recall =', recall_img)
        print('txt retrieval recall =', recall_txt)
        results[partition] = dict(loss=loss_val, accuracy=correct_ratio_val, recall_img=recall_img, recall_txt=recall_txt,
                                  recall_mean=float(np.mean(list(recall_img.values()) + list(recall_txt.values()))))
    return results


if __name__ == '__main__':
    syn = 0
    argv = sys.argv[1:]
    if '--synthetic' in argv:
        i = argv.index('--synthetic')
        syn = int(argv[i + 1])
        del argv[i:i + 2]
    EVAL_MODEL(argv[0], argv[1] if len(argv) > 1 else '', synthetic_images=syn)
