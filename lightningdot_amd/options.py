"""Config surface of the retrieval path — host-side mirror of dvl/options.py (argparse groups + JSON-under-CLI
precedence), so reference config files (config/*.json) and command lines keep working.

    default_params / add_itm_params / add_logging_params / add_kd_params   <- dvl/options.py:15-93
    parse_with_config                                                       <- dvl/options.py:96-109

One deliberate fix: the reference decides which JSON keys a CLI flag overrides by inspecting the real
``sys.argv`` even when ``cmds`` is passed; here the inspected list is ``cmds`` when given (``sys.argv[1:]``
otherwise), so programmatic callers can override JSON values too.  With ``cmds=None`` behaviour is identical.
"""
import argparse
import json
import sys

_DEFAULT = [
    ('--txt_model_type', dict(default='bert-base', type=str)), ('--txt_model_config', dict(default='bert-base', type=str)),
    ('--txt_checkpoint', dict(default=None, type=str)), ('--img_model_type', dict(default='uniter-base', type=str)),
    ('--img_model_config', dict(default='./config/img_base.json', type=str)),
    ('--img_checkpoint', dict(default=None, type=str)), ('--biencoder_checkpoint', dict(default=None, type=str)),
    ('--seperate_caption_encoder', dict(action='store_true')),
    ('--train_batch_size', dict(default=80, type=int)), ('--valid_batch_size', dict(default=80, type=int)),
    ('--gradient_accumulation_steps', dict(default=1, type=int)), ('--learning_rate', dict(default=1e-5, type=float)),
    ('--max_grad_norm', dict(default=2.0, type=float)), ('--warmup_steps', dict(default=500, type=int)),
    ('--valid_steps', dict(default=500, type=int)), ('--num_train_steps', dict(default=5000, type=int)),
    ('--num_train_epochs', dict(default=0, type=int)),
    ('--fp16', dict(action='store_true')), ('--seed', dict(default=42, type=int)),
    ('--output_dir', dict(default='./', type=str)), ('--max_txt_len', dict(default=64, type=int)),
    ('--local_rank', dict(type=int, default=-1)), ('--config', dict(default=None, type=str)),
    ('--itm_global_file', dict(default=None, type=str)), ('--no_cuda', dict(action='store_true')),
    ('--n_workers', dict(type=int, default=2)), ('--pin_mem', dict(action='store_true')),
    ('--hnsw_index', dict(action='store_true')), ('--fp16_opt_level', dict(type=str, default='O1')),
    ('--img_meta', dict(type=str, default=None)),
]
_ITM = [
    ('--conf_th', dict(default=0.2, type=float)), ('--caption_score_weight', dict(default=0.0, type=float)),
    ('--negative_size', dict(default=10, type=int)), ('--num_hard_negatives', dict(default=0, type=int)),
    ('--sample_init_hard_negatives', dict(action='store_true')),
    ('--hard_negatives_sampling', dict(default='none', type=str,
                                       choices=['none', 'random', 'top', 'top-random', '10-20', '20-30'])),
    ('--max_bb', dict(default=100, type=int)), ('--min_bb', dict(default=10, type=int)),
    ('--num_bb', dict(default=36, type=int)), ('--train_txt_dbs', dict(default=None, type=str)),
    ('--train_img_dbs', dict(default=None, type=str)), ('--txt_db_mapping', dict(default=None, type=str)),
    ('--img_db_mapping', dict(default=None, type=str)), ('--pretrain_mapping', dict(default=None, type=str)),
    ('--val_txt_db', dict(default=None, type=str)), ('--val_img_db', dict(default=None, type=str)),
    ('--test_txt_db', dict(default=None, type=str)), ('--test_img_db', dict(default=None, type=str)),
    ('--steps_per_hard_neg', dict(default=-1, type=int)), ('--inf_minibatch_size', dict(default=400, type=int)),
    ('--project_dim', dict(default=0, type=int)), ('--cls_concat', dict(default='', type=str)),
    ('--fix_txt_encoder', dict(action='store_true')), ('--fix_img_encoder', dict(action='store_true')),
    ('--compressed_db', dict(action='store_true')),
    ('--retrieval_mode', dict(default='both', choices=['img_only', 'txt_only', 'both'], type=str)),
]
_LOGGING = [
    ('--log_result_step', dict(default=4, type=int)), ('--project_name', dict(default='itm', type=str)),
    ('--expr_name_prefix', dict(default='', type=str)), ('--save_all_epochs', dict(action='store_true')),
]
_KD = [
    ('--teacher_checkpoint', dict(default=None, type=str)), ('--T', dict(default=1.0, type=float)),
    ('--kd_loss_weight', dict(default=1.0, type=float)),
]


def _add(parser, table):
    for flag, kw in table:
        parser.add_argument(flag, help='', **kw)


def default_params(parser: argparse.ArgumentParser):
    _add(parser, _DEFAULT)


def add_itm_params(parser: argparse.ArgumentParser):
    _add(parser, _ITM)


def add_logging_params(parser: argparse.ArgumentParser):
    _add(parser, _LOGGING)


def add_kd_params(parser: argparse.ArgumentParser):
    _add(parser, _KD)


def build_parser() -> argparse.ArgumentParser:
    """The parser eval_itm.py:40-52 / train_itm.py:54-63 assemble."""
    parser = argparse.ArgumentParser()
    default_params(parser)
    add_itm_params(parser)
    add_logging_params(parser)
    add_kd_params(parser)
    return parser


def parse_with_config(parser, cmds=None):
    args = parser.parse_args() if cmds is None else parser.parse_args(cmds)
    if args.config is not None:
        with open(args.config) as f:
            config_args = json.load(f)
        argv = sys.argv[1:] if cmds is None else list(cmds)
        override_keys = {arg[2:].split('=')[0] for arg in argv if arg.startswith('--')}
        for k, v in config_args.items():
            if k not in override_keys:
                setattr(args, k, v)
    return args
