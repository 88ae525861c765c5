"""Host-side towers (PyTorch-ROCm) — SURVEY §8f rank 1: the callers on either side of the hot path.

An own restatement of the two encoders with the REFERENCE'S PARAMETER NAMES, so that a reference checkpoint
(`flickr-ft.pt['model_dict']`, dvl/trainer.py:44-63) loads with ``strict=True``:

    TowerEncoder            <- dvl/models/bi_encoder.py:76-128 (BertEncoder) and :131-196 (UniterEncoder)
    BiEncoder               <- dvl/models/bi_encoder.py:199-290
    encode_proj             <- :82-88 / :137-143   Linear(H, 2H) -> erf-GELU -> LayerNorm(2H, 1e-12) -> Linear(2H, D)
    backbone                <- uniter_model/model/model.py:218-387 + layer.py:53-170 (the text tower uses the same
                               BERT stack without the image embeddings; HF BertModel has the same key names)
    CheckpointState I/O     <- dvl/trainer.py:18-20,44-90 ; eval_itm.py:97-107 (the 'bert.'-prefix fallback)

What is NOT carried over: apex FusedLayerNorm / amp (layer.py:25; bi_encoder.py:587-601) — plain ``nn.LayerNorm`` and
``torch.autocast(bfloat16)``; the attention is ``F.scaled_dot_product_attention`` with the reference's additive
(1 - mask) * -10000 bias.  ``[CLS]`` pooling goes through the HIP pooling kernel when the tower runs without autograd
on a GPU (serving.pool_cls) and is a plain slice otherwise (training needs the autograd edge).
"""
import collections
import json
import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

IMG_DIM = 2048          # region feature width (reference: dvl/const.py / uniter_model IMG_DIM)


class TowerConfig:
    """Subset of the BERT config JSON the towers need (config/img_base.json layout)."""

    def __init__(self, vocab_size=28996, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, hidden_act='gelu', hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                 initializer_range=0.02, **_ignored):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.hidden_act = hidden_act
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.initializer_range = initializer_range
        if hidden_act != 'gelu':
            raise ValueError('only the erf-GELU activation of the reference configs is supported')

    @classmethod
    def from_json(cls, path):
        with open(path) as f:
            d = json.load(f)
        if 'vocab_size_or_config_json_file' in d:
            d['vocab_size'] = d.pop('vocab_size_or_config_json_file')
        return cls(**d)


class _ErfGELU(nn.Module):
    def forward(self, x):          # uniter_model/model/layer.py:31-37
        return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


class _SelfAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.h = cfg.num_attention_heads
        self.query = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.key = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.value = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.p = cfg.attention_probs_dropout_prob

    def forward(self, x, bias):
        B, L, H = x.shape
        q = self.query(x).view(B, L, self.h, H // self.h).transpose(1, 2)
        k = self.key(x).view(B, L, self.h, H // self.h).transpose(1, 2)
        v = self.value(x).view(B, L, self.h, H // self.h).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias, dropout_p=self.p if self.training else 0.0)
        return o.transpose(1, 2).reshape(B, L, H)


class _AddNorm(nn.Module):
    """dense -> dropout -> LayerNorm(x + residual)   (BertSelfOutput / BertOutput, layer.py:103-113,144-154)"""

    def __init__(self, d_in, d_out, p):
        super().__init__()
        self.dense = nn.Linear(d_in, d_out)
        self.LayerNorm = nn.LayerNorm(d_out, eps=1e-12)
        self.dropout = nn.Dropout(p)

    def forward(self, x, residual):
        return self.LayerNorm(self.dropout(self.dense(x)) + residual)


class _Attention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self = _SelfAttention(cfg)
        self.output = _AddNorm(cfg.hidden_size, cfg.hidden_size, cfg.hidden_dropout_prob)

    def forward(self, x, bias):
        return self.output(self.self(x, bias), x)


class _Intermediate(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.intermediate_size)
        self.act = _ErfGELU()

    def forward(self, x):
        return self.act(self.dense(x))


class _Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.attention = _Attention(cfg)
        self.intermediate = _Intermediate(cfg)
        self.output = _AddNorm(cfg.intermediate_size, cfg.hidden_size, cfg.hidden_dropout_prob)

    def forward(self, x, bias):
        a = self.attention(x, bias)
        return self.output(self.intermediate(a), a)


class _Stack(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(cfg) for _ in range(cfg.num_hidden_layers)])

    def forward(self, x, bias):
        for l in self.layer:
            x = l(x, bias)
        return x


class _TextEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = nn.Embedding(cfg.vocab_size, cfg.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, cfg.hidden_size)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(cfg.hidden_dropout_prob)

    def forward(self, input_ids, position_ids, token_type_ids=None):
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        e = (self.word_embeddings(input_ids) + self.position_embeddings(position_ids)
             + self.token_type_embeddings(token_type_ids))
        return self.dropout(self.LayerNorm(e))


class _ImageEmbeddings(nn.Module):
    def __init__(self, cfg, img_dim):
        super().__init__()
        self.img_linear = nn.Linear(img_dim, cfg.hidden_size)
        self.img_layer_norm = nn.LayerNorm(cfg.hidden_size, eps=1e-12)
        self.pos_layer_norm = nn.LayerNorm(cfg.hidden_size, eps=1e-12)
        self.pos_linear = nn.Linear(7, cfg.hidden_size)
        self.mask_embedding = nn.Embedding(2, img_dim, padding_idx=0)
        self.LayerNorm = nn.LayerNorm(cfg.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(cfg.hidden_dropout_prob)

    def forward(self, img_feat, img_pos_feat, type_embeddings, img_masks=None):
        if img_masks is not None:
            mask = self.mask_embedding(img_masks.long())
            mask = mask * (img_masks.long() != 0).unsqueeze(-1)       # row 0 acts as zeros (model.py:263)
            img_feat = img_feat + mask
        e = (self.img_layer_norm(self.img_linear(img_feat)) + self.pos_layer_norm(self.pos_linear(img_pos_feat))
             + type_embeddings)
        return self.dropout(self.LayerNorm(e))


class _Pooler(nn.Module):
    """Present in checkpoints (``bert.pooler.dense.*``); the retrieval path pools [CLS] itself (bi_encoder.py:120,188)."""

    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)


class Backbone(nn.Module):
    """uniter_model/model/model.py:306-387 (UniterModel).  ``with_image=False`` gives the text-only BERT stack with
    HF BertModel's parameter names."""

    def __init__(self, cfg, with_image: bool, img_dim: int = IMG_DIM):
        super().__init__()
        self.cfg = cfg
        self.embeddings = _TextEmbeddings(cfg)
        if with_image:
            self.img_embeddings = _ImageEmbeddings(cfg, img_dim)
        self.encoder = _Stack(cfg)
        self.pooler = _Pooler(cfg)

    def forward(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index=None,
                img_masks=None):
        dtype = self.embeddings.LayerNorm.weight.dtype
        bias = (1.0 - attention_mask[:, None, None, :].to(dtype)) * -10000.0
        txt = self.embeddings(input_ids, position_ids) if input_ids is not None else None
        img = None
        if img_feat is not None:
            type_ids = torch.ones_like(img_feat[:, :, 0].long())
            img = self.img_embeddings(img_feat, img_pos_feat, self.embeddings.token_type_embeddings(type_ids), img_masks)
        if txt is None:
            x = img
        elif img is None:
            x = txt
        else:
            x = torch.cat([txt, img], dim=1)
            if gather_index is not None:
                x = torch.gather(x, 1, gather_index.unsqueeze(-1).expand(-1, -1, x.shape[-1]))
        if torch.is_autocast_enabled():
            bias = bias.to(torch.get_autocast_dtype('cuda'))
        return self.encoder(x, bias)


class TowerEncoder(nn.Module):
    """One tower: backbone (key prefix ``bert.``) + ``encode_proj``.  forward mirrors the reference signature and
    returns ``(sequence_output, pooled_output, hidden_states=None)``."""

    def __init__(self, cfg: TowerConfig, project_dim: int = 0, with_image: bool = True):
        super().__init__()
        self.config = cfg
        self.bert = Backbone(cfg, with_image)
        if project_dim > 0:
            self.encode_proj = nn.Sequential(
                nn.Linear(cfg.hidden_size, cfg.hidden_size * 2), _ErfGELU(),
                nn.LayerNorm(cfg.hidden_size * 2, eps=1e-12), nn.Linear(cfg.hidden_size * 2, project_dim))
        else:
            self.encode_proj = None
        self.apply(self._init)

    def _init(self, m):
        if isinstance(m, (nn.Linear, nn.Embedding)):
            m.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(m, nn.LayerNorm):
            m.bias.data.zero_()
            m.weight.data.fill_(1.0)
        if isinstance(m, nn.Linear) and m.bias is not None:
            m.bias.data.zero_()

    def forward(self, input_ids, attention_mask, position_ids, img_feat=None, img_pos_feat=None, img_masks=None,
                gather_index=None):
        seq = self.bert(input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index, img_masks)
        if seq.is_cuda and not torch.is_grad_enabled():
            from .serving import pool_cls                     # HIP [CLS] gather (ldot_cls_pool)
            pooled = pool_cls(seq).to(seq.dtype)
        else:
            pooled = seq[:, 0, :]
        if self.encode_proj is not None:
            pooled = self.encode_proj(pooled)
        return seq, pooled, None

    def get_out_size(self):
        return self.encode_proj[3].out_features if self.encode_proj is not None else self.config.hidden_size


class BiEncoder(nn.Module):
    """dvl/models/bi_encoder.py:199-290 — ``txt_model`` + ``img_model``; forward(batch) -> (txt, img, cap) pooled."""

    def __init__(self, args, fix_img_encoder: bool = False, fix_txt_encoder: bool = False, project_dim: int = 0,
                 txt_config: Optional[TowerConfig] = None, img_config: Optional[TowerConfig] = None):
        super().__init__()
        if getattr(args, 'img_model_type', 'uniter-base') != 'uniter-base':
            raise ValueError(f'image encoder does not support other types ({args.img_model_type}) for now')
        img_config = img_config or TowerConfig.from_json(args.img_model_config)
        txt_config = txt_config or img_config
        self.img_model = TowerEncoder(img_config, project_dim, with_image=True)
        ttype = getattr(args, 'txt_model_type', 'bert-base')
        if ttype == 'bert-base':
            self.txt_model = TowerEncoder(txt_config, project_dim, with_image=False)
        elif ttype == 'uniter-base':
            self.txt_model = TowerEncoder(txt_config, project_dim, with_image=True)
        else:
            raise ValueError(f'txt encoder does not support other types ({ttype}) for now')
        self.fix_img_encoder, self.fix_txt_encoder, self.project_dim = fix_img_encoder, fix_txt_encoder, project_dim
        if fix_txt_encoder:
            for p in self.txt_model.parameters():
                p.requires_grad = False
        if fix_img_encoder:
            for p in self.img_model.parameters():
                p.requires_grad = False

    @staticmethod
    def _run(sub_model, sb, fix):
        with torch.set_grad_enabled(torch.is_grad_enabled() and not fix):
            return sub_model(sb['input_ids'], sb['attention_mask'], sb['position_ids'], sb.get('img_feat'),
                             sb.get('img_pos_feat'), sb.get('img_masks'), sb.get('gather_index'))

    def forward(self, batch, output_all_encoded_layers=False):
        batch = collections.defaultdict(lambda: None, batch)
        txt_seq = txt_pooled = img_seq = img_pooled = cap_seq = cap_pooled = None
        if batch['txts'] is not None:
            txt_seq, txt_pooled, _ = self._run(self.txt_model, batch['txts'], self.fix_txt_encoder)
        if batch['imgs'] is not None:
            # (the reference passes fix_txt_encoder here too, bi_encoder.py:267; kept)
            img_seq, img_pooled, _ = self._run(self.img_model, batch['imgs'], self.fix_txt_encoder)
        if batch['caps'] is not None and batch['caps']['input_ids'] is not None:
            cap_seq, cap_pooled, _ = self._run(self.txt_model, batch['caps'], self.fix_txt_encoder)
        if output_all_encoded_layers:
            return txt_seq, img_seq, cap_seq
        return txt_pooled, img_pooled, cap_pooled


# ---- checkpoint surface (dvl/trainer.py:18-20,44-90; eval_itm.py:97-107) ------------------------------------------
CheckpointState = collections.namedtuple(
    'CheckpointState', ['model_dict', 'optimizer_dict', 'scheduler_dict', 'offset', 'epoch', 'encoder_params'])


def save_checkpoint(bi_encoder, optimizer, scheduler, epoch: int, offset: int, path: str, encoder_params=None):
    state = CheckpointState(bi_encoder.state_dict(), optimizer.state_dict() if optimizer is not None else None,
                            scheduler.state_dict() if scheduler is not None else None, offset, epoch, encoder_params)
    torch.save(state._asdict(), path)
    return path


def load_biencoder_checkpoint(bi_encoder, path_or_state, strict: bool = True):
    """eval_itm.py:97-107: a fine-tuning checkpoint carries ``state['model_dict']`` (strict load); a pre-training
    checkpoint is a flat dict whose BiEncoder keys are prefixed with ``bert.`` — those are kept (prefix stripped), every
    other key is dropped, then loaded strictly."""
    state = torch.load(path_or_state, map_location='cpu') if isinstance(path_or_state, str) else path_or_state
    if 'model_dict' in state:
        return bi_encoder.load_state_dict(state['model_dict'], strict=strict)
    stripped = {k[5:]: v for k, v in state.items() if k.startswith('bert.')}
    return bi_encoder.load_state_dict(stripped, strict=strict)
