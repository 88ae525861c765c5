"""Towers pinned against the reference: a small UniterEncoder with seeded weights (golden G5, generated
from the reference by oracle/gen_golden.py) — image path, text-only path, projection — and the real config's checkpoint key
manifest (strict-load surface)."""
import json
import os
import types

import numpy as np
import pytest
import torch

from lightningdot_amd.towers import (BiEncoder, CheckpointState, TowerConfig, TowerEncoder, load_biencoder_checkpoint,
                                     save_checkpoint)


def _small(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g5_tower_small.npz'))
    cfg = TowerConfig(**{('vocab_size' if k == 'vocab_size_or_config_json_file' else k): v
                         for k, v in json.loads(str(g['cfg'])).items()})
    enc = TowerEncoder(cfg, project_dim=int(g['project_dim']), with_image=True)
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd__')}
    missing = enc.load_state_dict(sd, strict=True)          # same key names as the reference
    assert not missing.missing_keys and not missing.unexpected_keys
    return g, enc.eval()


def test_small_tower_matches_reference_outputs(golden_dir):
    g, enc = _small(golden_dir)
    t = lambda k: torch.from_numpy(g[k])
    with torch.no_grad():
        seq, pooled, _ = enc(t('img_input_ids'), t('img_attn'), t('img_position_ids'), t('img_feat'), t('img_pos_feat'),
                             None, t('gather_index'))
        tseq, tpooled, _ = enc(t('txt_input_ids'), t('txt_attn'), t('txt_position_ids'))
    # padded positions attend like the reference (additive -10000 bias), so every position is comparable
    np.testing.assert_allclose(seq.numpy(), g['img_seq'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(pooled.numpy(), g['img_pooled'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(tseq.numpy(), g['txt_seq'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(tpooled.numpy(), g['txt_pooled'], rtol=1e-4, atol=2e-5)


def test_text_tower_matches_the_reference_bert_encoder(golden_dir):
    """Golden G9: the reference's BertEncoder (transformers.BertModel + [CLS] pooling + encode_proj, bi_encoder.py:76-128) with seeded
    weights -> its state_dict loads STRICTLY into TowerEncoder(with_image=False) and the outputs agree (padded batch)."""
    g = np.load(os.path.join(golden_dir, 'g9_text_tower_small.npz'))
    cfg = TowerConfig(**json.loads(str(g['cfg'])))
    enc = TowerEncoder(cfg, project_dim=int(g['project_dim']), with_image=False)
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd__')}
    res = enc.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    enc.eval()
    t = lambda k: torch.from_numpy(g[k])
    with torch.no_grad():
        seq, pooled, hidden = enc(t('input_ids'), t('attention_mask'), t('position_ids'))
    assert hidden is None
    valid = t('attention_mask').bool()
    # (padded QUERY positions are never read downstream; HF masks keys with finfo.min, the 2.3.0 code with -10000: same softmax)
    np.testing.assert_allclose(seq[valid].numpy(), g['seq'][valid.numpy()], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(pooled.numpy(), g['pooled'], rtol=1e-4, atol=2e-5)


def test_checkpoint_key_manifest_of_real_config(golden_dir):
    manifest = json.load(open(os.path.join(golden_dir, 'g5_img_tower_manifest.json')))
    with torch.device('meta'):
        args = types.SimpleNamespace(img_model_type='uniter-base', txt_model_type='bert-base')
        be = BiEncoder(args, project_dim=768, txt_config=TowerConfig(), img_config=TowerConfig())
    mine = {k: list(v.shape) for k, v in be.state_dict().items()}
    img = {k: v for k, v in mine.items() if k.startswith('img_model.')}
    assert img == manifest                                   # every key and shape of the reference image tower
    # text tower: every key and shape of the reference's BertEncoder (HF BertModel + encode_proj) at bert-base-cased size (G9)
    txt_manifest = json.load(open(os.path.join(golden_dir, 'g9_txt_tower_manifest.json')))
    assert {k: v for k, v in mine.items() if k.startswith('txt_model.')} == txt_manifest
    assert set(mine) == set(img) | set(txt_manifest)         # flickr-ft.pt['model_dict'] = exactly these two towers
    assert sum(int(np.prod(s)) for s in img.values()) == 112263424     # SURVEY §5 (measured on the reference)


def test_checkpoint_roundtrip_and_pretrain_prefix_fallback(tmp_path, golden_dir):
    cfg = TowerConfig(vocab_size=50, hidden_size=32, num_hidden_layers=1, num_attention_heads=4, intermediate_size=64,
                      max_position_embeddings=16)
    args = types.SimpleNamespace(img_model_type='uniter-base', txt_model_type='bert-base')
    a = BiEncoder(args, project_dim=16, txt_config=cfg, img_config=cfg)
    opt = torch.optim.AdamW(a.parameters(), lr=1e-3)
    path = save_checkpoint(a, opt, None, epoch=3, offset=0, path=str(tmp_path / 'biencoder.3.pt'))
    state = torch.load(path, map_location='cpu')
    assert list(state.keys()) == list(CheckpointState._fields) and state['epoch'] == 3
    b = BiEncoder(args, project_dim=16, txt_config=cfg, img_config=cfg)
    load_biencoder_checkpoint(b, path)
    for (k1, v1), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    # pre-training layout: flat dict, BiEncoder keys under a leading 'bert.', plus unrelated heads that are dropped
    flat = {('bert.' + k): v for k, v in a.state_dict().items()}
    flat['cls.predictions.bias'] = torch.zeros(3)
    c = BiEncoder(args, project_dim=16, txt_config=cfg, img_config=cfg)
    load_biencoder_checkpoint(c, flat)
    assert all(torch.equal(v, c.state_dict()[k]) for k, v in a.state_dict().items())


def test_biencoder_forward_batch_contract():
    """dvl/data/itm.py:254-287 batch layout -> (txt_pooled, img_pooled, cap_pooled)."""
    cfg = TowerConfig(vocab_size=50, hidden_size=32, num_hidden_layers=1, num_attention_heads=4, intermediate_size=64,
                      max_position_embeddings=16, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    args = types.SimpleNamespace(img_model_type='uniter-base', txt_model_type='bert-base')
    be = BiEncoder(args, project_dim=16, txt_config=cfg, img_config=cfg).eval()
    B, Lt, nbb = 4, 7, 5
    batch = {
        'txts': dict(input_ids=torch.randint(1, 50, (B, Lt)), position_ids=torch.arange(Lt).unsqueeze(0),
                     attention_mask=torch.ones(B, Lt, dtype=torch.long), img_feat=None, img_pos_feat=None, img_masks=None,
                     gather_index=None),
        'imgs': dict(input_ids=torch.full((B, 1), 3), position_ids=torch.zeros(1, 1, dtype=torch.long),
                     attention_mask=torch.ones(B, 1 + nbb, dtype=torch.long), img_feat=torch.randn(B, nbb, 2048),
                     img_pos_feat=torch.rand(B, nbb, 7), img_masks=None,
                     gather_index=torch.arange(1 + nbb).unsqueeze(0).repeat(B, 1)),
        'caps': dict(input_ids=None, position_ids=None, attention_mask=None, img_feat=None, img_pos_feat=None,
                     img_masks=None, gather_index=None),
        'sample_size': B, 'pos_ctx_indices': list(range(B)), 'neg_ctx_indices': [],
        'txt_index': list(range(B)), 'img_fname': [f'i{j}' for j in range(B)]}
    with torch.no_grad():
        t, i, c = be(batch)
    assert t.shape == (B, 16) and i.shape == (B, 16) and c is None
