#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — golden-vector generator.  Runs ONLY in the build container, where the
reference checkout exists at /root/reference.  It imports the reference's own Python modules (with the
sys.modules stubs of oracle/_ref_stubs.py for the third-party packages that are not installed),
feeds them seeded inputs and writes inputs + the reference's outputs as small fixtures under
tests/golden/.  The fixtures are data (arrays / JSON); no reference source travels.

    python oracle/gen_golden.py            # rewrites tests/golden/*

Golden sets:
  G1 loss            BiEncoderNllLoss.calc            dvl/models/bi_encoder.py:615-656  (+ autograd grads)
  G2 train step      the composition at               train_itm.py:195-222  (calls dvl/utils.py:_calc_loss)
  G3 recall harness  eval_model_on_dataloader         dvl/trainer.py:113-190
  G4 hard negatives  sampled_hard_negatives           dvl/hn.py:45-66
  G5 pooling+proj    UniterEncoder.forward            dvl/models/bi_encoder.py:131-191 (small config)
                     + state_dict key/shape manifest of the real config (checkpoint surface)
  G6 config surface  parse_with_config                dvl/options.py:96-109
  G7 indexer wrapper DenseFlatIndexer                 dvl/indexer/faiss_indexers.py:63-87 (numpy IndexFlatIP stand-in)
  G8 DB readers      TxtTokLmdb / DetectFeatLmdb / ItmFastDataset.new_epoch + __getitem__ / itm_fast_collate
                                                       uniter_model/data/data.py:44-125,177-246 ; dvl/data/itm.py:30-122,203-288
                     (the reference's own classes over an in-memory stand-in for the lmdb container)
  G9 text tower      BertEncoder.forward              dvl/models/bi_encoder.py:76-128 (the reference class over the installed
                     transformers.BertModel): small seeded model -> outputs + state_dict; key/shape manifest at bert-base-cased size
"""
import argparse
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stubs  # noqa: E402

_ref_stubs.install()

import dvl.hn as ref_hn  # noqa: E402
import dvl.indexer.faiss_indexers as ref_idx  # noqa: E402
import dvl.models.bi_encoder as ref_be  # noqa: E402
import dvl.options as ref_opt  # noqa: E402
import dvl.trainer as ref_tr  # noqa: E402
import dvl.utils as ref_utils  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(1)
torch.manual_seed(0)


def _t(a):
    return torch.from_numpy(np.asarray(a))


# ------------------------------------------------------------------ G1
def g1_loss():
    rng = np.random.default_rng(101)
    cases = []
    spec = [
        # n1, n2, D, has_cap, w, reduction, positive pattern
        (8, 8, 768, False, 0.1, 'mean', 'diag'),
        (16, 48, 768, False, 0.0, 'mean', 'diag'),      # hard-negative shaped: bs x (bs + bs*nh), nh=2
        (16, 16, 768, True, 0.1, 'mean', 'diag'),
        (16, 16, 768, True, 0.0, 'mean', 'diag'),       # w == 0 disables caption mixing
        (12, 20, 64, True, 0.25, 'none', 'rand'),
        (80, 80, 768, False, 0.1, 'mean', 'diag'),      # eval batch shape (dvl/options.py:26)
        (96, 96, 768, True, 0.1, 'sum', 'perm'),
        (5, 7, 32, False, 0.1, 'mean', 'tie'),          # exact ties in the arg-max
    ]
    out = {}
    for ci, (n1, n2, d, has_cap, w, red, pat) in enumerate(spec):
        q = (rng.standard_normal((n1, d)) * 0.3).astype(np.float32)
        ctx = (rng.standard_normal((n2, d)) * 0.3).astype(np.float32)
        cap = (rng.standard_normal((n2, d)) * 0.3).astype(np.float32) if has_cap else None
        if pat == 'diag':
            pos = list(range(n1))
            ctx[:n1] += q  # make positives likely, like a trained model
        elif pat == 'rand':
            pos = [int(x) for x in rng.integers(0, n2, size=n1)]
        elif pat == 'perm':
            pos = [int(x) for x in rng.permutation(n2)[:n1]]
        else:  # tie: duplicate ctx rows so two columns have identical scores
            pos = [1, 0, 2, 3, 4]
            ctx[1] = ctx[0]
        tq, tc = _t(q).requires_grad_(), _t(ctx).requires_grad_()
        tcap = _t(cap).requires_grad_() if has_cap else None
        loss, correct, scores = ref_be.BiEncoderNllLoss().calc(tq, tc, tcap, pos, None, w, None, red)
        gl = torch.ones_like(loss) if red != 'none' else _t(rng.standard_normal(n1).astype(np.float32))
        gs = _t((rng.standard_normal((n1, n2)) * 0.01).astype(np.float32))
        # loss AND scores both receive upstream gradients (scores feeds KD in train_itm.py:236-239)
        torch.autograd.backward([loss, scores], [gl, gs])
        p = f'c{ci}_'
        out[p + 'q'], out[p + 'ctx'] = q, ctx
        if has_cap:
            out[p + 'cap'] = cap
        out[p + 'pos'] = np.asarray(pos, np.int64)
        out[p + 'w'] = np.float64(w)
        out[p + 'loss'] = loss.detach().numpy()
        out[p + 'correct'] = np.int64(correct.item())
        out[p + 'scores'] = scores.detach().numpy()
        out[p + 'gl'] = gl.numpy()
        out[p + 'gs'] = gs.numpy()
        out[p + 'dq'] = tq.grad.numpy()
        out[p + 'dctx'] = tc.grad.numpy()
        if has_cap and tcap.grad is not None:
            out[p + 'dcap'] = tcap.grad.numpy()
        cases.append(dict(n1=n1, n2=n2, d=d, has_cap=has_cap, w=w, reduction=red, pattern=pat))
    np.savez_compressed(os.path.join(OUT, 'g1_loss.npz'), **out)
    json.dump(cases, open(os.path.join(OUT, 'g1_loss_cases.json'), 'w'), indent=1)


# ------------------------------------------------------------------ G2
def g2_train_step():
    """Replays train_itm.py:195-222 verbatim in structure, through the reference's _calc_loss."""
    rng = np.random.default_rng(202)
    out, cases = {}, []
    for ci, (bs, nh, w) in enumerate([(8, 0, 0.0), (8, 2, 0.0), (16, 2, 0.0), (12, 1, 0.0)]):
        n = bs + bs * nh
        txt = (rng.standard_normal((n, 768)) * 0.3).astype(np.float32)
        img = (txt + rng.standard_normal((n, 768)) * 0.2).astype(np.float32)
        args = types.SimpleNamespace(caption_score_weight=w, num_hard_negatives=nh)
        pos = list(range(bs))
        neg = [[bs + i * nh + j for j in range(nh)] for i in range(bs)]
        lf = ref_be.BiEncoderNllLoss()
        txt_vector, img_vectors, caption_vectors = _t(txt), _t(img), None
        if args.num_hard_negatives > 0:
            l_t, c_t, s_t = ref_utils._calc_loss(args, lf, img_vectors[:bs], txt_vector, caption_vectors, pos, neg, None)
            l_i, c_i, s_i = ref_utils._calc_loss(args, lf, txt_vector[:bs], img_vectors, caption_vectors, pos, neg, None)
        else:
            l_t, c_t, s_t = ref_utils._calc_loss(args, lf, img_vectors, txt_vector, caption_vectors, pos, neg, None)
            l_i, c_i, s_i = ref_utils._calc_loss(args, lf, txt_vector, img_vectors, caption_vectors, pos, neg, None)
        is_correct = (c_t.sum().item() + c_i.sum().item()) / 2
        loss_nce = 0.5 * l_t + 0.5 * l_i
        scores = s_t * 0.5 + s_i * 0.5
        p = f'c{ci}_'
        out[p + 'txt'], out[p + 'img'] = txt, img
        out[p + 'loss'] = loss_nce.numpy()
        out[p + 'loss_txt'], out[p + 'loss_img'] = l_t.numpy(), l_i.numpy()
        out[p + 'is_correct'] = np.float64(is_correct)
        out[p + 'scores'] = scores.numpy()
        cases.append(dict(bs=bs, nh=nh, w=w))
    np.savez_compressed(os.path.join(OUT, 'g2_train_step.npz'), **out)
    json.dump(cases, open(os.path.join(OUT, 'g2_train_step_cases.json'), 'w'), indent=1)


# ------------------------------------------------------------------ G3
class _FakeBiEncoder:
    """Stands in for the two towers: returns the vectors carried by the batch."""

    def eval(self):
        return self

    def __call__(self, batch):
        return batch['_q'], batch['_ctx'], batch.get('_cap')


def _make_stream(seed, n_img, caps_per_img, d, batch, noise, with_cap=False, dup_tail=0):
    rng = np.random.default_rng(seed)
    img = rng.standard_normal((n_img, d)).astype(np.float32) * 0.5
    items = []
    for i in range(n_img):
        for c in range(caps_per_img):
            t = (img[i] + noise * rng.standard_normal(d)).astype(np.float32)
            items.append((f'txt{i:04d}_{c}', f'img{i:04d}.npz', t, img[i]))
    order = rng.permutation(len(items))
    items = [items[j] for j in order]
    items = items + items[:dup_tail]      # duplicated items across batches (dict overwrite path)
    img2txt = {}
    for tid, iid, _, _ in items:
        img2txt.setdefault(iid, [])
        if tid not in img2txt[iid]:
            img2txt[iid].append(tid)
    batches = []
    for b0 in range(0, len(items), batch):
        chunk = items[b0:b0 + batch]
        q = np.stack([c[2] for c in chunk])
        ctx = np.stack([c[3] for c in chunk])
        bd = dict(txt_index=[c[0] for c in chunk], img_fname=[c[1] for c in chunk], q=q, ctx=ctx)
        if with_cap:
            bd['cap'] = (ctx + 0.1 * rng.standard_normal(ctx.shape)).astype(np.float32)
        batches.append(bd)
    return batches, img2txt


def g3_recall():
    out_json = {}
    arrs = {}
    for name, kw, num_tops, w in [
        ('small', dict(seed=7, n_img=40, caps_per_img=5, d=64, batch=16, noise=0.9, dup_tail=5), 20, 0.0),
        ('cap', dict(seed=8, n_img=24, caps_per_img=5, d=48, batch=10, noise=1.1, with_cap=True), 30, 0.1),
        ('ktoobig', dict(seed=9, n_img=6, caps_per_img=5, d=32, batch=8, noise=0.7), 100, 0.0),
    ]:
        batches, img2txt = _make_stream(**kw)
        ref_batches = []
        for b in batches:
            rb = dict(txt_index=list(b['txt_index']), img_fname=list(b['img_fname']),
                      txts={'input_ids': torch.zeros(len(b['txt_index']), 4, dtype=torch.long)},
                      _q=_t(b['q']), _ctx=_t(b['ctx']))
            if 'cap' in b:
                rb['_cap'] = _t(b['cap'])
            ref_batches.append(rb)
        args = types.SimpleNamespace(hnsw_index=False, vector_size=kw['d'], caption_score_weight=w)
        loss, acc, _idx, (r_txt, r_img), (rank_txt, rank_img) = ref_tr.eval_model_on_dataloader(
            _FakeBiEncoder(), ref_batches, args, img2txt, num_tops)
        out_json[name] = dict(
            kw=kw, num_tops=num_tops, w=w, loss=float(loss), acc=float(acc),
            recall_txt={str(k): float(v) for k, v in r_txt.items()},
            recall_img={str(k): float(v) for k, v in r_img.items()},
            rank_txt=rank_txt, rank_img=rank_img)
    json.dump(out_json, open(os.path.join(OUT, 'g3_recall.json'), 'w'))


# ------------------------------------------------------------------ G4
def g4_hardneg():
    rng = random.Random(44)
    n_img, cpi, nh = 30, 5, 3
    img_ids = [f'img{i:03d}' for i in range(n_img)]
    img2txt = {iid: [f't{i:03d}_{c}' for c in range(cpi)] for i, iid in enumerate(img_ids)}
    txt2img = {t: k for k, v in img2txt.items() for t in v}
    all_txt = list(txt2img)
    n_top = ref_hn.min(ref_hn.max(nh * 2 + 10, 50), 1000) if hasattr(ref_hn, 'min') else min(max(nh * 2 + 10, 50), 1000)
    # synthetic retrieval results shaped like eval_model_on_dataloader's rank dicts
    hard_neg_img = {t: rng.sample(img_ids, 20) for t in all_txt}            # txt -> ranked images
    for t in all_txt[::2]:                                                 # positives present in half of them
        if txt2img[t] not in hard_neg_img[t]:
            hard_neg_img[t][3] = txt2img[t]
    hard_neg_txt = {i: rng.sample(all_txt, 25) + img2txt[i][:2] for i in img_ids}   # img -> ranked texts (+ own caps)
    captured = {'pops': []}

    def fake_eval(bi_encoder, loader, args, img2txt_, num_tops):
        captured['num_tops'] = num_tops
        return 0.0, 0.0, (None, None), (None, None), (
            {k: list(v) for k, v in hard_neg_img.items()}, {k: list(v) for k, v in hard_neg_txt.items()})

    def fake_sample(pop, k):
        captured['pops'].append(sorted(pop))
        return sorted(pop)[:k]

    class _DS:
        datasets = [types.SimpleNamespace(new_epoch=lambda *a, **k: None)]

    saved = (ref_hn.load_dataset, ref_hn.build_dataloader, ref_hn.eval_model_on_dataloader, ref_hn.random.sample)
    ref_hn.load_dataset = lambda *a, **k: _DS()
    ref_hn.build_dataloader = lambda *a, **k: [0]
    ref_hn.eval_model_on_dataloader = fake_eval
    ref_hn.random.sample = fake_sample
    try:
        args = types.SimpleNamespace(num_hard_negatives=nh, train_txt_dbs=['a'], train_img_dbs=['a'],
                                     valid_batch_size=8)
        hn_txt_all, hn_img_all = ref_hn.sampled_hard_negatives(None, args, None, None, img2txt, txt2img)
    finally:
        ref_hn.load_dataset, ref_hn.build_dataloader, ref_hn.eval_model_on_dataloader, ref_hn.random.sample = saved
    # populations are captured in call order: first all hard_neg_txt items (img keys), then hard_neg_img (txt keys)
    pops_txt = dict(zip(hard_neg_txt.keys(), captured['pops'][:len(hard_neg_txt)]))
    pops_img = dict(zip(hard_neg_img.keys(), captured['pops'][len(hard_neg_txt):]))
    json.dump(dict(nh=nh, num_tops=captured['num_tops'], img2txt=img2txt, txt2img=txt2img,
                   hard_neg_img=hard_neg_img, hard_neg_txt=hard_neg_txt,
                   pops_txt_sorted=pops_txt, pops_img_sorted=pops_img,
                   out_txt=hn_txt_all, out_img=hn_img_all),
              open(os.path.join(OUT, 'g4_hardneg.json'), 'w'))


# ------------------------------------------------------------------ G5
def g5_pool_proj():
    from uniter_model.model.model import UniterConfig
    torch.manual_seed(5)
    small = dict(vocab_size_or_config_json_file=120, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                 intermediate_size=128, hidden_act='gelu', hidden_dropout_prob=0.0,
                 attention_probs_dropout_prob=0.0, max_position_embeddings=64, type_vocab_size=2,
                 initializer_range=0.02)
    cfg = UniterConfig(**small)
    cfg.output_hidden_states = False
    enc = ref_be.UniterEncoder(cfg, project_dim=32)
    # make LayerNorm affine + biases non-trivial so the restatement is really exercised
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if 'encode_proj' in n:
                p.add_(torch.randn_like(p) * 0.05)
    enc.eval()
    B, nbb = 3, 20
    g = torch.Generator().manual_seed(55)
    input_ids = torch.full((B, 1), 101, dtype=torch.long) % 120
    position_ids = torch.zeros(1, 1, dtype=torch.long)
    img_feat = torch.randn(B, nbb, 2048, generator=g)
    img_pos_feat = torch.rand(B, nbb, 7, generator=g)
    attn = torch.ones(B, 1 + nbb, dtype=torch.long)
    attn[1, 15:] = 0
    gather_index = torch.arange(0, 1 + nbb, dtype=torch.long).unsqueeze(0).repeat(B, 1)
    with torch.no_grad():
        seq, pooled, _ = enc(input_ids, attn, position_ids, img_feat, img_pos_feat, None, gather_index)
    sd = enc.state_dict()
    np.savez_compressed(
        os.path.join(OUT, 'g5_pool_proj.npz'),
        seq=seq.numpy(), pooled=pooled.numpy(),
        w0=sd['encode_proj.0.weight'].numpy(), b0=sd['encode_proj.0.bias'].numpy(),
        ln_g=sd['encode_proj.2.weight'].numpy(), ln_b=sd['encode_proj.2.bias'].numpy(),
        w3=sd['encode_proj.3.weight'].numpy(), b3=sd['encode_proj.3.bias'].numpy())
    # full small tower: every parameter + inputs + outputs of the image path AND of the text-only path (the reference's
    # `txt_model_type == 'uniter-base'` option, bi_encoder.py:216-217 — same BERT math and key names as the text tower)
    txt_ids = torch.randint(1, 120, (B, 9), generator=g)
    txt_ids[:, 0] = 101 % 120
    txt_pos = torch.arange(0, 9, dtype=torch.long).unsqueeze(0)
    txt_attn = torch.ones(B, 9, dtype=torch.long)
    txt_attn[2, 6:] = 0
    with torch.no_grad():
        tseq, tpooled, _ = enc(txt_ids, txt_attn, txt_pos, None, None, None, None)
    small_sd = {('sd__' + k): v.numpy() for k, v in sd.items()}
    np.savez_compressed(
        os.path.join(OUT, 'g5_tower_small.npz'), cfg=json.dumps(small), project_dim=np.int64(32),
        img_input_ids=input_ids.numpy(), img_position_ids=position_ids.numpy(), img_feat=img_feat.numpy(),
        img_pos_feat=img_pos_feat.numpy(), img_attn=attn.numpy(), gather_index=gather_index.numpy(),
        img_seq=seq.numpy(), img_pooled=pooled.numpy(),
        txt_input_ids=txt_ids.numpy(), txt_position_ids=txt_pos.numpy(), txt_attn=txt_attn.numpy(),
        txt_seq=tseq.numpy(), txt_pooled=tpooled.numpy(), **small_sd)
    # checkpoint surface: key/shape manifest of the REAL image tower config (config/img_base.json)
    real = ref_be.UniterEncoder(UniterConfig(os.path.join(_ref_stubs.REF_ROOT, 'config', 'img_base.json')),
                                project_dim=768)
    manifest = {('img_model.' + k): list(v.shape) for k, v in real.state_dict().items()}
    json.dump(manifest, open(os.path.join(OUT, 'g5_img_tower_manifest.json'), 'w'), indent=0)


# ------------------------------------------------------------------ G6
def g6_config():
    parser = argparse.ArgumentParser()
    ref_opt.default_params(parser)
    ref_opt.add_itm_params(parser)
    ref_opt.add_logging_params(parser)
    ref_opt.add_kd_params(parser)
    cfg = os.path.join(_ref_stubs.REF_ROOT, 'config', 'flickr30k_eval_config.json')
    saved = sys.argv
    out = {}
    try:
        sys.argv = ['eval_itm.py', '--config', cfg]
        out['plain'] = vars(ref_opt.parse_with_config(parser, ['--config', cfg]))
        # a CLI override beats the JSON only when it is on the REAL sys.argv (options.py:103-104)
        sys.argv = ['eval_itm.py', '--config', cfg, '--max_txt_len', '32', '--project_dim=256']
        out['override'] = vars(ref_opt.parse_with_config(parser, sys.argv[1:]))
        sys.argv = ['eval_itm.py']
        out['cmds_only'] = vars(ref_opt.parse_with_config(parser, ['--config', cfg, '--max_txt_len', '32']))
    finally:
        sys.argv = saved
    for v in out.values():
        v['config'] = 'config/flickr30k_eval_config.json'
    json.dump(out, open(os.path.join(OUT, 'g6_config.json'), 'w'), indent=1, sort_keys=True)


# ------------------------------------------------------------------ G7
def g7_indexer():
    rng = np.random.default_rng(77)
    d = 48
    data = [(f'id{i:03d}', rng.standard_normal(d).astype(np.float32)) for i in range(57)]
    q = rng.standard_normal((9, d)).astype(np.float32)
    out = {}
    for k in (1, 10, 57, 64):
        ix = ref_idx.DenseFlatIndexer(d, buffer_size=20)    # several add() chunks
        ix.index_data(data)
        res = ix.search_knn(q, k)
        out[str(k)] = dict(ids=[r[0] for r in res], scores=[np.asarray(r[1], np.float32).tolist() for r in res])
    np.savez_compressed(os.path.join(OUT, 'g7_indexer_inputs.npz'), x=np.stack([v for _, v in data]), q=q)
    json.dump(dict(ids=[i for i, _ in data], results=out), open(os.path.join(OUT, 'g7_indexer.json'), 'w'))


# ------------------------------------------------------------------ G8
class FakeTokenizer:
    """stands in for BertTokenizer in the caption branch (dvl/data/itm.py:92-96,113-115): only encode / cls / sep are used"""
    cls_token_id, sep_token_id = 101, 102

    def encode(self, text, add_special_tokens=False):
        return [200 + (sum(map(ord, w)) % 300) for w in text.split()]


def _flatten_batch(b):
    """collated batch dict -> {dotted key: ndarray} + the python-level fields"""
    arrs, meta = {}, {}
    for k, v in b.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                if vv is None:
                    meta[f'{k}.{kk}'] = None
                else:
                    arrs[f'{k}.{kk}'] = vv.numpy()
        else:
            meta[k] = v
    return arrs, meta


def g8_itm_data():
    import io
    import tempfile
    import msgpack
    import dvl.data.itm as ref_itm
    import uniter_model.data.data as ref_data
    rng = np.random.default_rng(808)
    FD, n_img, cpi = 16, 9, 4                       # small feature width: the readers / collate never look at it
    conf_th, max_bb, min_bb = 0.2, 9, 4
    imgs, nbb = {}, {}
    for i in range(n_img):
        f = f'img_{i:03d}.npz'
        n = int(rng.integers(5, 13))
        conf = rng.uniform(0.0, 1.0, n).astype(np.float16)
        imgs[f] = dict(features=rng.standard_normal((n, FD)).astype(np.float16),
                       norm_bb=rng.uniform(0, 1, (n, 6)).astype(np.float16), conf=conf)
        nbb[f] = int(min(max_bb, max(min_bb, int((conf > conf_th).sum()))))
        nbb[f] = min(nbb[f], n)                      # (a prepro'd nbb never exceeds the stored regions)
    examples, id2len, txt2img, img2txts = {}, {}, {}, {}
    for i in range(n_img):
        f = f'img_{i:03d}.npz'
        for c in range(cpi):
            tid = f'{i * cpi + c}'
            n_tok = int(rng.integers(3, 16))
            examples[tid] = dict(id=tid, img_fname=f, sent=f'caption {c} of {i}',
                                 input_ids=[int(t) for t in rng.integers(1000, 28000, n_tok)])
            id2len[tid], txt2img[tid] = n_tok, f
            img2txts.setdefault(f, []).append(tid)
    meta = {'CLS': 101, 'SEP': 102, 'MASK': 103, 'v_range': [106, 28996]}
    max_txt_len = 12
    kept = [t for t in examples if id2len[t] <= max_txt_len]
    hn_img = {t: [f for f in imgs if f != txt2img[t]][int(t) % 3:][:3] for t in kept}
    hn_txt = {f: [t for t in kept if txt2img[t] != f][i::5][:3] for i, f in enumerate(imgs)}
    img_meta = {f: {'caption_multiple': [f'a photo number {i}', f'second caption for {f[:7]} with more words']}
                for i, f in enumerate(imgs)}

    tmp = tempfile.mkdtemp()
    txt_dir, img_dir = os.path.join(tmp, 'txt.db'), os.path.join(tmp, 'img')
    os.makedirs(txt_dir)
    os.makedirs(img_dir)
    for name, obj in (('id2len', id2len), ('txt2img', txt2img), ('img2txts', img2txts), ('meta', meta)):
        json.dump(obj, open(os.path.join(txt_dir, name + '.json'), 'w'))
    json.dump(nbb, open(os.path.join(img_dir, f'nbb_th{conf_th}_max{max_bb}_min{min_bb}.json'), 'w'))
    # the container stand-in: text values are msgpack (lz4.frame is stubbed to the identity on the reference side; the build's reader
    # gets real LZ4 frames of the same msgpack bytes), image values npz archives / msgpack_numpy dicts
    reg = _ref_stubs.FakeLmdb.registry
    reg[os.path.normpath(txt_dir)] = {k.encode(): msgpack.dumps(v, use_bin_type=True) for k, v in examples.items()}
    npz = {}
    for f, d in imgs.items():
        buf = io.BytesIO()
        np.savez_compressed(buf, **d)
        npz[f.encode()] = buf.getvalue()
    reg[os.path.normpath(f'{img_dir}/feat_th{conf_th}_max{max_bb}_min{min_bb}_compressed')] = npz

    def mnp(o):                                      # msgpack-numpy 0.4.6's ndarray layout (bytes keys, use_bin_type)
        if isinstance(o, np.ndarray):
            return {b'nd': True, b'type': o.dtype.str, b'kind': b'', b'shape': list(o.shape), b'data': o.tobytes()}
        raise TypeError
    reg[os.path.normpath(f'{img_dir}/feat_th{conf_th}_max{max_bb}_min{min_bb}')] = {
        f.encode(): msgpack.dumps(d, default=mnp, use_bin_type=True) for f, d in imgs.items()}

    out_arr, out_meta = {}, {}
    for flavour, compress in (('npz', True), ('msgpack', False)):
        txt_db = ref_data.TxtTokLmdb(txt_dir, max_txt_len)
        img_db = ref_data.DetectFeatLmdb(img_dir, conf_th, max_bb, min_bb, 36, compress)
        assert txt_db.ids == kept
        f0 = list(imgs)[2]
        feat, bb = img_db[f0]
        dump = img_db.get_dump(f0)
        out_arr[f'{flavour}.getitem.feat'], out_arr[f'{flavour}.getitem.bb'] = feat.numpy(), bb.numpy()
        for k, v in dump.items():
            out_arr[f'{flavour}.get_dump.{k}'] = np.asarray(v)
        out_meta[f'{flavour}.getitem.fname'] = f0
        cases = {}
        # (a) evaluation style: no negatives, consecutive items
        ds = ref_itm.ItmFastDataset(txt_db, img_db, num_hard_negatives=2)
        ds.new_epoch()
        cases['eval'] = ([0, 1, 2, 3, 4, 5], ds)
        # (b) training style: nh = 2 of each kind per item, arbitrary item order
        ds2 = ref_itm.ItmFastDataset(txt_db, img_db, num_hard_negatives=2)
        ds2.new_epoch(hn_img, hn_txt)
        cases['train'] = ([7, 0, 11, 3, 20], ds2)
        # (c) with the caption branch (img_meta + tokenizer)
        ds3 = ref_itm.ItmFastDataset(txt_db, img_db, num_hard_negatives=1, img_meta=img_meta, tokenizer=FakeTokenizer())
        ds3.new_epoch(hn_img, hn_txt)
        cases['caps'] = ([2, 9, 4], ds3)
        for cname, (idx, d) in cases.items():
            batch = ref_itm.itm_fast_collate([d[i] for i in idx])
            arrs, m = _flatten_batch(batch)
            for k, v in arrs.items():
                out_arr[f'{flavour}.{cname}.{k}'] = v
            m.update(items=idx, lens=[int(x) for x in d.lens], ids=list(d.ids), train_imgs=list(d.train_imgs),
                     all_imgs=sorted(d.all_imgs))
            out_meta[f'{flavour}.{cname}'] = m
    inputs = dict(feat_dim=FD, conf_th=conf_th, max_bb=max_bb, min_bb=min_bb, max_txt_len=max_txt_len, examples=examples,
                  id2len=id2len, txt2img=txt2img, img2txts=img2txts, meta=meta, nbb=nbb, hn_img=hn_img, hn_txt=hn_txt,
                  img_meta=img_meta, kept=kept)
    json.dump(dict(inputs=inputs, expected=out_meta), open(os.path.join(OUT, 'g8_itm_data.json'), 'w'))
    img_arr = {f'img.{f}.{k}': v for f, d in imgs.items() for k, v in d.items()}
    np.savez_compressed(os.path.join(OUT, 'g8_itm_data.npz'), **img_arr, **out_arr)


# ------------------------------------------------------------------ G9
def g9_text_tower():
    """The reference's text tower: ``BertEncoder`` = transformers.BertModel + [CLS] pooling + encode_proj
    (dvl/models/bi_encoder.py:76-128; written for transformers==2.3.0, DVL.yml:180 — constructed here with the shim of
    _ref_stubs.patch_bert_encoder; the BertModel of the installed transformers has the same parameter names and math)."""
    from transformers import BertConfig
    _ref_stubs.patch_bert_encoder(ref_be)
    torch.manual_seed(9)
    small = dict(vocab_size=120, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                 hidden_act='gelu', hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, max_position_embeddings=64,
                 type_vocab_size=2, initializer_range=0.02)
    cfg = BertConfig(return_dict=False, **small)
    cfg.output_hidden_states = False
    model = ref_be.BertEncoder(cfg, project_dim=32).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():        # non-trivial LayerNorm affines and biases
            if 'LayerNorm' in n or n.endswith('.bias') or 'encode_proj' in n:
                p.add_(torch.randn_like(p) * 0.05)
    g = torch.Generator().manual_seed(99)
    B, L = 4, 11
    ids = torch.randint(1, 120, (B, L), generator=g)
    attn = torch.ones(B, L, dtype=torch.long)
    attn[1, 7:] = 0
    attn[3, 4:] = 0
    ids = ids * attn                                 # pad id 0 like pad_sequence in the collate
    pos = torch.arange(0, L, dtype=torch.long).unsqueeze(0)
    with torch.no_grad():
        seq, pooled, hidden = model(ids, attn, pos)
    assert hidden is None and model.get_out_size is not None
    sd = {('sd__' + k): v.numpy() for k, v in model.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, 'g9_text_tower_small.npz'), cfg=json.dumps(small), project_dim=np.int64(32),
                        input_ids=ids.numpy(), attention_mask=attn.numpy(), position_ids=pos.numpy(), seq=seq.numpy(),
                        pooled=pooled.numpy(), **sd)
    # strict-load surface of the real text tower: BertEncoder at bert-base-cased size (vocab 28996), as BiEncoder.txt_model holds it
    with torch.device('meta'):
        rcfg = BertConfig(vocab_size=28996, return_dict=False)
        rcfg.output_hidden_states = False
        real = ref_be.BertEncoder(rcfg, project_dim=768)
    manifest = {('txt_model.' + k): list(v.shape) for k, v in real.state_dict().items()}
    json.dump(manifest, open(os.path.join(OUT, 'g9_txt_tower_manifest.json'), 'w'), indent=0)


if __name__ == '__main__':
    only = set(sys.argv[1:])
    if only:
        for name in sorted(only):
            globals()[name]()
        sys.exit(0)
    g1_loss()
    g2_train_step()
    g3_recall()
    g4_hardneg()
    g5_pool_proj()
    g6_config()
    g7_indexer()
    g8_itm_data()
    g9_text_tower()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
