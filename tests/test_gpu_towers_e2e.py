"""GPU end-to-end: small towers -> index -> fused/dense search -> Recall@k through the mirrored harness, against the
oracle harness fed with the SAME tower outputs; one fine-tuning step with hard negatives on the HIP loss path."""
import types

import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


def _tiny():
    from lightningdot_amd.towers import BiEncoder, TowerConfig
    cfg = TowerConfig(vocab_size=30000, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                      intermediate_size=128, max_position_embeddings=64, hidden_dropout_prob=0.0,
                      attention_probs_dropout_prob=0.0)
    args = types.SimpleNamespace(img_model_type='uniter-base', txt_model_type='bert-base')
    torch.manual_seed(0)
    return BiEncoder(args, project_dim=48, txt_config=cfg, img_config=cfg).cuda()


def test_eval_flow_matches_oracle_harness_on_same_vectors():
    from lightningdot_amd.harness import eval_model_on_dataloader
    from lightningdot_amd.synthetic import synthetic_itm_batches
    be = _tiny().eval()
    batches, img2txt = synthetic_itm_batches(60, caps_per_img=5, batch_size=32, txt_len=12, num_bb=10, device='cuda', seed=3)
    recorded = []

    class Rec(torch.nn.Module):
        def forward(self, batch):
            out = be(batch)
            recorded.append((batch['txt_index'], batch['img_fname'], out[0].float().cpu().numpy(), out[1].float().cpu().numpy()))
            return out

    args = types.SimpleNamespace(hnsw_index=False, vector_size=48, caption_score_weight=0.0)
    loss, acc, _, (r_txt, r_img), (rank_txt, rank_img) = eval_model_on_dataloader(Rec(), batches, args, img2txt, 20)
    stream = [dict(txt_index=t, img_fname=i, q=q, ctx=c) for t, i, q, c in recorded]
    l2, a2, _, (o_txt, o_img), (orank_txt, orank_img) = O.eval_on_stream(stream, 48, img2txt, 20, 0.0)
    assert abs(loss - l2) < 1e-4 and acc == a2
    assert r_txt == o_txt and r_img == o_img
    # rank lists agree except inside exact/near ties (random towers give close scores): compare as sets at depth 20
    agree = np.mean([rank_txt[k][:1] == orank_txt[k][:1] for k in rank_txt])
    assert agree > 0.99


def test_train_step_with_hard_negatives_reduces_loss():
    from lightningdot_amd.synthetic import synthetic_itm_batches
    from lightningdot_amd.train import get_optimizer, get_schedule_linear, train_step
    be = _tiny()
    batches, _ = synthetic_itm_batches(16, caps_per_img=1, batch_size=16, txt_len=12, num_bb=10, device='cuda', seed=4,
                                       num_hard_negatives=2)
    args = types.SimpleNamespace(caption_score_weight=0.0, num_hard_negatives=2, max_grad_norm=2.0,
                                 gradient_accumulation_steps=1)
    opt = get_optimizer(be, learning_rate=3e-3)
    sch = get_schedule_linear(opt, 2, 400)
    losses = [train_step(be, batches[0], args, opt, sch)[0] for _ in range(80)]
    assert np.isfinite(losses).all() and min(losses[-5:]) < losses[0] - 1.0, losses[::8]


def test_small_tower_on_gpu_matches_reference_golden(golden_dir):
    """a3 on the device: the seeded small UniterEncoder of golden G5 (outputs of the REFERENCE's own module) run on cuda — image
    path and text-only path, fp32 (incl. the HIP [CLS] pooling branch taken under no_grad) and under bf16 autocast."""
    import json
    import os
    from lightningdot_amd.towers import TowerConfig, TowerEncoder
    g = np.load(os.path.join(golden_dir, 'g5_tower_small.npz'))
    cfg = TowerConfig(**{('vocab_size' if k == 'vocab_size_or_config_json_file' else k): v
                         for k, v in json.loads(str(g['cfg'])).items()})
    enc = TowerEncoder(cfg, project_dim=int(g['project_dim']), with_image=True)
    enc.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd__')}, strict=True)
    enc = enc.cuda().eval()
    t = lambda k: torch.from_numpy(g[k]).cuda()
    with torch.no_grad():                                   # -> pool_cls (ldot_cls_pool) is the pooling that runs
        seq, pooled, _ = enc(t('img_input_ids'), t('img_attn'), t('img_position_ids'), t('img_feat'), t('img_pos_feat'),
                             None, t('gather_index'))
        tseq, tpooled, _ = enc(t('txt_input_ids'), t('txt_attn'), t('txt_position_ids'))
    np.testing.assert_allclose(seq.cpu().numpy(), g['img_seq'], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(pooled.cpu().numpy(), g['img_pooled'], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(tseq.cpu().numpy(), g['txt_seq'], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(tpooled.cpu().numpy(), g['txt_pooled'], rtol=1e-4, atol=5e-5)
    # with autograd on, the plain slice pools (training path): same numbers
    seq2, pooled2, _ = enc(t('img_input_ids'), t('img_attn'), t('img_position_ids'), t('img_feat'), t('img_pos_feat'),
                           None, t('gather_index'))
    np.testing.assert_allclose(pooled2.detach().cpu().numpy(), g['img_pooled'], rtol=1e-4, atol=5e-5)
    # bf16 autocast (the reference evaluates under apex fp16): loose tolerance, same ranking signal
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        _, pooled_bf, _ = enc(t('img_input_ids'), t('img_attn'), t('img_position_ids'), t('img_feat'), t('img_pos_feat'),
                              None, t('gather_index'))
    ref = g['img_pooled']
    err = np.abs(pooled_bf.float().cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 0.05, err


class _StubTokenizer:
    """whitespace tokenizer with the BERT [CLS]/[SEP] framing (the real vocabulary file is not available offline)"""

    def encode(self, text):
        return [101] + [1000 + (hash(w) % 20000) for w in text.lower().split()] + [102]


def test_retrieve_query_single_string_over_demo_sized_index():
    """a13 (dvl/utils.py:204-211): one string -> tokenizer -> text tower -> top-100 over an index of the demo's size
    (123 287 rows), checked against the exact fp64 top-100 of the same query vector."""
    from lightningdot_amd.indexer import DenseFlatIndexer
    from lightningdot_amd.serving import retrieve_query
    be = _tiny().eval()
    n = 123287
    g = torch.Generator().manual_seed(12)
    x = torch.randn(n, 48, generator=g)
    ix = DenseFlatIndexer(48)
    ix.index_tensor([f'coco_{i}.npz' for i in range(n)], x.cuda())
    args = types.SimpleNamespace(tokenizer=_StubTokenizer(), device=torch.device('cuda'))
    res = retrieve_query(be, 'two dogs play in the snow', ix, args, top=10)
    assert len(res) == 1 and len(res[0][0]) == 100 and res[0][1].shape == (100,)
    ids = args.tokenizer.encode('two dogs play in the snow')
    inp = torch.LongTensor(ids).cuda().unsqueeze(0)
    with torch.no_grad():
        _, qv, _ = be.txt_model(input_ids=inp, attention_mask=torch.ones_like(inp), position_ids=torch.arange(len(ids)).cuda().unsqueeze(0))
    full = (x.double() @ qv[0].double().cpu())
    top = torch.topk(full, 100)
    assert res[0][0] == [f'coco_{i}.npz' for i in top.indices.tolist()]
    np.testing.assert_allclose(res[0][1], top.values.float().numpy(), rtol=0, atol=1e-4)


def test_eval_model_entry_point_synthetic(tmp_path, capsys):
    """f1 (eval_itm.py:40-152): EVAL_MODEL end to end on synthetic batches — config surface, tower construction from the JSON
    configs, both partitions, the reference's prints; small towers so that the test stays in seconds."""
    import json
    from lightningdot_amd.eval_itm import EVAL_MODEL
    small = dict(vocab_size=28996, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                 max_position_embeddings=512, type_vocab_size=2, hidden_act='gelu', hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1, initializer_range=0.02)
    (tmp_path / 'img_small.json').write_text(json.dumps(small))
    cfg = dict(txt_model_config='bert-base-cased', img_model_config=str(tmp_path / 'img_small.json'), itm_global_file='unused.json',
               seed=42, output_dir=str(tmp_path / 'out'), max_txt_len=60, conf_th=0.2, max_bb=100, min_bb=10, num_bb=36,
               project_dim=64, val_txt_db='val.db', val_img_db='img/', test_txt_db='test.db', test_img_db='img/',
               project_name='itm-debug', n_workers=0, fp16=True)
    (tmp_path / 'eval.json').write_text(json.dumps(cfg))
    res = EVAL_MODEL(str(tmp_path / 'eval.json'), '', synthetic_images=40)
    out = capsys.readouterr().out
    assert set(res) == {'dev', 'test'}
    for part in res.values():
        assert set(part['recall_img']) == {1, 5, 10} and set(part['recall_txt']) == {1, 5, 10}
        assert 0.0 <= part['recall_mean'] <= 1.0 and np.isfinite(part['loss'])
    assert 'image retrieval recall =' in out and 'txt retrieval recall =' in out and 'indexed  40 data' in out


def test_train_itm_loop_on_gpu(tmp_path):
    """f1 (train_itm.py:176-358) with the HIP defaults: in-batch loss, per-epoch Recall@k through the retrieval harness, device-side
    hard-negative mining feeding new_epoch, checkpoints; synthetic dataset with learnable pairing."""
    import os
    from lightningdot_amd.data import batch_to_device, itm_fast_collate
    from lightningdot_amd.synthetic import SyntheticItmDataset
    from lightningdot_amd.towers import CheckpointState
    from lightningdot_amd.train_itm import TRAIN
    be = _tiny()
    dev = torch.device('cuda')
    train_ds = SyntheticItmDataset(48, caps_per_img=2, txt_len=12, num_bb=10, num_hard_negatives=2, seed=1)
    val_ds = SyntheticItmDataset(16, caps_per_img=2, txt_len=12, num_bb=10, seed=2)
    loader_of = lambda ds: [batch_to_device(itm_fast_collate([ds[i] for i in range(b, min(b + 16, len(ds)))]), dev)
                            for b in range(0, len(ds), 16)]

    def mining_loaders():
        saved = (train_ds.neg_imgs, train_ds.neg_txts)
        train_ds.new_epoch()
        out = loader_of(train_ds)
        train_ds.neg_imgs, train_ds.neg_txts = saved
        return [out]

    def args_for(tag, epochs, nh, lr):
        return types.SimpleNamespace(output_dir=str(tmp_path / tag), learning_rate=lr, num_train_epochs=epochs, train_batch_size=16,
                                     gradient_accumulation_steps=1, max_grad_norm=2.0, num_hard_negatives=nh,
                                     sample_init_hard_negatives=nh > 0, hard_negatives_sampling='hard' if nh else 'none',
                                     save_all_epochs=False, seed=3, distributed_world_size=1, caption_score_weight=0.0,
                                     log_result_step=100, vector_size=48, hnsw_index=False)

    # (1) in-batch negatives only: the loop learns the synthetic pairing (loss well below its chance level ln 16)
    train_ds.num_hard_negatives = 0
    args = args_for('plain', 8, 0, 2e-3)
    hist = TRAIN(args, be, train_ds, loader_of(val_ds), val_ds.img2txts, train_img2txt=train_ds.img2txts,
                 train_txt2img=train_ds.txt2img, mining_loaders=mining_loaders)
    assert [h['epoch'] for h in hist] == list(range(8)) and not any(h['hard_negatives'] for h in hist)
    assert hist[-1]['loss'] < hist[0]['loss'] - 0.3, [h['loss'] for h in hist]
    assert all(set(h['recall']) == {1, 5, 10} for h in hist)
    st = torch.load(os.path.join(args.output_dir, 'biencoder.last.pt'), map_location='cpu')
    assert set(st) == set(CheckpointState._fields) and st['epoch'] == 7
    assert os.path.exists(os.path.join(args.output_dir, 'biencoder.best.pt'))
    # (2) mined hard negatives (initial mining + re-mining after every epoch, device-side): 16 positives + 32 appended negatives per
    # step, resumed from the run above
    train_ds.num_hard_negatives = 2
    before = torch.cat([p.detach().reshape(-1) for p in be.parameters()]).clone()
    args2 = args_for('hard', 10, 2, 5e-4)
    hist2 = TRAIN(args2, be, train_ds, loader_of(val_ds), val_ds.img2txts, train_img2txt=train_ds.img2txts,
                  train_txt2img=train_ds.txt2img, mining_loaders=mining_loaders,
                  resume_from=os.path.join(args.output_dir, 'biencoder.last.pt'))
    assert [h['epoch'] for h in hist2] == [8, 9] and all(h['hard_negatives'] for h in hist2)
    assert all(np.isfinite(h['loss']) for h in hist2)
    assert train_ds.neg_imgs[0] is not None and len(train_ds.neg_imgs[0]) == 2 and train_ds.txt2img[train_ds.ids[0]] not in train_ds.neg_imgs[0]
    after = torch.cat([p.detach().reshape(-1) for p in be.parameters()])
    assert float((after - before).abs().max()) > 0


def test_text_tower_on_gpu_matches_reference_bert_encoder_golden(golden_dir):
    """a1 on the device: golden G9 (outputs of the reference's BertEncoder over transformers.BertModel) — strict load, cuda forward
    with the HIP [CLS] pooling branch (no_grad) and with autograd on."""
    import json
    import os
    from lightningdot_amd.towers import TowerConfig, TowerEncoder
    g = np.load(os.path.join(golden_dir, 'g9_text_tower_small.npz'))
    enc = TowerEncoder(TowerConfig(**json.loads(str(g['cfg']))), project_dim=int(g['project_dim']), with_image=False)
    enc.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd__')}, strict=True)
    enc = enc.cuda().eval()
    t = lambda k: torch.from_numpy(g[k]).cuda()
    valid = g['attention_mask'].astype(bool)
    with torch.no_grad():
        seq, pooled, _ = enc(t('input_ids'), t('attention_mask'), t('position_ids'))
    np.testing.assert_allclose(seq.cpu().numpy()[valid], g['seq'][valid], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(pooled.cpu().numpy(), g['pooled'], rtol=1e-4, atol=5e-5)
    _, pooled2, _ = enc(t('input_ids'), t('attention_mask'), t('position_ids'))
    np.testing.assert_allclose(pooled2.detach().cpu().numpy(), g['pooled'], rtol=1e-4, atol=5e-5)


def _write_small_configs(tmp_path, db, **extra):
    import json
    small = dict(vocab_size=28996, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                 max_position_embeddings=512, type_vocab_size=2, hidden_act='gelu', hidden_dropout_prob=0.0,
                 attention_probs_dropout_prob=0.0, initializer_range=0.02)
    (tmp_path / 'img_small.json').write_text(json.dumps(small))
    cfg = dict(txt_model_config='bert-base-cased', img_model_config=str(tmp_path / 'img_small.json'), itm_global_file=None,
               seed=42, output_dir=str(tmp_path / 'out'), max_txt_len=60, conf_th=0.2, max_bb=100, min_bb=10, num_bb=36,
               project_dim=64, val_txt_db=str(db / 'txt_db'), val_img_db=str(db / 'img_db'), test_txt_db=str(db / 'txt_db'),
               test_img_db=str(db / 'img_db'), project_name='itm-debug', n_workers=0, fp16=False, valid_batch_size=16)
    cfg.update(extra)
    (tmp_path / 'cfg.json').write_text(json.dumps(cfg))
    return str(tmp_path / 'cfg.json')


def _db_fixture(tmp_path, n_img, cpi):
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('make_db_fixture', os.path.join(root, 'tools', 'make_db_fixture.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    db = tmp_path / 'db'
    return db, mod.make(str(db), n_img=n_img, cpi=cpi, seed=5)


@pytest.mark.parametrize('compressed', [True, False])
def test_eval_model_from_on_disk_dbs(tmp_path, capsys, compressed):
    """f-2 -> hot path end to end (eval_itm.py:130-152): text DB (LZ4-framed msgpack records) + region-feature DB (npz / msgpack fp16) on
    disk -> readers -> ItmFastDataset -> itm_fast_collate -> towers on cuda -> HIP pooling / loss / index / search -> Recall@k.
    The recalls are checked against the oracle harness fed with the towers' own embeddings of the same batches."""
    import os
    from lightningdot_amd import data as D
    from lightningdot_amd.eval_itm import EVAL_MODEL, db_dataloader
    from lightningdot_amd.options import build_parser, parse_with_config
    from lightningdot_amd.towers import BiEncoder, save_checkpoint
    db, (examples, feats, nbb) = _db_fixture(tmp_path, 12, 3)
    if not compressed:                                        # the 'all' flavour is keyed by the thresholded name in real DB folders
        for ext in ('.bin', '.idx.json'):
            os.rename(db / 'img_db' / ('all' + ext), db / 'img_db' / ('feat_th0.2_max100_min10' + ext))
    cfg = _write_small_configs(tmp_path, db, **({'compressed_db': True} if compressed else {}))
    args = parse_with_config(build_parser(), ['--config', cfg])
    torch.manual_seed(1)
    be = BiEncoder(args, project_dim=64)
    ckdir = tmp_path / 'run_0.001_16_0_none_0.0_x'
    ckdir.mkdir()
    ck = save_checkpoint(be, None, None, 0, 0, str(ckdir / 'biencoder.last.pt'))
    res = EVAL_MODEL(cfg, ck)
    out = capsys.readouterr().out
    n_txt = len(examples)                                     # (max_txt_len = -1 for evaluation: nothing is filtered, trainer.py:206)
    assert set(res) == {'dev', 'test'} and f'indexed  {len(feats)} data' in out
    # the same batches through the same towers, scored by the oracle harness
    args.device = torch.device('cuda')
    args.inf_minibatch_size, args.vector_size = 400, 64
    loader, img2txt = db_dataloader(args, args.val_txt_db, args.val_img_db)
    be = be.cuda().eval()
    stream = []
    for b in loader:
        with torch.no_grad():
            q, c, _ = be(b)
        stream.append(dict(txt_index=b['txt_index'], img_fname=b['img_fname'], q=q.cpu().numpy(), ctx=c.cpu().numpy()))
    assert sum(len(s['txt_index']) for s in stream) == n_txt
    l2, a2, _, (o_txt, o_img), _ = O.eval_on_stream(stream, 64, img2txt, 100, 0.0)
    # (EVAL_MODEL keeps the reference's naming swap: its "recall_img" is text-query -> image retrieval, eval_itm.py:143,150)
    assert res['dev']['recall_img'] == o_txt and res['dev']['recall_txt'] == o_img
    assert res['dev']['loss'] == pytest.approx(l2, rel=1e-4, abs=1e-5) and res['dev']['accuracy'] == a2
    assert res['test'] == res['dev']                          # same DBs


def test_train_itm_cli_from_on_disk_dbs(tmp_path):
    """train_itm.main on on-disk DBs (no --synthetic): TxtTokDb / DetectFeatDb / ItmFastDataset feed one epoch of the fine-tuning
    loop with mined hard negatives, per-epoch evaluation and checkpoints (train_itm.py:176-358)."""
    import os
    from lightningdot_amd.train_itm import main
    db, (examples, feats, nbb) = _db_fixture(tmp_path, 16, 3)
    cfg = _write_small_configs(tmp_path, db, train_txt_dbs=[str(db / 'txt_db')], train_img_dbs=[str(db / 'img_db')],
                               train_batch_size=8, num_train_epochs=2, learning_rate=1e-3, num_hard_negatives=2,
                               sample_init_hard_negatives=True, hard_negatives_sampling='top', compressed_db=True)
    hist = main(['--config', cfg])
    assert [h['epoch'] for h in hist] == [0, 1] and all(h['hard_negatives'] for h in hist)
    assert all(np.isfinite(h['loss']) and set(h['recall']) == {1, 5, 10} for h in hist)
    assert os.path.exists(tmp_path / 'out' / 'biencoder.last.pt')
