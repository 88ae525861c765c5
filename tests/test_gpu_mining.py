"""GPU tests of the device-side result handling: hard-negative mining (dvl/hn.py:45-66) on the label tensors of the two mining
searches, get_indexer (dvl/trainer.py:93-110) and the lazy rank dicts of eval_model_on_dataloader."""
import time
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class FakeEncoder:
    """stands in for the two towers: the loader's batches carry the pooled embeddings"""

    def eval(self):
        return self

    def __call__(self, batch):
        return batch['_q'], batch['_ctx'], None


def _loader(img, txt, cpi, bs):
    import torch
    n = txt.shape[0]
    img_of = torch.arange(n, device=txt.device) // cpi
    batches = []
    for b0 in range(0, n, bs):
        b1 = min(b0 + bs, n)
        batches.append(dict(txt_index=[f't{j}' for j in range(b0, b1)], img_fname=[f'i{j // cpi}' for j in range(b0, b1)],
                            txts={'input_ids': torch.zeros(b1 - b0, 1, dtype=torch.long)},
                            _q=txt[b0:b1], _ctx=img[img_of[b0:b1]]))
    img2txt = {f'i{i}': [f't{i * cpi + c}' for c in range(cpi)] for i in range(img.shape[0])}
    txt2img = {f't{j}': f'i{j // cpi}' for j in range(n)}
    return batches, img2txt, txt2img


def test_device_hard_negatives_small():
    import torch
    from lightningdot_amd import _lib
    _lib.require_gpu()
    from lightningdot_amd.hn import num_hard_sampled, sampled_hard_negatives
    from lightningdot_amd.harness import eval_model_on_dataloader
    from lightningdot_amd.synthetic import s2_embeddings
    img, txt = s2_embeddings(60, 64, 5, seed=3, device='cuda')
    batches, img2txt, txt2img = _loader(img, txt, 5, 80)
    args = types.SimpleNamespace(hnsw_index=False, vector_size=64, caption_score_weight=0.0, num_hard_negatives=3)
    g = torch.Generator(device='cuda').manual_seed(5)
    hn_txt, hn_img = sampled_hard_negatives([batches], args, FakeEncoder(), img2txt, txt2img, generator=g)
    n_top = num_hard_sampled(3)
    _, _, _, _, (rank_txt, rank_img) = eval_model_on_dataloader(FakeEncoder(), batches, args, img2txt, n_top)
    assert set(hn_img) == set(txt2img) and set(hn_txt) == set(img2txt)
    for t, negs in hn_img.items():                       # text -> hard negative images
        assert len(negs) == 3 and len(set(negs)) == 3
        assert txt2img[t] not in negs and set(negs) <= set(rank_txt[t])
    for i, negs in hn_txt.items():                       # image -> hard negative captions
        assert len(negs) == 3 and len(set(negs)) == 3
        assert not (set(negs) & set(img2txt[i])) and set(negs) <= set(rank_img[i])
    # same seed -> same draw; the host sampler hook reproduces the reference's post-processing on the same searches
    g2 = torch.Generator(device='cuda').manual_seed(5)
    again = sampled_hard_negatives([batches], args, FakeEncoder(), img2txt, txt2img, generator=g2)
    assert again == (hn_txt, hn_img)
    from lightningdot_amd.hn import postprocess_hard_negatives
    first = lambda pop, k: sorted(pop)[:k]
    host = sampled_hard_negatives([batches], args, FakeEncoder(), img2txt, txt2img, sample=first)
    want = postprocess_hard_negatives(dict(rank_txt), dict(rank_img), img2txt, txt2img, 3, sample=first)
    assert host == want


def test_get_indexer_both_sides_and_hnsw_flag():
    """dvl/trainer.py:93-110: one index over the image (or text) side of a loader, flat or --hnsw_index surface; searched against
    the oracle's exact top-k on the same de-duplicated vectors."""
    import torch
    from oracle import oracle_np as O
    from lightningdot_amd.harness import get_indexer
    from lightningdot_amd.synthetic import s2_embeddings
    img, txt = s2_embeddings(40, 64, 5, seed=9, device='cuda')
    batches, img2txt, txt2img = _loader(img, txt, 5, 32)
    args = types.SimpleNamespace(vector_size=64)
    q = txt[:17]
    for hnsw in (False, True):
        ix_img = get_indexer(FakeEncoder(), batches, args, hnsw, img_retrieval=True)
        ix_txt = get_indexer(FakeEncoder(), batches, args, hnsw, img_retrieval=False)
        assert ix_img.index_id_to_db_id == [f'i{i}' for i in range(40)]          # de-duplicated, first-insertion order
        assert ix_txt.index_id_to_db_id == [f't{j}' for j in range(200)]
        for ix, rows, ids in ((ix_img, img, ix_img.index_id_to_db_id), (ix_txt, txt, ix_txt.index_id_to_db_id)):
            ref = O.FlatIP(64)
            ref.add(rows.cpu().numpy())
            es, el = ref.search(q.cpu().numpy(), 10)
            got = ix.search_knn(q.cpu().numpy(), 10)
            assert [g[0] for g in got] == [[ids[i] for i in row] for row in el]
            if not hnsw:
                np.testing.assert_allclose(np.stack([g[1] for g in got]), es, rtol=0, atol=1e-4)


def test_mining_at_flickr_train_scale():
    """29 000 images x 145 000 captions (the Flickr30k train set's size), nh = 3 -> top-50 both ways (145k x 29k and
    145k x 145k searches): the whole sampled_hard_negatives call, fake towers, in about two seconds of wall time."""
    import torch
    from lightningdot_amd.hn import sampled_hard_negatives
    from lightningdot_amd.synthetic import s2_embeddings
    img, txt = s2_embeddings(29000, 768, 5, seed=1, device='cuda')
    batches, img2txt, txt2img = _loader(img, txt, 5, 4096)
    args = types.SimpleNamespace(hnsw_index=False, vector_size=768, caption_score_weight=0.0, num_hard_negatives=3)
    g = torch.Generator(device='cuda').manual_seed(0)
    sampled_hard_negatives([batches[:4]], args, FakeEncoder(), img2txt, txt2img, generator=g)     # warm-up (allocations, code)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hn_txt, hn_img = sampled_hard_negatives([batches], args, FakeEncoder(), img2txt, txt2img, generator=g)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'mining 145k x 29k + 145k x 145k, nh=3: {dt:.2f} s')
    assert len(hn_img) == 145000 and len(hn_txt) == 29000
    for t in ('t0', 't77777', 't144999'):
        assert len(hn_img[t]) == 3 and txt2img[t] not in hn_img[t]
    for i in ('i0', 'i28999'):
        assert len(hn_txt[i]) == 3 and not (set(hn_txt[i]) & set(img2txt[i]))
    assert dt < 4.0, dt


def test_harness_and_mining_with_hnsw_index_flag():
    """args.hnsw_index=True (dvl/trainer.py:122-127): the harness and the mining go through DenseHNSWFlatIndexer.search_knn_tensors;
    exact-backed, so rank lists and recalls equal the flat run's."""
    import torch
    from lightningdot_amd.harness import eval_model_on_dataloader
    from lightningdot_amd.hn import sampled_hard_negatives
    from lightningdot_amd.indexer import DenseHNSWFlatIndexer
    from lightningdot_amd.synthetic import s2_embeddings
    img, txt = s2_embeddings(50, 64, 5, seed=11, device='cuda')
    batches, img2txt, txt2img = _loader(img, txt, 5, 64)
    out = {}
    for hnsw in (False, True):
        args = types.SimpleNamespace(hnsw_index=hnsw, vector_size=64, caption_score_weight=0.0, num_hard_negatives=2)
        loss, ratio, (ix_img, ix_txt), recalls, (rank_txt, rank_img) = eval_model_on_dataloader(FakeEncoder(), batches, args, img2txt, 20)
        out[hnsw] = (loss, ratio, recalls, {k: rank_txt[k] for k in rank_txt}, {k: rank_img[k] for k in rank_img})
        if hnsw:
            assert isinstance(ix_img, DenseHNSWFlatIndexer) and isinstance(ix_txt, DenseHNSWFlatIndexer)
            d, lab = ix_img.search_knn_tensors(txt[:9], 7)                  # squared L2 in the augmented space, ascending
            assert d.is_cuda and lab.is_cuda and bool((d[:, 1:] >= d[:, :-1]).all())
            host = ix_img.search_knn(txt[:9].cpu().numpy(), 7)
            np.testing.assert_allclose(np.stack([h[1] for h in host]), d.cpu().numpy(), rtol=1e-5, atol=1e-3)
            assert [h[0] for h in host] == [[ix_img.index_id_to_db_id[i] for i in row] for row in lab.cpu().tolist()]
        g = torch.Generator(device='cuda').manual_seed(2)
        hn_txt, hn_img = sampled_hard_negatives([batches], args, FakeEncoder(), img2txt, txt2img, generator=g)
        assert set(hn_img) == set(txt2img) and all(len(v) == 2 for v in hn_txt.values())
    assert out[False] == out[True]


def test_rank_dicts_hold_the_last_occurrence_of_duplicated_query_ids():
    """dvl/trainer.py:168,171: {id: result} comprehensions keep the result of an id's LAST occurrence; the harness searches only that
    occurrence.  Image vectors that differ between occurrences of the same id make the rule observable."""
    import torch
    from oracle import oracle_np as O
    from lightningdot_amd.harness import eval_model_on_dataloader
    from lightningdot_amd.synthetic import s2_embeddings
    img, txt = s2_embeddings(30, 64, 5, seed=4, device='cuda')
    batches, img2txt, txt2img = _loader(img, txt, 5, 40)
    g = torch.Generator(device='cpu').manual_seed(0)
    for b in batches:                                              # every occurrence of an image gets its own perturbation
        b['_ctx'] = b['_ctx'] + 0.3 * torch.randn(b['_ctx'].shape, generator=g).cuda()
    args = types.SimpleNamespace(hnsw_index=False, vector_size=64, caption_score_weight=0.0)
    _, _, (ix_img, ix_txt), _, (rank_txt, rank_img) = eval_model_on_dataloader(FakeEncoder(), batches, args, img2txt, 10)
    allq = torch.cat([b['_ctx'] for b in batches]).cpu().numpy()
    ids = sum([b['img_fname'] for b in batches], [])
    last = {k: i for i, k in enumerate(ids)}
    ref = O.FlatIP(64)
    ref.add(ix_txt.index.get_rows(0, ix_txt.index.ntotal))
    _, el = ref.search(allq, 10)                                   # the reference searches EVERY occurrence ...
    want = {k: [ix_txt.index_id_to_db_id[j] for j in el[i]] for k, i in last.items()}     # ... and keeps the last one
    assert list(rank_img) == list(dict.fromkeys(ids))
    assert {k: rank_img[k] for k in rank_img} == want


def test_reranker_candidate_export_on_device_equals_host_loops():
    """rerank.py:168-204,256-290: first-stage candidates at top-100 both ways + re-ranking of the first {10, 20, 50, 100} candidates
    with an external scorer (a fake score matrix): the device path (label tensors, gather + topk) gives the recalls of the host
    loops that mirror the reference."""
    import torch
    from lightningdot_amd.harness import eval_model_on_dataloader
    from lightningdot_amd.rerank import (RECALL_TOPS, first_stage_candidates, first_stage_rankings, rerank_recall,
                                         rerank_recall_device)
    from lightningdot_amd.synthetic import s2_embeddings
    img, txt = s2_embeddings(150, 64, 5, seed=13, device='cuda')
    txt = txt + 1.5 * torch.randn(txt.shape, generator=torch.Generator().manual_seed(1)).cuda()      # make the first stage imperfect
    batches, img2txt, txt2img = _loader(img, txt, 5, 128)
    args = types.SimpleNamespace(hnsw_index=False, vector_size=64, caption_score_weight=0.0)
    _, _, (ix_img, ix_txt), _, _ = eval_model_on_dataloader(FakeEncoder(), batches, args, img2txt, no_eval=True)   # rerank.py:149-150
    host = first_stage_rankings(FakeEncoder(), ix_img, ix_txt, batches, img2txt, txt2img)
    devr = first_stage_candidates(FakeEncoder(), ix_img, ix_txt, batches, img2txt, txt2img)
    assert devr['labels_img'].shape == (750, 100) and devr['labels_txt'].shape == (750, 100)
    assert devr['recall_img'] == host[2] and devr['recall_txt'] == host[3] and devr['total_len'] == host[4]
    assert 0 < host[2][1] < host[2][100]                                   # (the first stage is neither perfect nor useless)
    # the exported candidate lists are the host path's lists
    lab = devr['labels_img'].cpu().tolist()
    assert all(host[0][t] == [ix_img.index_id_to_db_id[r] for r in lab[j]] for j, t in enumerate(devr['txt_ids']))
    # external scorer: a fixed random matrix with a bonus on the true pairs (a cross-encoder that is better than the first stage)
    g = torch.Generator().manual_seed(7)
    n_txt, n_img = len(ix_txt.index_id_to_db_id), len(ix_img.index_id_to_db_id)
    mat = torch.randn(n_txt, n_img, generator=g)
    txt_row = {k: r for r, k in enumerate(ix_txt.index_id_to_db_id)}
    img_row = {k: r for r, k in enumerate(ix_img.index_id_to_db_id)}
    for t, i in txt2img.items():
        mat[txt_row[t], img_row[i]] += 8.0
    # image retrieval: query = text, db = images
    q_rows = torch.as_tensor([txt_row[t] for t in devr['txt_ids']])
    got_ir = rerank_recall_device(devr['labels_img'], mat[q_rows].cuda(), devr['pos_img'])
    want_ir = rerank_recall(host[0], lambda t, i: float(mat[txt_row[t], img_row[i]]), lambda t, ids: txt2img[t] in ids)
    assert got_ir == want_ir
    # a scorer that always ranks the true pair first turns the first stage's R@threshold into R@1 (rerank.py's point)
    assert all(got_ir[t][1] == host[2][t] / 750 for t in (10, 20, 50, 100)) and got_ir[100][1] > host[2][1] / 750
    # text retrieval: query = image (every occurrence, like the reference's loop), db = texts
    q_rows = torch.as_tensor([img_row[i] for i in devr['img_ids']])
    got_tr = rerank_recall_device(devr['labels_txt'], mat.T[q_rows].contiguous().cuda(), devr['pos_txt'], denominator=150)
    want_tr = rerank_recall(host[1], lambda i, t: float(mat[txt_row[t], img_row[i]]), lambda i, ids: any(t in ids for t in img2txt[i]),
                            denominator=150)
    # (host[1] holds one list per distinct image = the last occurrence; all occurrences of an image carry the same vector here)
    uniq_last = {i: j for j, i in enumerate(devr['img_ids'])}
    sel = torch.as_tensor(list(uniq_last.values()))
    got_tr_u = rerank_recall_device(devr['labels_txt'][sel.cuda()], mat.T[q_rows[sel]].contiguous().cuda(), devr['pos_txt'][sel.cuda()],
                                    denominator=150)
    assert got_tr_u == want_tr and all(abs(got_tr[t][k] - 5 * want_tr[t][k]) < 1e-9 for t in got_tr for k in (1, 5, 10))
    # the selection + denominator of the reference's loop (distinct image ids, last occurrence) come with the candidates
    assert devr['img_unique_ids'] == list(uniq_last.keys()) and devr['img_unique_rows'].cpu().tolist() == list(uniq_last.values())
    got_tr_r = rerank_recall_device(devr['labels_txt'], mat.T[q_rows].contiguous().cuda(), devr['pos_txt'], rows=devr['img_unique_rows'])
    assert got_tr_r == want_tr
