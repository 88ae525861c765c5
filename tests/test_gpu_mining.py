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


def _np_batches(batches):
    """the loader's batches in the oracle's form (towers factored out: 'q' text vectors, 'ctx' image vectors)"""
    return [dict(txt_index=b['txt_index'], img_fname=b['img_fname'], q=b['_q'].cpu().numpy(), ctx=b['_ctx'].cpu().numpy(),
                 n_txt=len(b['txt_index'])) for b in batches]


@pytest.mark.parametrize('nh', [3, 45])
def test_sampled_hard_negatives_against_the_oracle(nh):
    """dvl/hn.py:45-66 at the Flickr30k-1k shape (1 000 images x 5 captions, D = 768, eval batch 80), num_tops = 50 (nh 3) / 100 (nh 45),
    HIP route beside the CPU oracle (O.eval_on_stream = dvl/trainer.py:113-190 over O.FlatIP, O.hard_negative_postprocess = dvl/hn.py:57-63,
    both pinned by the reference's own outputs, goldens G3 / G4):
      * the populations the device route samples from (admissible_populations: the ids-only searches' label tensors minus the
        positives) equal the oracle's pre-sample populations as sets, for every key of both directions;
      * every id the device route draws comes from the oracle's population of its key;
      * with a deterministic sampler in place of random.sample the outputs of sampled_hard_negatives (host route, ranked lists) equal
        the oracle's."""
    import torch
    from oracle import oracle_np as O
    from lightningdot_amd import _lib
    _lib.require_gpu()
    from lightningdot_amd.harness import eval_model_on_dataloader
    from lightningdot_amd.hn import admissible_populations, num_hard_sampled, sampled_hard_negatives
    from lightningdot_amd.synthetic import s2_embeddings
    img, txt = s2_embeddings(1000, 768, 5, seed=21, device='cuda')
    # a first stage that is neither perfect nor useless: every third caption drifts far from its image (positives outside the top lists too)
    g = torch.Generator().manual_seed(3)
    drift = torch.randn(txt.shape, generator=g).cuda()
    txt = txt + torch.where((torch.arange(txt.shape[0], device='cuda') % 3 == 0)[:, None], 6.0 * drift, 0.2 * drift)
    batches, img2txt, txt2img = _loader(img, txt, 5, 80)
    n_top = num_hard_sampled(nh)
    assert n_top == (50 if nh == 3 else 100)
    args = types.SimpleNamespace(hnsw_index=False, vector_size=768, caption_score_weight=0.0, num_hard_negatives=nh)

    # ---- oracle ----------------------------------------------------------------------------------------------------------------------
    _, _, _, _, (o_rank_txt, o_rank_img) = O.eval_on_stream(_np_batches(batches), 768, img2txt, n_top)
    first = lambda pop, k: sorted(pop)[:k]
    o_pop_img, o_pop_txt, o_s_txt, o_s_img = O.hard_negative_postprocess(o_rank_txt, o_rank_img, txt2img, img2txt, nh, sample=first)
    n_missing = sum(txt2img[t] not in o_rank_txt[t] for t in o_rank_txt)
    assert 0 < n_missing < len(o_rank_txt)                  # (both branches of :57 are exercised)

    # ---- populations of the device route (ids-only searches) -------------------------------------------------------------------------
    _, _, (ix_img, ix_txt), recalls, (rank_txt, rank_img) = eval_model_on_dataloader(FakeEncoder(), batches, args, img2txt, n_top,
                                                                                     rank_sets_only=True)
    assert recalls == (None, None)
    (txt_ids, lab_t, ok_t), (img_ids, lab_i, ok_i) = admissible_populations(rank_txt, rank_img, img2txt, txt2img)
    assert txt_ids == list(o_pop_img) and img_ids == list(o_pop_txt)              # same keys, same (first-occurrence) order
    lab_t, ok_t, lab_i, ok_i = lab_t.cpu().numpy(), ok_t.cpu().numpy(), lab_i.cpu().numpy(), ok_i.cpu().numpy()
    img_names, txt_names = ix_img.index_id_to_db_id, ix_txt.index_id_to_db_id
    # Equal as sets, for every key.  The one admissible difference: the oracle's blocked sgemm and the HIP re-score sum 768 fp32 products in
    # different orders, so two rows whose exact scores tie to fp32 rounding at the BOUNDARY of a top-n_top list may be swapped (the near-tie
    # rule of test_gpu_configs.py): such a key must differ by exactly one id each way, the two ids' fp64 scores must agree to 2e-5 |q| max|x|,
    # and fewer than one key in 200 may be affected.
    img64, txt64 = img.double().cpu().numpy(), txt.double().cpu().numpy()
    vec = {f'i{j}': img64[j] for j in range(img64.shape[0])}
    vec.update({f't{j}': txt64[j] for j in range(txt64.shape[0])})
    xmax = {'i': float(np.linalg.norm(img64, axis=1).max()), 't': float(np.linalg.norm(txt64, axis=1).max())}
    near_tied = set()

    def same_population(key, got, want):
        if got == want:
            return
        extra, missing = got - want, want - got
        assert len(extra) == 1 and len(missing) == 1, (key, extra, missing)
        a, b = extra.pop(), missing.pop()
        assert abs(vec[key] @ vec[a] - vec[key] @ vec[b]) <= 2e-5 * np.linalg.norm(vec[key]) * xmax[a[0]], (key, a, b)
        near_tied.add(key)
    for r, t in enumerate(txt_ids):
        same_population(t, {img_names[l] for l in lab_t[r][ok_t[r]]}, set(o_pop_img[t]))
    for r, i in enumerate(img_ids):
        same_population(i, {txt_names[l] for l in lab_i[r][ok_i[r]]}, set(o_pop_txt[i]))
    assert len(near_tied) <= (len(txt_ids) + len(img_ids)) // 200, sorted(near_tied)
    st = ix_img.index.last_set_stats()
    assert 0 < st['rescored'] < st['candidates']                                  # (the searches did run in the ids-only mode)

    # ---- the device draw -------------------------------------------------------------------------------------------------------------
    gen = torch.Generator(device='cuda').manual_seed(5)
    hn_txt, hn_img = sampled_hard_negatives([batches], args, FakeEncoder(), img2txt, txt2img, generator=gen)
    assert list(hn_img) == txt_ids and list(hn_txt) == img_ids
    for t, negs in hn_img.items():
        assert len(negs) == nh and len(set(negs)) == nh and (set(negs) <= set(o_pop_img[t]) or t in near_tied)
    for i, negs in hn_txt.items():
        assert len(negs) == nh and len(set(negs)) == nh and (set(negs) <= o_pop_txt[i] or i in near_tied)

    # ---- deterministic sampler: outputs equal ------------------------------------------------------------------------------------------
    # (:58 goes through set(): the order random.sample sees is implementation-defined in the reference; the sampler sorts, like the oracle)
    got_txt, got_img = sampled_hard_negatives([batches], args, FakeEncoder(), img2txt, txt2img, sample=first)
    assert list(got_img) == list(o_s_img) and list(got_txt) == list(o_s_txt)
    assert all(got_img[t] == o_s_img[t] for t in got_img if t not in near_tied)
    assert all(got_txt[i] == o_s_txt[i] for i in got_txt if i not in near_tied)


@pytest.mark.parametrize('case', ['fused_k50', 'fused_k1000', 'dense_k100', 'narrow_k50', 'short_index', 'ties', 'shuffled', 'normalised'])
def test_ids_only_search_reports_the_set_of_the_full_rescore(case):
    """LDOT_OPT_RESULT_SET (consumers that drop the scores, dvl/hn.py:54-63): the labels of an ids-only search are, as a set, bit for bit
    those of the default search (every candidate re-scored) on every path; only a fraction of the candidates is gathered; and the premise
    of the shortcut — |exact - bf16 candidate score| <= E, the verify bound — holds on these data with a wide berth."""
    import torch
    from lightningdot_amd import _lib as L
    L.require_gpu()
    from lightningdot_amd.indexer import FlatIPIndex
    g = torch.Generator().manual_seed(sum(map(ord, case)))
    d = 768
    n, nq, k = {'fused_k50': (40000, 3000, 50), 'fused_k1000': (40000, 1500, 1000), 'dense_k100': (6000, 700, 100),
                'narrow_k50': (50000, 5, 50), 'short_index': (37, 300, 50), 'ties': (30000, 600, 50), 'shuffled': (40000, 1200, 100),
                'normalised': (40000, 1000, 100)}[case]
    x = torch.randn(n, d, generator=g)
    if case == 'ties':                     # every row stored three times: exact ties straddle the boundary of most sets
        x = x[: n // 3].repeat(3, 1)
    if case == 'normalised':               # unit rows and queries: scores in [-1, 1], crowded against the bf16 resolution
        x = torch.nn.functional.normalize(x, dim=1)
    q = x[torch.randint(0, x.shape[0], (nq,), generator=g)] + 0.7 * torch.randn(nq, d, generator=g) * (x.std().item())
    if case == 'normalised':
        q = torch.nn.functional.normalize(q, dim=1)
    x, q = x.cuda(), q.cuda()
    ix = FlatIPIndex(d)
    if case == 'shuffled':
        ix.set_option(L.OPT_ROW_SHUFFLE, 1)
    ix.add(x)
    s_full, l_full = ix.search_tensors(q, k)
    s_set, l_set = ix.search_tensors(q, k, ids_only=True)
    st = ix.last_set_stats()
    assert torch.equal(l_full.sort(dim=1).values, l_set.sort(dim=1).values)
    if case == 'short_index':
        assert st['rescored'] == 0 and bool((l_set[:, x.shape[0]:] == -1).all())     # fewer rows than k: nothing to decide
        return
    assert 0 < st['rescored'] < 0.7 * st['candidates'], st
    # the option is per call: the next default search re-scores everything again, in exact order
    s_again, l_again = ix.search_tensors(q, k)
    assert torch.equal(l_again, l_full) and torch.equal(s_again, s_full)
    # boundary entries carry exact scores, certain ones the bf16-input score: both within E of the exact score of their row
    exact = (q.double() @ x.double().T)
    got = torch.gather(exact, 1, l_set)
    E = 4.0 * 2.0 ** -8 * q.norm(dim=1) * x.norm(dim=1).max() / d ** 0.5
    err = (s_set.double() - got).abs()
    assert bool((err <= E[:, None]).all())
    # (measured: the largest error of a case is 0.7-0.85 E.min — the planted rows, q ~ x, whose products all have one sign; E is ~5 sigma there,
    # ~7 sigma for a candidate at the boundary of a set)


def test_reranker_candidate_export_against_the_oracle():
    """rerank.py:168-204,256-290: first-stage candidates at top-100 both ways + re-ranking of the first {10, 20, 50, 100} candidates
    with an external scorer (a fake score matrix).  Checker = the oracle's restatement of the reference's loops over the oracle's own
    exact index (O.rerank_first_stage / O.rerank_recall on O.DenseFlatIndexerOracle); checked = the device path (label tensors,
    gather + topk) AND the product's host loops."""
    import torch
    from oracle import oracle_np as O
    from lightningdot_amd.harness import eval_model_on_dataloader
    from lightningdot_amd.rerank import (first_stage_candidates, first_stage_rankings, rerank_recall, rerank_recall_device)
    from lightningdot_amd.synthetic import s2_embeddings
    img, txt = s2_embeddings(150, 64, 5, seed=13, device='cuda')
    txt = txt + 1.5 * torch.randn(txt.shape, generator=torch.Generator().manual_seed(1)).cuda()      # make the first stage imperfect
    batches, img2txt, txt2img = _loader(img, txt, 5, 128)
    args = types.SimpleNamespace(hnsw_index=False, vector_size=64, caption_score_weight=0.0)
    _, _, (ix_img, ix_txt), _, _ = eval_model_on_dataloader(FakeEncoder(), batches, args, img2txt, no_eval=True)   # rerank.py:149-150
    # ---- oracle: its own indexes over the same vectors, the reference's loops ---------------------------------------------------------
    nb = _np_batches(batches)
    _, _, (o_img, o_txt), _, _ = O.eval_on_stream(nb, 64, img2txt, no_eval=True)
    assert o_img.index_id_to_db_id == ix_img.index_id_to_db_id and o_txt.index_id_to_db_id == ix_txt.index_id_to_db_id
    want = O.rerank_first_stage(nb, o_img, o_txt, img2txt, txt2img)
    # ---- product -------------------------------------------------------------------------------------------------------------------------
    host = first_stage_rankings(FakeEncoder(), ix_img, ix_txt, batches, img2txt, txt2img)
    devr = first_stage_candidates(FakeEncoder(), ix_img, ix_txt, batches, img2txt, txt2img)
    assert host[0] == want[0] and host[1] == want[1] and host[2:] == want[2:]
    assert devr['labels_img'].shape == (750, 100) and devr['labels_txt'].shape == (750, 100)
    assert devr['recall_img'] == want[2] and devr['recall_txt'] == want[3] and devr['total_len'] == want[4]
    assert 0 < want[2][1] < want[2][100]                                   # (the first stage is neither perfect nor useless)
    lab = devr['labels_img'].cpu().tolist()
    assert all(want[0][t] == [ix_img.index_id_to_db_id[r] for r in lab[j]] for j, t in enumerate(devr['txt_ids']))
    lab = devr['labels_txt'].cpu().tolist()
    assert all(want[1][i] == [ix_txt.index_id_to_db_id[r] for r in lab[j]] for j, i in enumerate(devr['img_ids']))
    # external scorer: a fixed random matrix with a bonus on the true pairs (a cross-encoder that is better than the first stage)
    g = torch.Generator().manual_seed(7)
    n_txt, n_img = len(ix_txt.index_id_to_db_id), len(ix_img.index_id_to_db_id)
    mat = torch.randn(n_txt, n_img, generator=g)
    txt_row = {k: r for r, k in enumerate(ix_txt.index_id_to_db_id)}
    img_row = {k: r for r, k in enumerate(ix_img.index_id_to_db_id)}
    for t, i in txt2img.items():
        mat[txt_row[t], img_row[i]] += 8.0
    matn = mat.numpy()
    # image retrieval (rerank.py:256-270): query = text, db = images, denominator total_len
    want_ir = O.rerank_recall(want[0], devr['txt_ids'], lambda t, i: matn[txt_row[t], img_row[i]], lambda t, ids: txt2img[t] in ids, want[4])
    q_rows = torch.as_tensor([txt_row[t] for t in devr['txt_ids']])
    got_ir = rerank_recall_device(devr['labels_img'], mat[q_rows].cuda(), devr['pos_img'])
    host_ir = rerank_recall(host[0], lambda t, i: float(mat[txt_row[t], img_row[i]]), lambda t, ids: txt2img[t] in ids)
    assert got_ir == want_ir and host_ir == want_ir
    # a scorer that always ranks the true pair first turns the first stage's R@threshold into R@1 (rerank.py's point)
    assert all(got_ir[t][1] == want[2][t] / 750 for t in (10, 20, 50, 100)) and got_ir[100][1] > want[2][1] / 750
    # text retrieval (rerank.py:272-290): query = the DISTINCT image ids (the dict keeps an id's last occurrence), denominator len(img_ids)
    uniq = list(want[1])
    want_tr = O.rerank_recall(want[1], uniq, lambda i, t: matn[txt_row[t], img_row[i]], lambda i, ids: any(t in ids for t in img2txt[i]), len(uniq))
    assert devr['img_unique_ids'] == uniq
    q_rows = torch.as_tensor([img_row[i] for i in devr['img_ids']])
    got_tr = rerank_recall_device(devr['labels_txt'], mat.T[q_rows].contiguous().cuda(), devr['pos_txt'], rows=devr['img_unique_rows'])
    host_tr = rerank_recall(host[1], lambda i, t: float(mat[txt_row[t], img_row[i]]), lambda i, ids: any(t in ids for t in img2txt[i]),
                            denominator=len(uniq))
    assert got_tr == want_tr and host_tr == want_tr
