"""Shared helpers for the parity tests (test infrastructure; may import the oracle)."""
import numpy as np

from oracle import oracle_np as O


def assert_topk_matches(q, x, scores, labels, k, atol=1e-3, eps=None, check_labels_exact=True):
    """Compare a (scores, labels) top-k result with the exact answer computed in fp64 on the CPU.

    * reported scores must equal the true inner products of the reported rows within `atol`
    * rows must be sorted by descending score
    * the reported set must be a valid top-k: every reported row's true score >= (k-th best true score) - eps
    * rank-1 must be exact unless the true top-2 gap is below eps
    * with check_labels_exact, labels must equal the oracle ordering wherever neighbouring true scores differ by
      more than eps (near-ties may legitimately swap under fp32 summation-order differences)
    """
    q64, x64 = np.asarray(q, np.float64), np.asarray(x, np.float64)
    full = q64 @ x64.T
    n = x64.shape[0]
    kk = min(k, n)
    scale = max(1.0, float(np.abs(full).max()) if full.size else 1.0)
    if eps is None:
        eps = 4e-6 * scale * 10
    order = np.argsort(-full, axis=1, kind='stable')[:, :kk]
    top = np.take_along_axis(full, order, axis=1)
    assert scores.shape == (q64.shape[0], k) and labels.shape == (q64.shape[0], k)
    if kk < k:
        assert (labels[:, kk:] == -1).all()
        assert (scores[:, kk:] == np.float32(O.NEG_FLT_MAX)).all()
    lab = labels[:, :kk]
    assert (lab >= 0).all() and (lab < n).all()
    # no duplicates
    srt = np.sort(lab, axis=1)
    assert (np.diff(srt, axis=1) > 0).all(), 'duplicate labels in a result row'
    true_of_mine = np.take_along_axis(full, lab, axis=1)
    np.testing.assert_allclose(scores[:, :kk], true_of_mine, rtol=0, atol=atol)
    assert (np.diff(scores[:, :kk].astype(np.float64), axis=1) <= 1e-30).all(), 'scores not sorted descending'
    kth = top[:, -1:]
    assert (true_of_mine >= kth - eps).all(), 'a reported row is not in the true top-k'
    gap12 = top[:, 0] - top[:, 1] if kk > 1 else np.full(q64.shape[0], np.inf)
    clear = gap12 > eps
    assert (lab[clear, 0] == order[clear, 0]).all(), 'rank-1 mismatch'
    if check_labels_exact:
        mism = lab != order
        if mism.any():
            # allowed only inside near-tie runs
            d = np.abs(np.take_along_axis(full, lab, axis=1) - top)
            assert (d[mism] <= eps).all(), f'label mismatch outside near-ties: max diff {d[mism].max()}'
    return full


def planted_queries(x, nq, seed=4321, noise=0.5):
    """q_i = X[g_i] + noise * eps_i with g_i = (i * 9973) mod N."""
    n, d = x.shape
    rng = np.random.default_rng(seed)
    g = (np.arange(nq, dtype=np.int64) * 9973) % n
    q = x[g] + noise * rng.standard_normal((nq, d)).astype(np.float32)
    return q.astype(np.float32), g
