"""Shared problem generators for the parity tests (seeded, small)."""
import numpy as np
import scipy.sparse as sp


def make_interactions(n_users, n_items, nnz, seed, ratings=False, zipf=0.9):
    """Random COO interactions [n_users, n_items], duplicates removed, float32."""
    rng = np.random.RandomState(seed)
    p = 1.0 / np.arange(1, n_items + 1) ** zipf
    p /= p.sum()
    u = rng.randint(0, n_users, size=nnz)
    i = rng.choice(n_items, size=nnz, p=p)
    key = np.unique(u.astype(np.int64) * n_items + i)
    rng.shuffle(key)
    u, i = (key // n_items).astype(np.int32), (key % n_items).astype(np.int32)
    if ratings:
        data = rng.choice([-1.0, 0.0, 1.0, 2.0, 5.0], size=len(u)).astype(np.float32)
    else:
        data = np.ones(len(u), np.float32)
    return sp.coo_matrix((data, (u, i)), shape=(n_users, n_items), dtype=np.float32)


def identity_features(n):
    return sp.identity(n, dtype=np.float32, format="csr")


def tag_features(n_rows, n_tags, per_row, seed, with_identity=True, normalise=False):
    """[identity | random tags] CSR like a Dataset-built feature matrix."""
    rng = np.random.RandomState(seed)
    rows = np.repeat(np.arange(n_rows), per_row)
    cols = rng.randint(0, n_tags, size=n_rows * per_row)
    vals = (rng.rand(n_rows * per_row) + 0.5).astype(np.float32)
    tags = sp.coo_matrix((vals, (rows, cols)), shape=(n_rows, n_tags), dtype=np.float32).tocsr()
    tags.sum_duplicates()
    m = sp.hstack([identity_features(n_rows), tags]).tocsr() if with_identity else tags
    m = m.astype(np.float32)
    if normalise:
        rs = np.asarray(m.sum(axis=1)).ravel()
        rs[rs == 0] = 1
        m = sp.diags((1.0 / rs).astype(np.float32)).dot(m).tocsr().astype(np.float32)
    m.sort_indices()
    return m


def positives_csr(coo):
    """lightfm.py:365-372 -- CSR with sorted indices (duplicates summed)."""
    m = coo.tocsr()
    if not m.has_sorted_indices:
        m = m.sorted_indices()
    return m


def rank_problem(coo, seed=99, nnz=300):
    """(train CSR, disjoint test CSR) for predict_ranks / AUC fixtures."""
    nu, ni = coo.shape
    train = positives_csr(coo).astype(np.float32)
    test = make_interactions(nu, ni, nnz, seed=seed).tocsr().astype(np.float32)
    test = (test - test.multiply(train.astype(bool))).tocsr().astype(np.float32)
    test.eliminate_zeros()
    test.sort_indices()
    return train, test


def epoch_inputs(coo, rng, num_threads=1):
    """Host RNG order of lightfm.py:689-690 + _lightfm_fast.pyx:812-814."""
    shuffle = np.arange(len(coo.data), dtype=np.int32)
    rng.shuffle(shuffle)
    seeds = rng.randint(0, np.iinfo(np.int32).max, size=num_threads).astype(np.uint32)
    return shuffle, seeds


class FixedRandom:
    """Stands in for numpy RandomState in calls to the compiled reference:
    returns pre-drawn seeds from .randint so both sides see the same values."""

    def __init__(self, seeds):
        self.seeds = np.asarray(seeds)

    def randint(self, lo, hi, size=None):
        assert size == len(self.seeds)
        return self.seeds.astype(np.int64)


def assert_states_equal(a, b, exact=True, rtol=1e-4, atol=1e-7):
    from oracle.oracle import ARRAYS
    for n in ARRAYS:
        x, y = getattr(a, n), getattr(b, n)
        if exact:
            assert np.array_equal(x, y), "%s differs: max abs %g" % (n, np.abs(x - y).max())
        else:
            np.testing.assert_allclose(x, y, rtol=rtol, atol=atol, err_msg=n)


def assert_states_within_ulps(a, b, ulps=1, min_exact=0.5):
    """The bar for updates published as old + float32(new - old) (global_atomic_add_f32): the sum
    reproduces `new` wherever the subtraction is exact (Sterbenz: new / 2 <= old <= 2 new) and is
    within one float32 ulp of the larger of the two elsewhere -- an ABSOLUTE error that later updates
    of the cell carry along even when its value shrinks.  So: every cell within `ulps` ulps of the
    array's largest magnitude, and at least `min_exact` of the cells of every array bit-identical."""
    from oracle.oracle import ARRAYS
    for n in ARRAYS:
        x, y = getattr(a, n), getattr(b, n)
        assert x.shape == y.shape, n
        if not x.size:
            continue
        top = np.float32(max(np.abs(x).max(), np.abs(y).max()))
        tol = ulps * float(np.spacing(top))
        err = np.abs(x.astype(np.float64) - y.astype(np.float64))
        print("within_ulps %-28s max err %.2f ulp of %g, bit-identical %.4f" % (n, err.max() / max(float(np.spacing(top)), 1e-45), top, np.mean(x == y)))
        assert err.max() <= tol, "%s: %d cells beyond %d ulp of %g, max abs %g" % (
            n, int((err > tol).sum()), ulps, top, err.max())
        assert np.mean(x == y) >= min_exact, "%s: only %.3f of the cells bit-identical" % (n, np.mean(x == y))
