"""Deterministic shape-matched synthetic interaction data.

No MovieLens file is reachable from the build container or the GPU box (no
network), so every "ML-*" configuration runs on data with the published SHAPE of
the dataset: Zipf item popularity, log-normal user activity, and a latent
cluster structure so that ranking metrics (precision@k) are learnable and
comparable between backends.
"""
import numpy as np
import scipy.sparse as sp

SHAPES = {
    # name: (n_users, n_items, n_interactions)
    "ml-100k": (943, 1682, 100000),
    "ml-20m": (138493, 26744, 20000263),
}


def make_interactions(n_users, n_items, nnz, seed=42, n_clusters=64, zipf=0.9, affinity=2.0,
                      min_per_user=1):
    """COO float32 [n_users, n_items] with ~nnz unique (user, item) pairs, all 1.0."""
    rng = np.random.RandomState(seed)
    pop = 1.0 / np.arange(1, n_items + 1) ** zipf
    pop = pop[rng.permutation(n_items)]
    zi = rng.randn(n_items, 8)
    zc = rng.randn(n_clusters, 8)
    user_cluster = rng.randint(0, n_clusters, size=n_users)
    act = rng.lognormal(mean=0.0, sigma=1.0, size=n_users)
    target = np.maximum(min_per_user, np.round(act / act.sum() * nnz * 1.08)).astype(np.int64)
    target = np.minimum(target, n_items // 2)
    probs = []
    for c in range(n_clusters):
        logits = affinity * (zi @ zc[c]) / np.sqrt(8.0)
        p = pop * np.exp(logits - logits.max())
        probs.append(p / p.sum())
    members = [np.where(user_cluster == c)[0] for c in range(n_clusters)]
    key = np.empty(0, np.int64)
    want = target.astype(np.float64)
    for _ in range(12):  # sampling with replacement collides on popular items: top up
        have = np.bincount(key // n_items, minlength=n_users) if len(key) else np.zeros(n_users)
        deficit = np.maximum(0.0, want - have)
        if deficit.sum() < 1 or len(key) >= nnz:
            break
        draws = np.ceil(deficit * 1.3).astype(np.int64)
        us, its = [], []
        for c in range(n_clusters):
            if len(members[c]) == 0:
                continue
            counts = draws[members[c]]
            total = int(counts.sum())
            if total == 0:
                continue
            its.append(rng.choice(n_items, size=total, p=probs[c]))
            us.append(np.repeat(members[c], counts))
        u = np.concatenate(us).astype(np.int64)
        i = np.concatenate(its).astype(np.int64)
        key = np.unique(np.concatenate([key, u * n_items + i]))
    if len(key) > nnz:
        key = key[np.sort(rng.choice(len(key), size=nnz, replace=False))]
    rng.shuffle(key)
    u = (key // n_items).astype(np.int32)
    i = (key % n_items).astype(np.int32)
    return sp.coo_matrix((np.ones(len(u), np.float32), (u, i)), shape=(n_users, n_items),
                         dtype=np.float32)


def big_interactions(n_users, n_items, nnz, seed=4, zipf=0.9, sigma=1.0, chunk=8_000_000):
    """Fast generator for the large synthetic shapes (BASELINE configs C4 / C5): ~nnz unique
    (user, item) pairs, log-normal user activity, Zipf(zipf) item popularity drawn by inverse CDF
    (continuous density ~ (rank + 1)^-zipf, ranks scattered over the id space by a multiplicative
    permutation), no latent structure.  Users come out sorted (a COO in CSR order); duplicates
    within a user are removed chunk by chunk so the host never sorts the whole list at once."""
    rng = np.random.RandomState(seed)
    act = rng.lognormal(mean=0.0, sigma=sigma, size=n_users)
    counts = np.maximum(1, np.round(act / act.sum() * nnz * 1.03)).astype(np.int64)
    counts = np.minimum(counts, max(1, n_items // 4))
    a = 1.0 - zipf
    top = (n_items + 1.0) ** a - 1.0
    mult = 2654435761 % n_items
    while np.gcd(mult, n_items) != 1:
        mult += 1
    rows, cols = [], []
    bounds = np.concatenate([[0], np.cumsum(counts)])
    u0 = 0
    while u0 < n_users:
        u1 = int(np.searchsorted(bounds, bounds[u0] + chunk, side="right"))
        u1 = min(max(u1 - 1, u0 + 1), n_users)
        c = counts[u0:u1]
        users = np.repeat(np.arange(u0, u1, dtype=np.int64), c)
        rank = np.floor((rng.rand(len(users)) * top + 1.0) ** (1.0 / a) - 1.0).astype(np.int64)
        np.clip(rank, 0, n_items - 1, out=rank)
        items = (rank * mult) % n_items
        key = np.unique(users * n_items + items)
        rows.append((key // n_items).astype(np.int32))
        cols.append((key % n_items).astype(np.int32))
        u0 = u1
    u = np.concatenate(rows)
    i = np.concatenate(cols)
    return sp.coo_matrix((np.ones(len(u), np.float32), (u, i)), shape=(n_users, n_items), dtype=np.float32)


def hashed_item_features(n_items, n_cols=1_000_000, mean_nnz=8, seed=5):
    """Item feature CSR of config C5: one row per item over `n_cols` feature embeddings,
    max(1, Poisson(mean_nnz)) random columns per row, values 1/nnz (no identity block)."""
    rng = np.random.RandomState(seed)
    counts = np.maximum(1, rng.poisson(mean_nnz, size=n_items)).astype(np.int64)
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    indices = rng.randint(0, n_cols, size=int(indptr[-1])).astype(np.int32)
    data = np.repeat((1.0 / counts).astype(np.float32), counts)
    return sp.csr_matrix((data, indices, indptr), shape=(n_items, n_cols), dtype=np.float32)


def split_off_test(coo, n_train, seed=0):
    """(train with exactly n_train interactions, test = the rest), disjoint by construction."""
    rng = np.random.RandomState(seed)
    perm = rng.permutation(coo.nnz)
    tr, te = np.sort(perm[:n_train]), np.sort(perm[n_train:])

    def sub(idx):
        return sp.coo_matrix((coo.data[idx], (coo.row[idx], coo.col[idx])), shape=coo.shape, dtype=np.float32)
    return sub(tr), sub(te)


def named(name, seed=42, scale=1.0):
    """Synthetic data with the shape of a named dataset; scale < 1 keeps the tables
    full-size and sub-samples the interactions."""
    n_users, n_items, nnz = SHAPES[name]
    return make_interactions(n_users, n_items, int(nnz * scale), seed=seed)


def train_test_split(coo, test_fraction=0.1, seed=0):
    """Random split of the interactions (disjoint by construction)."""
    rng = np.random.RandomState(seed)
    n = len(coo.data)
    mask = rng.rand(n) < test_fraction

    def sub(m):
        return sp.coo_matrix((coo.data[m], (coo.row[m], coo.col[m])), shape=coo.shape,
                             dtype=np.float32)
    return sub(~mask), sub(mask)


def tag_item_features(n_items, n_tags=1128, per_item=8, seed=7, normalise=False):
    """[identity | tags] item feature CSR (config C3's "item tag/genre CSR")."""
    rng = np.random.RandomState(seed)
    rows = np.repeat(np.arange(n_items), per_item)
    cols = rng.randint(0, n_tags, size=n_items * per_item)
    tags = sp.coo_matrix((np.ones(len(rows), np.float32), (rows, cols)),
                         shape=(n_items, n_tags)).tocsr()
    tags.data[:] = 1.0
    m = sp.hstack([sp.identity(n_items, dtype=np.float32, format="csr"), tags]).tocsr()
    m = m.astype(np.float32)
    if normalise:
        rs = np.asarray(m.sum(axis=1)).ravel()
        m = sp.diags((1.0 / rs).astype(np.float32)).dot(m).tocsr().astype(np.float32)
    m.sort_indices()
    return m
