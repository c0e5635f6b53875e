"""Ranking metrics on top of `LightFM.predict_rank` (API of the reference's
lightfm/evaluation.py, "EVAL": precision_at_k EVAL:14-87, recall_at_k EVAL:90-166, auc_score
EVAL:169-256, reciprocal_rank EVAL:259-327 -- same names, arguments, defaults and result shapes).

The O(users x items x d) part -- the rank of every test interaction among all items -- runs on
the GPU (predict_ranks as an MFMA sweep, calculate_auc_from_rank).  What is left for the host is
one segmented reduction per metric over the rank CSR's value array: `_per_user` below.
"""
import numpy as np

from ._lightfm_fast import CSRMatrix, calculate_auc_from_rank

__all__ = ["precision_at_k", "recall_at_k", "auc_score", "reciprocal_rank"]


class _Ranks(object):
    """The rank CSR of one evaluation: values per test interaction, row extents, which users count."""

    def __init__(self, model, test_interactions, train_interactions, user_features, item_features,
                 num_threads, check_intersections):
        if num_threads < 1:
            raise ValueError("Number of threads must be 1 or larger.")
        self.csr = model.predict_rank(test_interactions, train_interactions=train_interactions,
                                      user_features=user_features, item_features=item_features,
                                      num_threads=num_threads, check_intersections=check_intersections)
        self.values = self.csr.data
        self.indptr = self.csr.indptr
        self.per_row = np.diff(self.indptr)
        self.has_test = self.per_row > 0

    def per_user(self, values, ufunc, empty=0):
        """ufunc-reduction of `values` over every user's test interactions; `empty` for users without any.
        (reduceat over the starts of the non-empty rows: consecutive starts delimit exactly one row each.)"""
        out = np.full(len(self.per_row), empty, dtype=values.dtype)
        rows = np.flatnonzero(self.has_test)
        if len(rows):
            out[rows] = ufunc.reduceat(values, self.indptr[rows])
        return out

    def select(self, per_user, preserve_rows):
        return per_user if preserve_rows else per_user[self.has_test]


def precision_at_k(model, test_interactions, train_interactions=None, k=10, user_features=None,
                   item_features=None, preserve_rows=False, num_threads=1,
                   check_intersections=True):
    """Fraction of the top-k that are known positives, per user (EVAL:14-87)."""
    r = _Ranks(model, test_interactions, train_interactions, user_features, item_features, num_threads,
               check_intersections)
    hits = r.per_user((r.values < k).astype(r.values.dtype), np.add)
    return r.select(hits / k, preserve_rows)


def recall_at_k(model, test_interactions, train_interactions=None, k=10, user_features=None,
                item_features=None, preserve_rows=False, num_threads=1,
                check_intersections=True):
    """Positives in the top-k over all positives of the user (EVAL:90-166); users without test
    interactions give nan (0 / 0) when their rows are preserved, as in the reference."""
    r = _Ranks(model, test_interactions, train_interactions, user_features, item_features, num_threads,
               check_intersections)
    hits = r.select(r.per_user((r.values < k).astype(r.values.dtype), np.add), preserve_rows)
    relevant = r.select(np.asarray(test_interactions.getnnz(axis=1)).ravel(), preserve_rows)
    with np.errstate(divide="ignore", invalid="ignore"):
        return hits / relevant


def auc_score(model, test_interactions, train_interactions=None, user_features=None,
              item_features=None, preserve_rows=False, num_threads=1, check_intersections=True):
    """Probability that a random positive outranks a random negative (EVAL:169-256); the per-user
    reduction is the device kernel of PYX:1326-1376."""
    r = _Ranks(model, test_interactions, train_interactions, user_features, item_features, num_threads,
               check_intersections)
    assert np.all(r.values >= 0)
    n_users = test_interactions.shape[0]
    if train_interactions is None:
        train_positives = np.zeros(n_users, dtype=np.int32)
    else:
        train_positives = np.ascontiguousarray(np.asarray(train_interactions.getnnz(axis=1)).ravel(), dtype=np.int32)
    auc = np.zeros(n_users, dtype=np.float32)
    # the rank buffer is the CSR's own value array, sorted in place per user (EVAL:247-249)
    calculate_auc_from_rank(CSRMatrix(r.csr), train_positives, r.values, auc, num_threads)
    return r.select(auc, preserve_rows)


def reciprocal_rank(model, test_interactions, train_interactions=None, user_features=None,
                    item_features=None, preserve_rows=False, num_threads=1,
                    check_intersections=True):
    """1 / (rank of the best-ranked positive + 1), per user (EVAL:259-327)."""
    r = _Ranks(model, test_interactions, train_interactions, user_features, item_features, num_threads,
               check_intersections)
    best = r.per_user(1.0 / (r.values + 1.0), np.maximum)
    return r.select(best, preserve_rows)
