"""Ranking metrics on top of `LightFM.predict_rank` (API of the reference's
lightfm/evaluation.py, "EVAL"): precision_at_k, recall_at_k, auc_score,
reciprocal_rank.  The O(users x items x d) rank computation runs on the GPU
(predict_ranks kernel); the reductions here are a few numpy lines.
"""
import numpy as np

from ._lightfm_fast import CSRMatrix, calculate_auc_from_rank

__all__ = ["precision_at_k", "recall_at_k", "auc_score", "reciprocal_rank"]


def _ranks(model, test_interactions, train_interactions, user_features, item_features,
           num_threads, check_intersections):
    return model.predict_rank(test_interactions, train_interactions=train_interactions,
                              user_features=user_features, item_features=item_features,
                              num_threads=num_threads, check_intersections=check_intersections)


def precision_at_k(model, test_interactions, train_interactions=None, k=10, user_features=None,
                   item_features=None, preserve_rows=False, num_threads=1,
                   check_intersections=True):
    """Fraction of the top-k that are known positives, per user (EVAL:14-87)."""
    if num_threads < 1:
        raise ValueError("Number of threads must be 1 or larger.")
    ranks = _ranks(model, test_interactions, train_interactions, user_features, item_features,
                   num_threads, check_intersections)
    ranks.data = np.less(ranks.data, k, ranks.data)
    precision = np.squeeze(np.array(ranks.sum(axis=1))) / k
    if not preserve_rows:
        precision = precision[test_interactions.getnnz(axis=1) > 0]
    return precision


def recall_at_k(model, test_interactions, train_interactions=None, k=10, user_features=None,
                item_features=None, preserve_rows=False, num_threads=1,
                check_intersections=True):
    """Positives in the top-k over all positives of the user (EVAL:90-166)."""
    if num_threads < 1:
        raise ValueError("Number of threads must be 1 or larger.")
    ranks = _ranks(model, test_interactions, train_interactions, user_features, item_features,
                   num_threads, check_intersections)
    ranks.data = np.less(ranks.data, k, ranks.data)
    retrieved = np.squeeze(test_interactions.getnnz(axis=1))
    hit = np.squeeze(np.array(ranks.sum(axis=1)))
    if not preserve_rows:
        hit = hit[test_interactions.getnnz(axis=1) > 0]
        retrieved = retrieved[test_interactions.getnnz(axis=1) > 0]
    return hit / retrieved


def auc_score(model, test_interactions, train_interactions=None, user_features=None,
              item_features=None, preserve_rows=False, num_threads=1, check_intersections=True):
    """Probability that a random positive outranks a random negative (EVAL:169-256)."""
    if num_threads < 1:
        raise ValueError("Number of threads must be 1 or larger.")
    ranks = _ranks(model, test_interactions, train_interactions, user_features, item_features,
                   num_threads, check_intersections)
    assert np.all(ranks.data >= 0)
    auc = np.zeros(ranks.shape[0], dtype=np.float32)
    if train_interactions is not None:
        num_train_positives = np.squeeze(
            np.array(train_interactions.getnnz(axis=1)).astype(np.int32))
    else:
        num_train_positives = np.zeros(test_interactions.shape[0], dtype=np.int32)
    num_train_positives = np.ascontiguousarray(np.atleast_1d(num_train_positives), dtype=np.int32)
    # the reference passes ranks.data as the rank buffer (EVAL:247-249)
    calculate_auc_from_rank(CSRMatrix(ranks), num_train_positives, ranks.data, auc, num_threads)
    if not preserve_rows:
        auc = auc[test_interactions.getnnz(axis=1) > 0]
    return auc


def reciprocal_rank(model, test_interactions, train_interactions=None, user_features=None,
                    item_features=None, preserve_rows=False, num_threads=1,
                    check_intersections=True):
    """1 / (rank of the best-ranked positive + 1), per user (EVAL:259-327)."""
    if num_threads < 1:
        raise ValueError("Number of threads must be 1 or larger.")
    ranks = _ranks(model, test_interactions, train_interactions, user_features, item_features,
                   num_threads, check_intersections)
    ranks.data = 1.0 / (ranks.data + 1.0)
    ranks = np.squeeze(np.array(ranks.max(axis=1).todense()))
    if not preserve_rows:
        ranks = ranks[test_interactions.getnnz(axis=1) > 0]
    return ranks
