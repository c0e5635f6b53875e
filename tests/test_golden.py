"""Pins the oracle against golden fixtures produced by the reference itself
(tests/golden/make_golden.py).  Runs anywhere -- needs neither /root/reference
nor oracle/_ref."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H
from tests.test_oracle_vs_reference import LOSS_CASES, _problem, _run_orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {c[0]: c for c in LOSS_CASES}
FILES = sorted(glob.glob(os.path.join(GOLD, "*.npz")))


def test_fixtures_present():
    assert len(FILES) == len(LOSS_CASES) * 4


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_reproduces_reference_fixture(path):
    case_id, loss = os.path.basename(path)[:-4].split("__")
    gold = np.load(path)
    coo, item_f, user_f, st, rng, alpha = _problem(CASES[case_id])
    for _ in range(2):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _run_orc(loss, coo, item_f, user_f, st, shuffle, seeds, alpha)
    for n in oracle.ARRAYS:
        assert np.array_equal(getattr(st, n), gold[n]), n
    nu, ni = coo.shape
    uids = np.repeat(np.arange(nu, dtype=np.int32), ni)
    iids = np.tile(np.arange(ni, dtype=np.int32), nu)
    pred = oracle.predict(item_f, user_f, uids, iids, st)
    assert np.array_equal(pred[:: max(1, len(pred) // 512)], gold["predictions"])
    # ranks and AUC against the reference's predict_ranks / calculate_auc_from_rank
    import scipy.sparse as sp
    train, test = H.rank_problem(coo)
    ranks = np.zeros_like(test.data)
    oracle.predict_ranks(item_f, user_f, test, train, ranks, st)
    assert np.array_equal(ranks, gold["ranks"])
    rmat = sp.csr_matrix((ranks.copy(), test.indices, test.indptr), shape=test.shape)
    ntp = np.squeeze(np.array(train.getnnz(axis=1)).astype(np.int32))
    auc = np.zeros(nu, np.float32)
    oracle.auc_from_rank(rmat, ntp, rmat.data, auc)
    assert np.array_equal(auc, gold["auc"])
