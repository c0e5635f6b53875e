"""Generates tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref/strict,
compiled from /root/reference by `make -C oracle ref`).  Run in the build
container only:  python tests/golden/make_golden.py

Each fixture stores the seeds/case id needed to regenerate the inputs with
tests/helpers.py plus the reference's outputs after 2 epochs (12 arrays).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.test_oracle_vs_reference import LOSS_CASES, _problem, _run_ref  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    oracle.build()
    ref = oracle.ref_module("strict")
    for case in LOSS_CASES:
        for loss in ("warp", "bpr", "logistic", "warp-kos"):
            coo, item_f, user_f, st, rng, alpha = _problem(case)
            for _ in range(2):
                shuffle, seeds = H.epoch_inputs(coo, rng)
                _run_ref(ref, loss, coo, item_f, user_f, st, shuffle, seeds, alpha)
            # predictions over all pairs from the reference's predict_lightfm
            nu, ni = coo.shape
            uids = np.repeat(np.arange(nu, dtype=np.int32), ni)
            iids = np.tile(np.arange(ni, dtype=np.int32), nu)
            pred = np.empty(len(uids), np.float32)
            C = ref.CSRMatrix
            ref.predict_lightfm(C(item_f), C(user_f), uids, iids, pred, st.ref_struct(ref), 1)
            # ranks / AUC of a fixed held-out set from the reference's predict_ranks and
            # calculate_auc_from_rank (evaluation.py:247-249)
            train, test = H.rank_problem(coo)
            ranks = np.zeros_like(test.data)
            ref.predict_ranks(C(item_f), C(user_f), C(test), C(train), ranks, st.ref_struct(ref), 1)
            import scipy.sparse as sp
            rmat = sp.csr_matrix((ranks.copy(), test.indices, test.indptr), shape=test.shape)
            ntp = np.squeeze(np.array(train.getnnz(axis=1)).astype(np.int32))
            auc = np.zeros(nu, np.float32)
            ref.calculate_auc_from_rank(C(rmat), ntp, rmat.data, auc, 1)
            np.savez_compressed(
                os.path.join(OUT, "%s__%s.npz" % (case[0], loss)),
                predictions=pred[:: max(1, len(pred) // 512)], ranks=ranks, auc=auc,
                **{n: getattr(st, n) for n in oracle.ARRAYS})


if __name__ == "__main__":
    main()
