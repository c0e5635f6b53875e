"""`RefLightFM`: the host-side LightFM class driving the REFERENCE's compiled
native module (oracle/_ref) -- or the C oracle -- instead of the HIP backend.

TEST INFRASTRUCTURE ONLY (see oracle/lfm_oracle.c).  Used by tests/, tools/ and
bench.py's cpu_baseline leg to time / evaluate the reference CPU path on the same
inputs, with the same host-side RNG consumption as lightfm/lightfm.py:668-759.
"""
import time

import numpy as np

from lightfm_amd.lightfm import LightFM
from oracle import oracle


class RefLightFM(LightFM):
    kind = "fast"  # oracle/_ref/<kind>: "fast" = default-flag build, "strict" = parity build

    def _run_epochs(self, item_features, user_features, interactions, sample_weight, num_threads,
                    epochs, verbose):
        ref = oracle.ref_module(self.kind)
        C = ref.CSRMatrix
        for _ in self._progress(epochs, verbose=verbose):
            loss = self.loss
            if loss in ("warp", "bpr", "warp-kos"):
                positives = C(self._get_positives_lookup_matrix(interactions))
            shuffle = np.arange(len(interactions.data), dtype=np.int32)
            self.random_state.shuffle(shuffle)
            fl = ref.FastLightFM(
                self.item_embeddings, self.item_embedding_gradients, self.item_embedding_momentum,
                self.item_biases, self.item_bias_gradients, self.item_bias_momentum,
                self.user_embeddings, self.user_embedding_gradients, self.user_embedding_momentum,
                self.user_biases, self.user_bias_gradients, self.user_bias_momentum,
                self.no_components, int(self.learning_schedule == "adadelta"), self.learning_rate,
                self.rho, self.epsilon, self.max_sampled)
            args = (C(item_features), C(user_features))
            t_native = time.perf_counter()
            if loss == "warp":
                ref.fit_warp(*args, positives, interactions.row, interactions.col,
                             interactions.data, sample_weight, shuffle, fl, self.learning_rate,
                             self.item_alpha, self.user_alpha, num_threads, self.random_state)
            elif loss == "bpr":
                ref.fit_bpr(*args, positives, interactions.row, interactions.col,
                            interactions.data, sample_weight, shuffle, fl, self.learning_rate,
                            self.item_alpha, self.user_alpha, num_threads, self.random_state)
            elif loss == "warp-kos":
                ref.fit_warp_kos(*args, positives, interactions.row, shuffle, fl,
                                 self.learning_rate, self.item_alpha, self.user_alpha, self.k,
                                 self.n, num_threads, self.random_state)
            else:
                ref.fit_logistic(*args, interactions.row, interactions.col, interactions.data,
                                 sample_weight, shuffle, fl, self.learning_rate, self.item_alpha,
                                 self.user_alpha, num_threads)
            if hasattr(self, "native_seconds"):  # bench.py: duration of the native epoch call alone
                self.native_seconds.append(time.perf_counter() - t_native)
            self._check_finite()
