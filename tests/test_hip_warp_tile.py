"""GPU parity tests of the lane-group WARP tile kernel (lightfm_amd/csrc/warp_tile.hip),
the kernel bench.py measures: parallel mode, identity features, no regularisation.

Bars (all through the C ABI's Python binding, against the CPU oracle with one PRNG
stream per shuffled position -- the rule both sides share):
  * frozen weights (sample_weight = 0, the reference's own trick,
    tests/test_movielens.py:517-533): chosen negative and sample count of EVERY
    position exact, totals of draws / updates / in_positives probes exact, weights
    untouched -- for every supported d, single- and multi-batch max_sampled, skipped
    (Y <= 0) rows and positives rows long enough for several search rounds;
  * one interaction per launch (launches_per_epoch = n): the parallel kernel is then
    sequential, and weights, biases and accumulators must be BIT-EXACT;
  * four concurrent interactions per launch that share no row (checked in numpy from
    the shared PRNG rule): BIT-EXACT as well -- every lane group's update path;
  * the generic one-interaction-per-wavefront kernel gives the same logs;
  * the SHIPPED steady-state kernel (warp_tile_ahead.hpp; atomic publication, `tile_ahead == 1`
    asserted): conflict-free launches of one pass of one wavefront (four lane groups updating at
    once) and of two passes of four wavefronts (the gather of pass t + 1 issued before pass t
    publishes), and a sequential run over dense positives rows in which the first violator is
    regularly one of the user's positives, so that the re-read branch performs non-zero updates
    -- states within one float32 ulp of the oracle's, bit-identical where new - old is exact.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fast():
    import lightfm_amd._lightfm_fast as f
    from lightfm_amd import _native
    assert _native.device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    return f


_DEFAULTS = dict(mode="parallel", launches_per_epoch=0, first_batch=0, max_waves=0,
                 log_samples=False, warp_kernel=0, update_mode=0, debug=0)

# options.debug bits 0-2 force the tile kernel's interactions per wavefront pass (NG = 4, 2, 1:
# 16, 32, 64 lanes per row); 0 = the session's automatic choice.  Bit 6 (64) selects the
# register-staged NG = 4 variant instead of the LDS-DMA one (global_load_lds_dwordx4).
NGS = [0, 4, 2, 1]
REGS = 64


@pytest.fixture(autouse=True)
def _reset_options():
    from lightfm_amd.options import options
    options.set(**_DEFAULTS)
    yield
    options.set(**_DEFAULTS)


def _hip_warp(fast, coo, st, shuffle, seeds, weight):
    Cm = fast.CSRMatrix
    nu, ni = coo.shape
    fl = fast.FastLightFM(*st.arrays(), st.d, int(st.schedule == "adadelta"), st.lr, st.rho, st.eps,
                          st.max_sampled)
    fast.fit_warp(Cm(H.identity_features(ni)), Cm(H.identity_features(nu)), Cm(H.positives_csr(coo)),
                  coo.row, coo.col, coo.data, weight, shuffle, fl, 0.05, 0.0, 0.0, len(seeds),
                  H.FixedRandom(seeds))


def _orc_warp(coo, st, shuffle, seeds, weight):
    nu, ni = coo.shape
    o = oracle.Opts(len(shuffle), rng_mode=1, log=True)
    oracle.fit_warp(H.identity_features(ni), H.identity_features(nu), H.positives_csr(coo), coo.row,
                    coo.col, coo.data, weight, shuffle, st, 0.0, 0.0, seeds, o)
    return o


def _spread(st):
    """Scale the freshly initialised embeddings so that scores have a standard deviation
    of about 3: the margin test (PYX:875) then sees both outcomes and sample counts
    cover the whole 1..max_sampled range."""
    a = 3.0 / st.d ** 0.25
    st.item_embeddings *= 2 * st.d * a
    st.user_embeddings *= 2 * st.d * a


FROZEN = [
    # (id, n_users, n_items, nnz, d, max_sampled, first_batch, ratings)
    ("d64-ms10", 300, 200, 6000, 64, 10, 0, False),
    ("d64-ms10-fb3", 300, 200, 6000, 64, 10, 3, True),
    ("d64-ms1", 120, 90, 1500, 64, 1, 0, False),
    ("d64-ms40-multibatch", 200, 150, 5000, 64, 40, 0, True),
    ("d8-ms7", 60, 40, 500, 8, 7, 0, True),
    ("d32-ms15", 150, 400, 3000, 32, 15, 0, False),
    ("d48-ms12", 90, 70, 1200, 48, 12, 5, False),
    ("d100-ms10", 200, 120, 4000, 100, 10, 0, True),
    ("d128-ms35-multibatch", 100, 300, 3000, 128, 35, 0, False),
    ("longrows", 12, 6000, 30000, 64, 10, 0, False),
    ("two-items", 50, 2, 60, 64, 10, 0, False),
    ("d256-ms10", 60, 90, 1500, 256, 10, 0, False),
    # widths that are no multiple of 4 (the reference's default is 10): rows padded on the device, production kernels
    ("d10-ms10", 150, 100, 3000, 10, 10, 0, True),
    ("d30-ms10", 150, 100, 3000, 30, 10, 0, False),
    ("d50-ms12", 90, 70, 1200, 50, 12, 0, True),
    ("d200-ms10", 60, 90, 1500, 200, 10, 0, False),
    ("d1-ms5", 40, 30, 400, 1, 5, 0, False),
]


@pytest.mark.parametrize("case", FROZEN, ids=[c[0] for c in FROZEN])
@pytest.mark.parametrize("kernel", ["generic", "tile-auto", "tile-ng4", "tile-ng4-regs", "tile-ng2", "tile-ng1"])
def test_frozen_weights_samples_exact(fast, case, kernel):
    from lightfm_amd.options import options
    _, nu, ni, nnz, d, ms, fb, ratings = case
    ng = {"generic": 0, "tile-auto": 0, "tile-ng4": 4, "tile-ng4-regs": 4 | REGS, "tile-ng2": 2, "tile-ng1": 1}[kernel]
    if ((ng & 7) == 4 and d > 64) or ((ng & 7) == 2 and d > 128) or ((ng & 7) == 1 and d > 128):
        pytest.skip("row wider than the lane group covers: the session falls back to fewer per wave")
    coo = H.make_interactions(nu, ni, nnz, seed=17, ratings=ratings, zipf=0.6)
    rng = np.random.RandomState(9)
    st = oracle.State(ni, nu, d, rng, max_sampled=ms)
    _spread(st)
    st.item_biases[:] = rng.randn(ni).astype(np.float32) * 0.3
    st.user_biases[:] = rng.randn(nu).astype(np.float32) * 0.3
    a, b = st.copy(), st.copy()
    zeros = np.zeros_like(coo.data)
    shuffle, seeds = H.epoch_inputs(coo, rng)
    options.set(log_samples=True, launches_per_epoch=3, first_batch=fb,
                warp_kernel=1 if kernel == "generic" else 0, debug=ng)
    _hip_warp(fast, coo, a, shuffle, seeds, zeros)
    o = _orc_warp(coo, b, shuffle, seeds, zeros)
    neg, sampled = options.last_logs
    assert np.array_equal(sampled, o.sampled), "sample counts differ"
    assert np.array_equal(neg, o.neg), "negative (rank) indices differ"
    assert options.last_counters == o.counters
    assert o.counters[2] > 0 or ni <= 2, "no violator found: the case does not exercise the update path"
    H.assert_states_equal(a, st, exact=True)
    if kernel == "generic":   # warp_kernel = 1: anything but the tile kernel (the row-stream kernels where they apply)
        assert options.last_kernel_used != 1
    elif kernel == "tile-auto":  # every width up to 256 runs the tile kernel (one interaction per pass beyond 128)
        assert options.last_kernel_used == 1, (d, options.last_kernel_used)


SEQ = [("d64-adagrad", 64, "adagrad", 10), ("d32-adadelta", 32, "adadelta", 6),
       ("d128-adagrad", 128, "adagrad", 10), ("d20-adagrad", 20, "adagrad", 20),
       ("d200-adagrad", 200, "adagrad", 10), ("d10-adagrad", 10, "adagrad", 10), ("d50-adagrad", 50, "adagrad", 10),
       ("d30-adadelta", 30, "adadelta", 10)]


@pytest.mark.parametrize("case", SEQ, ids=[c[0] for c in SEQ])
@pytest.mark.parametrize("update_mode", [1, 3], ids=["store", "atomic"])
@pytest.mark.parametrize("ng", [4, 4 | REGS, 2, 1], ids=["4", "4-regs", "2", "1"])
def test_one_interaction_per_launch_is_bit_exact(fast, case, update_mode, ng):
    """launches_per_epoch = n makes the Hogwild kernel sequential: the sample logs must then
    equal the oracle's (same per-position streams) exactly and, two epochs later, every
    array bit for bit with the default plain-store Hogwild update (update_mode 1).  Mode 3
    publishes new - old with global_atomic_add_f32: old + fl32(new - old) reproduces `new`
    except where the subtraction is inexact (a weight crossing zero), so there the bar is
    one float32 ulp of the largest weight."""
    from lightfm_amd.options import options
    _, d, sched, ms = case
    if ((ng & 7) == 4 and d > 64) or ((ng & 7) in (1, 2) and d > 128):
        pytest.skip("row wider than the lane group covers")
    coo = H.make_interactions(40, 30, 260, seed=3, ratings=True)
    rng = np.random.RandomState(4)
    st = oracle.State(30, 40, d, rng, schedule=sched, max_sampled=ms)
    _spread(st)
    a, b = st.copy(), st.copy()
    options.set(log_samples=True, launches_per_epoch=len(coo.data), update_mode=update_mode, debug=ng)
    for _ in range(2):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _hip_warp(fast, coo, a, shuffle, seeds, coo.data)
        o = _orc_warp(coo, b, shuffle, seeds, coo.data)
        neg, sampled = options.last_logs
        assert np.array_equal(sampled, o.sampled)
        assert np.array_equal(neg, o.neg)
        assert options.last_counters == o.counters
    assert not np.array_equal(a.item_embeddings, st.item_embeddings)
    assert options.last_kernel_used == 1 and options.last_tile_ng == (ng & 7), "the forced tile kernel did not run"
    if update_mode == 1:
        H.assert_states_equal(a, b, exact=True)
    else:
        H.assert_states_equal(a, b, exact=False, rtol=1e-6, atol=1e-6)


def _candidates(base_seed, positions, max_sampled, n_items):
    """All draws position i may make (numpy restatement of the shared PRNG rule)."""
    out = np.empty((len(positions), max_sampled), np.int64)
    for j, i in enumerate(positions):
        s = oracle.position_seed(int(base_seed), int(i))
        draws, _ = oracle.rand_r_stream(s, max_sampled)
        out[j] = draws % n_items
    return out


@pytest.mark.parametrize("d,group,variant", [(64, 4, 0), (64, 4, REGS), (128, 2, 0)], ids=["d64-4", "d64-4-regs", "d128-2"])
def test_concurrent_disjoint_groups_bit_exact(fast, d, group, variant):
    """`group` interactions per launch = one per lane group of ONE wavefront.  When no
    interaction of a launch reads or writes a row another one writes, the concurrent
    result equals the sequential one bit for bit."""
    from lightfm_amd.options import options
    nu, ni, n, ms = 400, 40000, 160, 10
    rng = np.random.RandomState(11)
    found = None
    for attempt in range(50):
        users = rng.permutation(nu)[:n].astype(np.int32)
        items = rng.permutation(ni)[:n].astype(np.int32)
        coo = sp.coo_matrix((np.ones(n, np.float32), (users, items)), shape=(nu, ni), dtype=np.float32)
        shuffle, seeds = H.epoch_inputs(coo, rng)
        cands = _candidates(seeds[0], np.arange(n), ms, ni)
        ok = True
        for l0 in range(0, n, group):
            rows = [set(cands[j].tolist()) | {int(coo.col[shuffle[j]])} for j in range(l0, l0 + group)]
            if len(set().union(*rows)) != sum(len(r) for r in rows):
                ok = False
                break
        if ok:
            found = (coo, shuffle, seeds)
            break
    assert found is not None, "no conflict-free arrangement found"
    coo, shuffle, seeds = found
    st = oracle.State(ni, nu, d, np.random.RandomState(2), max_sampled=ms)
    _spread(st)
    a, b = st.copy(), st.copy()
    options.set(log_samples=True, launches_per_epoch=n // group, update_mode=1, debug=group | variant)
    _hip_warp(fast, coo, a, shuffle, seeds, coo.data)
    o = _orc_warp(coo, b, shuffle, seeds, coo.data)
    neg, sampled = options.last_logs
    assert np.array_equal(sampled, o.sampled)
    assert np.array_equal(neg, o.neg)
    assert o.counters[2] > n // 4
    H.assert_states_equal(a, b, exact=True)


def _conflict_free(nu, ni, n, per_launch, ms, rng, attempts=200):
    """n interactions with distinct users and positives such that no launch of `per_launch` consecutive shuffled
    positions reads or writes an item row another interaction of the launch writes."""
    for _ in range(attempts):
        users = rng.permutation(nu)[:n].astype(np.int32)
        items = rng.permutation(ni)[:n].astype(np.int32)
        coo = sp.coo_matrix((np.ones(n, np.float32), (users, items)), shape=(nu, ni), dtype=np.float32)
        shuffle, seeds = H.epoch_inputs(coo, rng)
        cands = _candidates(seeds[0], np.arange(n), ms, ni)
        ok = True
        for l0 in range(0, n, per_launch):
            rows = [set(cands[j].tolist()) | {int(coo.col[shuffle[j]])} for j in range(l0, min(n, l0 + per_launch))]
            if len(set().union(*rows)) != sum(len(r) for r in rows):
                ok = False
                break
        if ok:
            return coo, shuffle, seeds
    raise AssertionError("no conflict-free arrangement found")


@pytest.mark.parametrize("d", [64, 16, 10, 4], ids=["d64", "d16-narrow", "d10-narrow", "d4-narrow"])
@pytest.mark.parametrize("ustore", [False, True], ids=["atomics", "user-rows-stored"])
@pytest.mark.parametrize("per_launch", [4, 32], ids=["one-pass", "four-waves-two-passes"])
def test_tile_ahead_concurrent_updates_match_the_oracle(fast, per_launch, ustore, d):
    """The kernel bench.py's headline runs on (fit_warp_tile_ahead_kernel: update_mode 0, debug 0 apart from the forced
    lane-group width the session picks at full residency anyway), with NON-ZERO updates under concurrency: four lane
    groups of a wavefront update in the same pass (speculative cU / cP / cN copies, the four-at-once bias cells);
    with 32 positions per launch four wavefronts run two passes each, the second pass's rows requested before the
    first publishes.  No two interactions of a launch share a row, so the result is the sequential oracle's.
    "user-rows-stored": the instantiation that writes the user row of an update with plain stores (lfm_opts.user_store;
    the session picks it for models with >= 8 users per interaction in flight -- C2 -- here forced by debug bit 11)."""
    from lightfm_amd.options import options
    ms = 10
    nu, ni, n = (400, 40000, 160) if per_launch == 4 else (800, 300000, 256)
    coo, shuffle, seeds = _conflict_free(nu, ni, n, per_launch, ms, np.random.RandomState(23))
    st = oracle.State(ni, nu, d, np.random.RandomState(2), max_sampled=ms)
    _spread(st)
    st.item_biases[:] = np.random.RandomState(3).randn(ni).astype(np.float32) * 0.3
    a, b = st.copy(), st.copy()
    options.set(log_samples=True, launches_per_epoch=n // per_launch, update_mode=0, debug=4 | (2048 if ustore else 0), max_waves=16)
    _hip_warp(fast, coo, a, shuffle, seeds, coo.data)
    assert options.last_tile_ahead == 1 and options.last_tile_ng == 4, "the steady-state tile kernel did not run"
    # rows of <= 16 floats: the narrow-model kernel (csrc/warp_tile_narrow.hpp: two interactions per lane group)
    assert bool(options.last_plan_flags & 64) == (d <= 16), (d, options.last_plan_flags)
    if d >= 10:  # (at d = 4 the 6 KB user table is not in uncached memory: the forced plain stores do not apply)
        assert options.last_user_store == int(ustore)  # (a few hundred users: the session's own rule says atomics)
    o = _orc_warp(coo, b, shuffle, seeds, coo.data)
    neg, sampled = options.last_logs
    assert np.array_equal(sampled, o.sampled)
    assert np.array_equal(neg, o.neg)
    assert options.last_counters == o.counters
    assert o.counters[2] > n // 4
    assert not np.array_equal(a.item_embeddings, st.item_embeddings)
    H.assert_states_within_ulps(a, b, ulps=4)


@pytest.mark.parametrize("d", [4, 10, 12, 16])
@pytest.mark.parametrize("layout", [0, 1, 2], ids=["separate-tables", "row-pairs", "row-pairs-with-biases"])
@pytest.mark.parametrize("ustore", [False, True], ids=["atomics", "user-rows-stored"])
def test_narrow_kernel_row_layouts_match_the_oracle(fast, monkeypatch, ustore, layout, d):
    """The three table layouts of the narrow-model kernel (LIGHTFM_AMD_ROW_PAIRS): W / G / b / bG in their own tables; W and G
    of a feature in one 128-byte line (RP instantiations: both halves published by one instruction); and, for d <= 12 --
    the SHIPPED layout at the reference's default width -- the bias cells in slot d of each half as well (BIN
    instantiations, lfm_opts.plan_flags bit 7: the gather brings the bias with the row, lane d of a group runs the bias
    cell, an update is three line operations).  Non-zero updates under concurrency (four wavefronts, two passes each, no
    two interactions of a launch sharing a row): the sequential oracle's result, incl. both sides' bias cells."""
    from lightfm_amd.options import options
    monkeypatch.setenv("LIGHTFM_AMD_ROW_PAIRS", str(layout))
    ms, per_launch = 10, 32
    nu, ni, n = 800, 300000, 256
    coo, shuffle, seeds = _conflict_free(nu, ni, n, per_launch, ms, np.random.RandomState(29))
    st = oracle.State(ni, nu, d, np.random.RandomState(2), max_sampled=ms)
    _spread(st)
    st.item_biases[:] = np.random.RandomState(3).randn(ni).astype(np.float32) * 0.3
    st.user_biases[:] = np.random.RandomState(4).randn(nu).astype(np.float32) * 0.3
    a, b = st.copy(), st.copy()
    options.set(log_samples=True, launches_per_epoch=n // per_launch, update_mode=0, debug=4 | (2048 if ustore else 0), max_waves=16)
    _hip_warp(fast, coo, a, shuffle, seeds, coo.data)
    assert options.last_plan_flags & 64, options.last_plan_flags
    assert bool(options.last_plan_flags & 128) == (layout == 2 and d <= 12), (layout, d, options.last_plan_flags)
    o = _orc_warp(coo, b, shuffle, seeds, coo.data)
    neg, sampled = options.last_logs
    assert np.array_equal(sampled, o.sampled)
    assert np.array_equal(neg, o.neg)
    assert options.last_counters == o.counters
    assert o.counters[2] > n // 4
    assert not np.array_equal(a.item_biases, st.item_biases) and not np.array_equal(a.user_biases, st.user_biases)
    H.assert_states_within_ulps(a, b, ulps=4)


@pytest.mark.parametrize("d", [64, 10, 16], ids=["d64", "d10-narrow", "d16-narrow"])
def test_tile_ahead_reread_branch_updates_match_the_oracle(fast, d):
    """Dense positives rows (a third of the catalogue per user) and near-zero scores: nearly every candidate violates
    the margin, so the FIRST violator is one of the user's positives in about a third of the interactions; the choice
    is then a later violator whose row the steady-state kernel re-reads from the table (warp_tile_ahead.hpp, `only_neg`)
    instead of using its speculative copy.  One interaction per launch (sequential): states within an ulp of the
    oracle's after two epochs.  That the branch ran is read off the counters: every position found a negative, so
    each in_positives probe that hit (probes - updates) belongs to a position whose first violator was a positive."""
    from lightfm_amd.options import options
    nu, ni, ms = 24, 60, 10
    rng = np.random.RandomState(8)
    dense = rng.rand(nu, ni) < 0.33
    dense[:, 0] = True
    coo = sp.coo_matrix(dense.astype(np.float32))
    coo = sp.coo_matrix((coo.data.astype(np.float32), (coo.row.astype(np.int32), coo.col.astype(np.int32))),
                        shape=(nu, ni), dtype=np.float32)
    st = oracle.State(ni, nu, d, rng, max_sampled=ms)
    a, b = st.copy(), st.copy()
    options.set(log_samples=True, launches_per_epoch=len(coo.data), update_mode=0, debug=4)
    reread = 0
    for _ in range(2):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _hip_warp(fast, coo, a, shuffle, seeds, coo.data)
        assert options.last_tile_ahead == 1 and options.last_tile_ng == 4
        assert bool(options.last_plan_flags & 64) == (d <= 16)
        o = _orc_warp(coo, b, shuffle, seeds, coo.data)
        neg, sampled = options.last_logs
        assert np.array_equal(sampled, o.sampled)
        assert np.array_equal(neg, o.neg)
        assert options.last_counters == o.counters
        if (o.neg >= 0).all():  # positives visited, draws, updates, probes
            reread += o.counters[3] - o.counters[2]
    assert reread > len(coo.data) // 8, "the first violator was rarely a positive: the re-read branch is not exercised"
    H.assert_states_within_ulps(a, b, ulps=4, min_exact=0.1)  # (every cell is updated dozens of times here)


def test_tile_training_learns_like_the_oracle(fast):
    """Full Hogwild training with the tile kernel: fit quality within a few percent of the
    sequential oracle's after 5 epochs (neither side is order-deterministic in general)."""
    coo = H.make_interactions(2000, 1500, 80000, seed=21)
    rng = np.random.RandomState(5)
    st = oracle.State(1500, 2000, 64, rng)
    a, b = st.copy(), st.copy()
    for _ in range(5):
        shuffle, seeds = H.epoch_inputs(coo, rng)
        _hip_warp(fast, coo, a, shuffle, seeds, coo.data)
        _orc_warp(coo, b, shuffle, seeds, coo.data)
    item_f, user_f = H.identity_features(1500), H.identity_features(2000)

    def margin(s):
        r = np.random.RandomState(0)
        pos = oracle.predict(item_f, user_f, coo.row, coo.col, s)
        neg = oracle.predict(item_f, user_f, coo.row,
                             r.randint(0, 1500, size=len(coo.row)).astype(np.int32), s)
        return float(np.mean(pos - neg)), float(np.mean(pos > neg))

    (ma, aa), (mb, ab) = margin(a), margin(b)
    assert ab > 0.8
    assert abs(aa - ab) < 0.04, (aa, ab)  # Hogwild: run-to-run variation
    assert abs(ma - mb) / abs(mb) < 0.2, (ma, mb)


@pytest.mark.parametrize("d", [4, 10, 12, 16])
@pytest.mark.parametrize("waves", [0, 64], ids=["full-grid", "two-workgroups-many-passes"])
def test_narrow_kernel_frozen_weights_samples_exact(fast, d, waves):
    """The narrow-model kernel (csrc/warp_tile_narrow.hpp: rows of <= 16 floats, two interactions per lane group, eight per
    wavefront pass, four rows per LDS-DMA instruction) with frozen weights: every position's negative and sample count and
    the counters equal the oracle's -- across the whole grid, and with two workgroups walking hundreds of passes each (the
    record pipeline, the gather issued one pass ahead, launch tails that end inside a pass)."""
    from lightfm_amd.options import options
    nu, ni = 3000, 2500
    coo = H.make_interactions(nu, ni, 60_011, seed=29, ratings=True, zipf=0.7)
    rng = np.random.RandomState(4)
    st = oracle.State(ni, nu, d, rng, max_sampled=10)
    _spread(st)
    st.item_biases[:] = rng.randn(ni).astype(np.float32) * 0.3
    st.user_biases[:] = rng.randn(nu).astype(np.float32) * 0.3
    a, b = st.copy(), st.copy()
    zeros = np.zeros_like(coo.data)
    shuffle, seeds = H.epoch_inputs(coo, rng)
    options.set(log_samples=True, launches_per_epoch=3, ramp_k=-1, max_waves=waves, debug=4)
    _hip_warp(fast, coo, a, shuffle, seeds, zeros)
    assert options.last_kernel_used == 1 and options.last_plan_flags & 64, (options.last_kernel_used, options.last_plan_flags)
    o = _orc_warp(coo, b, shuffle, seeds, zeros)
    neg, sampled = options.last_logs
    assert np.array_equal(sampled, o.sampled), "sample counts differ at %d positions" % int((sampled != o.sampled).sum())
    assert np.array_equal(neg, o.neg), "negatives differ at %d positions" % int((neg != o.neg).sum())
    assert options.last_counters == o.counters
    assert o.counters[2] > 1000
    H.assert_states_equal(a, st, exact=True)


def test_narrow_kernel_training_learns_like_the_wide_one():
    """Full-concurrency training at the reference's default width (no_components = 10) through the narrow-model kernel and,
    with LIGHTFM_AMD_TILE_PAIRS-independent means (debug bit 10: the plain tile kernel), through the wide one: both learn."""
    from lightfm_amd import LightFM
    from lightfm_amd.options import options
    # (a catalogue large enough for four interactions per lane group at full residency: the smaller side bounds the
    # interactions in flight, and below 8 192 of them the session runs one interaction per wavefront pass)
    coo = H.make_interactions(14000, 11000, 450_000, seed=12, zipf=0.8)
    rows, cols = np.ascontiguousarray(coo.row), np.ascontiguousarray(coo.col)
    negs = np.random.RandomState(0).randint(0, 11000, size=coo.nnz).astype(np.int32)
    acc = {}
    for arm, debug in (("narrow", 0), ("plain", 1024)):
        options.set(debug=debug)
        m = LightFM(loss="warp", random_state=7)  # no_components = 10
        m.fit(coo, epochs=6)
        st = m._last_epoch_stats[-1]
        assert st["kernel_used"] == 1 and bool(st["plan_flags"] & 64) == (arm == "narrow"), (arm, st)
        acc[arm] = float(np.mean(m.predict(rows, cols) > m.predict(rows, negs)))
    options.set(debug=0)
    print("pairwise accuracy", acc)
    assert acc["narrow"] > 0.8 and abs(acc["narrow"] - acc["plain"]) < 0.02, acc
