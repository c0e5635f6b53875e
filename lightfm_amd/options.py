"""Execution options of the HIP backend (not part of the reference's API).

mode               "parallel" (default): Hogwild over thousands of wavefronts, one PRNG
                   stream per shuffled position.  "serial": one wavefront in the
                   reference's order with its rand_r streams -- bit-exact, for parity tests.
launches_per_epoch kernel launches per epoch in parallel mode (0 = auto).
first_batch        negatives scored speculatively in the first batch (0 = auto).
max_waves          FIXED cap on interactions in flight, parallel mode; 0 = auto = ramp with the
                   training history: at most (interactions trained on so far) / ramp_k in flight,
                   up to the whole chip.
ramp_k             see max_waves; 0 = auto (32), < 0 = no ramp.
shared_cap         steady-state bound on interactions in flight when feature rows are shared between
                   items / users (hybrid models); 0 = auto, < 0 = none.
history            interactions the model was already trained on, for the low-level epoch
                   functions (LightFM.fit_partial keeps count itself).
update_mode        0 auto (= 3), 1 plain load/store Hogwild, 2 no writes (profiling),
                   3 atomic deltas (global_atomic_add_f32, the default).
warp_kernel        0 auto (lane-group tile kernel where it applies), 1 force the generic
                   one-interaction-per-wavefront WARP kernel, 2 tile kernel with phase timers.
feat_kernel        0 auto (the pipelined row-stream kernels for feature CSRs / BPR / k-OS / logistic
                   where they apply), 1 force the generic one-interaction-per-wavefront kernels.
debug              bits 0-2 force the tile kernel's interactions per wavefront pass (1, 2, 4).
log_samples        record (negative, sampled) per shuffled position into last_logs.
device_shuffle     LightFM.fit_partial, parallel mode: True (default) builds each epoch's shuffle
                   on the device from two RandomState draws (lfm_session_device_shuffle) -- unless
                   the model was constructed with a RandomState INSTANCE, whose stream is always
                   consumed exactly like the reference; False draws numpy's
                   random_state.shuffle(arange(n)) on the host like the reference (LFM:689-690)
                   and uploads it.  Serial mode always uses the host shuffle.
cache_scoring_session  True (default): LightFM keeps a device session with its embeddings and biases
                   between predict / predict_rank / get_*_representations calls (re-validated by checksum
                   per call, dropped by fit_partial); False: every call uploads them anew.
device             the GPU LightFM.fit_partial / predict / predict_rank run on (LIGHTFM_AMD_DEVICE, default 0;
                   DistributedFit and bench.py take theirs from the rank).
shuffle_ahead      with device_shuffle: the permutation of epoch e + 1 is written on a stream of its own while epoch e
                   trains (two slots alternate; lfm_session_device_shuffle_ahead).  The RandomState draws keep their order.
                   Off by default: measured without effect on C2 (1.250 / 1.249 against 1.261 / 1.238 G interactions/s,
                   profiles/r05_visit_f.txt -- the 0.36 ms the shuffle takes alone it takes from the epoch kernels when
                   it runs beside them).
trim_after_fit     True: LightFM.fit_partial hands the device memory its session used back to the HIP runtime when it returns
                   (lfm_device_trim; for processes that share the GPU with other code).  Off by default: the pool that keeps
                   released blocks is what fixed the stale-memory fault of round 2, and the next fit reuses them.
host_positives     True: the positives lookup is built on the host (interactions.tocsr(), LFM:365-372)
                   and uploaded; False (default): built on the device from the uploaded COO.

Environment: LIGHTFM_AMD_MODE, _LAUNCHES, _FIRST_BATCH, _MAX_WAVES, _RAMP_K, _UPDATE_MODE,
_WARP_KERNEL, _FEAT_KERNEL, _DEBUG, _DEVICE_SHUFFLE, _DEVICE; LIGHTFM_AMD_TABLE_ALLOC / _TABLE_ALLOC_MASK select the
allocation flavour of the weight tables (csrc/session.hip).
"""
import os


class _Options(object):
    def __init__(self):
        self.mode = os.environ.get("LIGHTFM_AMD_MODE", "parallel")
        self.launches_per_epoch = int(os.environ.get("LIGHTFM_AMD_LAUNCHES", "0"))
        self.first_batch = int(os.environ.get("LIGHTFM_AMD_FIRST_BATCH", "0"))
        self.max_waves = int(os.environ.get("LIGHTFM_AMD_MAX_WAVES", "0"))
        self.update_mode = int(os.environ.get("LIGHTFM_AMD_UPDATE_MODE", "0"))
        self.feat_kernel = int(os.environ.get("LIGHTFM_AMD_FEAT_KERNEL", "0"))
        self.warp_kernel = int(os.environ.get("LIGHTFM_AMD_WARP_KERNEL", "0"))
        self.debug = int(os.environ.get("LIGHTFM_AMD_DEBUG", "0"))
        self.ramp_k = int(os.environ.get("LIGHTFM_AMD_RAMP_K", "0"))
        self.shared_cap = int(os.environ.get("LIGHTFM_AMD_SHARED_CAP", "0"))
        self.history = 0
        self.device = int(os.environ.get("LIGHTFM_AMD_DEVICE", "0"))
        self.device_shuffle = os.environ.get("LIGHTFM_AMD_DEVICE_SHUFFLE", "1") != "0"
        self.host_positives = os.environ.get("LIGHTFM_AMD_HOST_POSITIVES", "0") != "0"
        self.shuffle_ahead = os.environ.get("LIGHTFM_AMD_SHUFFLE_AHEAD", "0") != "0"
        self.trim_after_fit = os.environ.get("LIGHTFM_AMD_TRIM_AFTER_FIT", "0") != "0"
        self.cache_scoring_session = os.environ.get("LIGHTFM_AMD_CACHE_SCORING", "1") != "0"
        self.log_samples = False
        self.last_counters = None
        self.last_kernel_ms = None
        self.last_kernel_used = None  # 0 generic kernels, 1 lane-group tile kernel, 2 row-stream kernels
        self.last_launches = None
        self.last_streams_used = None
        self.last_tile_ng = None     # interactions per wavefront pass of the tile kernel (4, 2, 1), 0 = another kernel ran
        self.last_tile_ahead = None  # 1 = the steady-state (gather-ahead) tile kernel ran the last launch
        self.last_user_store = None  # 1 = user rows were written with plain stores (lfm_opts.user_store)
        self.last_plan_flags = None  # lfm_opts.plan_flags of the last epoch (bias snapshots, uncached tables, 64-bit item rows)
        self.last_phase_cycles = None
        self.last_logs = None

    def set(self, **kw):
        for k, v in kw.items():
            if not hasattr(self, k):
                raise ValueError("unknown option %r" % k)
            setattr(self, k, v)
        if self.mode not in ("parallel", "serial"):
            raise ValueError("mode must be 'parallel' or 'serial'")
        return self


options = _Options()
