"""Device-resident DQN loop over a batch of lattices: act -> environment step -> replay ring -> update.

This is the vectorised form of the loop keras-rl runs one lattice / one minibatch at a time
(`Agent.fit` + `DQNAgent.forward/backward`, call site
/root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:138-152).  With one lattice it
performs exactly that sequence per step; with N lattices every launch handles all of them and nothing is
copied to the host inside a step.

Replay ring (SequentialMemory(limit, window_length=1), :109): time-major device tensors
    obs      uint8 [T, N, C, H, W]     action int32 [T, N]     reward float [T, N]     terminal uint8 [T, N]
row (t, i) = what lattice i saw / did / received at vector step t; its successor observation is row (t+1, i).
The environment kernel writes the new observation straight into slot t+1, so there is no append copy; a
lattice that terminated spends its next step being reset (keras-rl's extra forward/backward on the terminal
observation), which is what makes keras-rl's "skip entries whose predecessor was terminal" sampling rule carry
over unchanged.
"""
import ctypes
import os

import torch

from . import _lib, dist as _dist, qnet as _q
from ._lib import check, ptr


MIN_FILLED = 4      # ring slots an update needs: keras-rl asserts nb_entries >= window_length + 2 (csrc/common.h dq_replay_row)


class ObsRingView:
    """`DQNCore.obs_ring` where the ring holds PATCH WORDS (DQNCore.compact): the padded uint8 observations [T, N, C, H, W] the
    reference's memory would hold (SequentialMemory stores the env's board_state, TRAIN:109), decoded on demand from
    `DQNCore.patch_ring` (env.patch_to_obs: the image is a fixed function of the words).  Indexing returns decoded tensors; copy_()
    encodes.  Tests, pickling and diagnostics only -- nothing in the loop touches it."""

    def __init__(self, core):
        self._c = core

    @property
    def shape(self):
        p = self._c.patch_ring
        return torch.Size(tuple(p.shape[:2]) + tuple(self._c.env.obs_shape))

    dtype = torch.uint8

    @property
    def device(self):
        return self._c.patch_ring.device

    def __getitem__(self, idx):
        return self._c.env.patch_to_obs(self._c.patch_ring[idx])

    def clone(self):
        return self._c.env.patch_to_obs(self._c.patch_ring)

    def cpu(self):
        return self.clone().cpu()

    def copy_(self, src):
        if isinstance(src, ObsRingView):
            self._c.patch_ring.copy_(src._c.patch_ring)
        else:
            self._c.patch_ring.copy_(self._c.env.obs_to_patch(torch.as_tensor(src).to(self.device)))
        return self

    def __eq__(self, other):
        return self.clone() == (other.clone() if isinstance(other, ObsRingView) else other)


class DQNCore:
    def __init__(self, env, net, batch_size=32, memory_limit=50000, gamma=0.99, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7,
                 target_model_update=10000, enable_double_dqn=True, seed=None, rank=0, world_size=1, process_group=None,
                 params=None, compact=None):
        self.env, self.net = env, net
        self.N, self.A = env.n_envs, env.num_actions
        assert net.n_actions == self.A and tuple(net.input_shape) == tuple(env.obs_shape)
        self.device = env.device
        self.batch_size = int(batch_size)
        assert self.batch_size <= net.max_batch and self.N <= net.max_batch
        self.gamma, self.lr, self.beta_1, self.beta_2, self.epsilon = gamma, lr, beta_1, beta_2, epsilon
        self.target_model_update = target_model_update
        self.enable_double_dqn = enable_double_dqn
        self.seed = tuple(env.seed) if seed is None else tuple(seed)
        self.rank, self.world_size, self.pg = rank, world_size, process_group
        self.last_index = None       # rows of the most recent update (self.index, or a row of _index_multi)
        self._index_multi = None     # [k - 1][B] rows of the extra updates of a vector step (_extra_updates)
        # The extra updates' TARGET forwards on a second stream (round 5; DQ_TARGET_AHEAD=1, off by default): the target network does not change inside a
        # vector step and all the extra updates' rows are drawn by one launch in front of them (dq_replay_sample_multi), so Q_target(s1) of update i
        # needs nothing update i - 1 produces.  A second network handle (its own per-job scratch) evaluates it on a side stream, released by a mark
        # inside update i - 1's backward (dq_qnet_mark_conv_backward) and handed to update i's TD launch by an event; the updates' own launch pair then
        # carries two forwards instead of three.  Bit-identical results (tests/test_agent_gpu.py) -- and SLOWER on one MI355X: 32 updates of 4096 per
        # vector step run 123.9 us per update on one stream and 138.5 us this way (135.8 us with the side stream running free): the main stream's
        # kernels do shrink to 113.5 us, but the two event hand-offs per update cost more than the 10 us they move (NOTEBOOK.md Round 5 section 3;
        # the same result as DQ_ENV_STREAM in round 1 and DQ_DIST_MODE=overlap in round 4).
        self.target_ahead = os.environ.get("DQ_TARGET_AHEAD", "0") == "1"
        self._side = None            # (QNetwork, stream, row event, [events], Q_target(s1) rows [k - 1][B][A])
        self.pair_targets = os.environ.get("DQ_PAIR_TARGETS", "1") != "0"      # extra updates in pairs (_extra_updates); 0: three forwards per launch pair
        self._q1_pair = None
        self._rccl, self._rccl2, self._rccl_tried = None, None, False       # the learner's own RCCL communicator (dist.make_rccl), created at the first several-GPU update
        self.L = _lib.lib()
        dev = self.device
        # ring
        self.T = max(MIN_FILLED, int(memory_limit) // self.N + 1)
        C, H, W = env.obs_shape
        # Compact observations (include/deepq_hip.h dq_env_patch_output, dq_qnet_set_patch_input): the ring holds d * d patch words per
        # transition (128 bytes at d = 5 instead of the 847-byte padded image; 2^20 transitions: 134 MB instead of 0.9 GB), the environment
        # writes them, the first convolution reads them.  Where the fused chains and the one-word-per-plane environment cover the
        # configuration; DQ_COMPACT_OBS=0 (or compact=False) keeps the uint8 ring.
        if compact is None:
            compact = os.environ.get("DQ_COMPACT_OBS", "1") != "0"
        self.compact = bool(compact and not getattr(env, "wide", False) and getattr(env, "patch_supported", False) and net.fused_supported
                            and net.fused_enabled and C == env.volume_depth + env.n_action_layers and net.c_layers[0][1:] == [3, 2])
        if self.compact:
            try:
                net.set_patch_input(env.volume_depth, env.patch_stride)        # (before the first pack(): the packed buffer carries the compact kernel)
            except _lib.DeepQError:
                self.compact = False
        if self.compact:
            self.patch_ring = torch.zeros((self.T, self.N, env.patch_stride), dtype=torch.int32, device=dev)
            self._obs_ring = None
        else:
            self.patch_ring = None
            self._obs_ring = torch.zeros((self.T, self.N, C, H, W), dtype=torch.uint8, device=dev)
        self.action_ring = torch.zeros((self.T, self.N), dtype=torch.int32, device=dev)
        self.reward_ring = torch.zeros((self.T, self.N), dtype=torch.float32, device=dev)
        self.terminal_ring = torch.zeros((self.T, self.N), dtype=torch.uint8, device=dev)
        self.cur, self.filled = 0, 0
        # parameters
        self.params = net.init_params(self.seed) if params is None else params
        _dist.broadcast_(self.params, src=0, group=self.pg)
        self.target = self.params.clone()
        # f16 pieces of the weights for the fused chains: packed once per parameter change (repack()), shared by every forward
        self.params_pk = net.pack(self.params)
        self.target_pk = None if self.params_pk is None else self.params_pk.clone()
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.grads = torch.zeros_like(self.params)
        # scratch
        self.q_act = torch.zeros((self.N, self.A), dtype=torch.float32, device=dev)
        B = self.batch_size
        self.index = torch.zeros(B, dtype=torch.int32, device=dev)         # rows of the update in progress
        self._index_next = torch.zeros(B, dtype=torch.int32, device=dev)   # rows drawn ahead by an environment launch (_presampled)
        self.q1_online = torch.zeros((B, self.A), dtype=torch.float32, device=dev)
        self.q1_target = torch.zeros((B, self.A), dtype=torch.float32, device=dev)
        self.q0 = torch.zeros((B, self.A), dtype=torch.float32, device=dev)
        self.y = torch.zeros(B, dtype=torch.float32, device=dev)
        self.dq = torch.zeros((B, self.A), dtype=torch.float32, device=dev)
        self.metrics = torch.zeros(_q.TD_METRICS_FLOATS, dtype=torch.float32, device=dev)
        self.stats = torch.zeros(4, dtype=torch.int64, device=dev)
        self._stats_pending = None   # episode bookkeeping not yet launched (slot,)
        self._presampled = None      # (update number, head slot, filled slots) the minibatch in self.index was drawn for
        self.defer_stats = True      # act_and_step leaves the bookkeeping launch to the next update() (dq_post_step) / act / read_stats
        self._metrics_stale = False
        self.vector_steps = 0        # policy / environment counter
        self.updates = 0             # optimizer steps taken
        self.started = False
        # The environment launch of a fused step does not feed the same step's backward (keras-rl's sampling range), so it CAN run on its
        # own stream beside the backward chain (DQ_ENV_STREAM=1).  Measured on one MI355X: 0.299 ms per step against 0.279 on one stream
        # -- the two cross-stream event hand-offs cost more than the 15 us launch they hide -- so it is off by default.
        self._env_stream = torch.cuda.Stream(device=dev) if os.environ.get("DQ_ENV_STREAM", "0") == "1" else None
        # step_and_update on one GPU: the environment launch rides on the dense backward's first kernel (DQ_RIDE_ENV=0: separate launches)
        self.ride_env = os.environ.get("DQ_RIDE_ENV", "1") != "0"
        # DQ_TD_AUTOSCALE=1 (or auto_scale = True): the fused backward's gradient scale MEASURED from every minibatch's TD errors (dq_td_job.auto_scale: one
        # small launch per update, +3 us; any finite TD error is carried -- keras-rl's delta_clip = inf).  Default: the host-known scale (TD errors up to
        # several thousand -- the reference's recorded losses stay below 160 --; a larger one makes the WHOLE update a no-op on every rank and
        # read_metrics() at the next synchronisation switches this on for good, with a warning (DQ_TD_AUTOSCALE=0: raises DQ_ERR_RANGE instead):
        # nothing is ever partially applied)
        self.auto_scale = os.environ.get("DQ_TD_AUTOSCALE", "0") == "1"
        self.discarded_updates = 0   # optimizer steps the range guard discarded whole in this run (read_metrics; the reference would have applied them)
        self._range_ok_at = 0        # self.updates at the last synchronisation that found the range flag clear
        self.local_stats = [0, 0, 0, 0]
        self.inexact_total = 0
        self._inexact_acc = torch.zeros((), dtype=torch.int64, device=dev) if getattr(env, "wide", False) else None
        self.ar_events = None        # bench.py: a list here collects HIP-event pairs around the exposed part of the gradient all-reduce
        self.ar_pool = []            # ... taken from this pool of pre-created pairs (creating two timing events per step costs host time inside the timed region)
        self.ar_stride, self._ar_seen = 1, 0     # ... in every ar_stride-th step only (two marker packets in the stream cost the step ~4 us)
        self._e_fwd, self._e_env = torch.cuda.Event(), torch.cuda.Event()
        self._ar_stream = None       # DQ_DIST_MODE=overlap: the second stream of the dense range's all-reduce (created at its first use)
        self._env_inflight = False

    # ------------------------------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    @property
    def obs_ring(self):
        """uint8 [T, N, C, H, W]: the tensor itself, or -- compact -- a view that decodes the patch words on demand (ObsRingView)."""
        return ObsRingView(self) if self.compact else self._obs_ring

    @obs_ring.setter
    def obs_ring(self, value):
        if self.compact:
            raise AttributeError("compact ring: assign patch_ring")
        self._obs_ring = value

    def _obs_slot(self, slot):
        """(obs pointer, ring slot to arm as patch output) of an environment launch that writes ring slot `slot`."""
        if self.compact:
            self.env.arm_patch_output(self.patch_ring[slot])
            return None
        return ptr(self._obs_ring[slot])

    def _obs_job(self, **kw):
        """A forward job on the ring (rows through kw['index']) or on one slot of it (kw['slot'])."""
        slot = kw.pop("slot", None)
        ring = self.patch_ring if self.compact else self._obs_ring
        kw["obs"] = ring if slot is None else ring[slot]
        if self.compact:
            kw["patch"] = True
        return kw

    def reset_env(self):
        """env.reset() for every lattice; the new observation lands in ring slot `cur` (which has no action recorded
        yet).  If the ring already holds transitions (a second fit(), or a memory restored from a pickle), the entry
        before it is marked terminal so that no TD target bootstraps across the discontinuity."""
        self._join_env()
        if self.filled >= 2:
            prev = self.cur - 1 if self.cur > 0 else self.T - 1
            self.terminal_ring[prev].fill_(1)
        if self.compact:
            self.env.reset(out_patch=self.patch_ring[self.cur], write_obs=False)
        else:
            self.env.reset(out_obs=self._obs_ring[self.cur])
        self.filled = max(self.filled, 1)
        self.started = True

    def act_and_step(self, eps, masked_greedy=False, use_q=True, record_stats=True, presample=False):
        """One vector step: Q forward on the current observations, epsilon-greedy over the legal set, environment
        step with auto-reset; the transition is recorded in the ring by construction.  presample=True when update() follows this
        step: its replay sampling then rides on the environment launch (the rule never reads the slot this step writes)."""
        self._flush_stats()          # bookkeeping of the previous step, if no update() took it along (env buffers are about to be reused)
        env, cur = self.env, self.cur
        nxt = cur + 1 if cur + 1 < self.T else 0
        q = None
        if use_q:
            q = self.net.forward_multi([self._obs_job(params=self.params, slot=cur, batch=self.N, out=self.q_act, packed=self.params_pk)])[0]
        # action selection + environment step in one launch (dq_env_act_step == dq_policy_select then dq_env_step)
        seed = (ctypes.c_uint32 * 2)(*env.seed)
        args = (env._h, ptr(q), float(eps), int(masked_greedy), seed, int(self.vector_steps), ptr(self.action_ring[cur]), 1,
                self._obs_slot(nxt), ptr(self.reward_ring[cur]), ptr(self.terminal_ring[cur]), ptr(env.legal),
                ptr(env.lifetime), ptr(env.was_reset))
        filled = min(self.T, self.filled + 1)
        if presample and filled >= MIN_FILLED:
            self._launch_env(args, self._sample_job(self.updates + 1, nxt, filled))
        else:
            self._launch_env(args, None)
        # episode bookkeeping of this step: rides on the next update's TD launch (dq_td_update_stats) when an update follows, else
        # launched on its own
        self._stats_pending = (cur,) if record_stats else None
        if record_stats and not self.defer_stats:
            self._flush_stats()
        self.cur = nxt
        self.filled = filled
        self.vector_steps += 1

    def _sample_job(self, t, head, filled):
        """dq_sample_job for update number t, to be run on a ring with `head` / `filled`, into the look-ahead buffer (the update in
        progress may still be reading self.index)."""
        sj = _lib.SampleJob()
        sj.terminal_ring_dev, sj.n_slots, sj.head_slot, sj.filled_slots, sj.batch = ptr(self.terminal_ring), self.T, head, filled, self.batch_size
        sj.seed[0], sj.seed[1] = int(self.seed[0]) & 0xFFFFFFFF, int(self.seed[1]) & 0xFFFFFFFF
        sj.t, sj.sample_base, sj.index_dev = t, _dist.shard(self.rank, self.N, self.batch_size)[1], ptr(self._index_next)
        self._presampled = (t, head, filled)
        return sj

    def _launch_env(self, args, sj):
        if getattr(self.env, "wide", False):
            # lattices beyond d = 7 (csrc/env_big.hip): selection + step in one launch; no replay sampling rides on it -- the update draws
            # its own minibatch (_take_minibatch notices that no look-ahead draw was made)
            self._presampled = None
            check(self.L.dq_envb_act_step(*args, ptr(self.env.inexact), self._stream()))
            self._inexact_acc += self.env.inexact.sum()              # (read and reported at the next read_stats())
            return
        try:
            if sj is not None:
                check(self.L.dq_env_act_step_sample(*args, ctypes.byref(sj), self._stream()))
            else:
                check(self.L.dq_env_act_step(*args, self._stream()))
        except BaseException:
            self.env.disarm_patch_output()                           # (_obs_slot armed a ring slot for this launch)
            raise

    def _join_env(self):
        """Orders the current stream behind an environment launch still running on the side stream."""
        if self._env_inflight:
            torch.cuda.current_stream(self.device).wait_event(self._e_env)
            self._env_inflight = False

    def _flush_stats(self):
        self._join_env()
        if self._stats_pending is not None:
            (slot,), env = self._stats_pending, self.env
            check(self.L.dq_episode_stats(ptr(self.terminal_ring[slot]), ptr(env.was_reset), ptr(env.lifetime), ptr(self.reward_ring[slot]),
                                          self.N, ptr(self.stats), self._stream()))
            self._stats_pending = None

    def update(self, rows=None, target_ready=None, target_next=None):
        """One minibatch update (keras-rl DQNAgent.backward's training branch).  rows: its minibatch, already drawn (_extra_updates);
        target_ready: (Q_target(s1) of those rows, None or the event behind the forward that writes it): no target job in this update's launch pair;
        target_next: (rows of the NEXT update, buffer): its Q_target(s1) rides on this update's launch pair as a fourth job."""
        assert self.filled >= MIN_FILLED, "fewer than three complete transitions in the replay ring"
        self._join_env()
        B, N, T = self.batch_size, self.N, self.T
        self.updates += 1
        t = self.updates
        _, sample_base = _dist.shard(self.rank, N, B)
        if rows is None:
            self._take_minibatch(t, self.cur, self.filled, sample_base)
            self.last_index = self.index
            self.net.forward_multi(self._update_jobs(t, sample_base))
            self._learn(t)
            return
        self.last_index = rows                                      # (the rows of the last update: self.index keeps its own buffer for the look-ahead draws)
        own, self.index = self.index, rows                          # (every job record and the TD step read self.index)
        own_q1 = self.q1_target
        try:
            jobs = self._update_jobs(t, sample_base, with_target=target_ready is None)
            if target_next is not None:
                jobs.insert(0, self._obs_job(params=self.target, batch=B, index=target_next[0], index_off=N, index_mod=T * N, out=target_next[1],
                                             packed=self.target_pk))
            self.net.forward_multi(jobs)
            if target_ready is not None:
                self.q1_target = target_ready[0]
                if target_ready[1] is not None:
                    torch.cuda.current_stream(self.device).wait_event(target_ready[1])
            self._learn(t)
        finally:
            self.index, self.q1_target = own, own_q1

    def _extra_updates(self, k):
        """k further updates on the ring as it stands (DQNAgent.updates_per_vector_step - 1).  The first one's minibatch came with the environment
        launch; the others' are drawn by ONE launch here (dq_replay_sample_multi: the draws of consecutive updates on one ring state do not depend
        on each other) instead of a launch in front of every update -- round 5: one launch fewer per extra update."""
        if k <= 0:
            return
        self.update()
        if k == 1:
            return
        B = self.batch_size
        if self._index_multi is None or self._index_multi.shape[0] < k - 1:
            self._index_multi = torch.empty((k - 1, B), dtype=torch.int32, device=self.device)
        _, sample_base = _dist.shard(self.rank, self.N, B)
        _q.replay_sample_multi(self.terminal_ring, self.N, self.T, self.cur, self.filled, B, self.seed, self.updates + 1, k - 1,
                               sample_base=sample_base, out=self._index_multi)
        if not (self.target_ahead and self.net.fused_supported and self.net.fused_enabled):
            # The target network does not change inside a vector step and the rows of all these updates are known, so Q_target(s1) of update i + 1 needs
            # nothing update i produces: updates go in PAIRS -- the first one's launch pair carries four forwards (its own three and the second's target
            # forward), the second one's two.  At c3 the wave-private convolution kernel then runs 2 + 1 whole trips per pair of updates where three
            # forwards per launch are 1.5 trips timed as 2 each, and the dense kernel fills the chip (256 workgroups) instead of 192.
            if self.pair_targets and self.net.fused_supported and self.net.fused_enabled:
                if self._q1_pair is None:
                    self._q1_pair = torch.empty((B, self.A), dtype=torch.float32, device=self.device)
                i = 0
                while i + 1 < k - 1:
                    self.update(rows=self._index_multi[i], target_next=(self._index_multi[i + 1], self._q1_pair))
                    self.update(rows=self._index_multi[i + 1], target_ready=(self._q1_pair, None))
                    i += 2
                if i < k - 1:
                    self.update(rows=self._index_multi[i])
                return
            for i in range(k - 1):
                self.update(rows=self._index_multi[i])
            return
        # Q_target(s1) of update i + 1 on the side stream, released by a mark inside update i's backward (dq_qnet_mark_conv_backward: behind the
        # convolutional backward's launch -- the final reduction and the repacking that follow leave most of the device idle) and handed to update
        # i + 1's TD launch by an event of its own.  The first one goes out at once.
        side = self._side_setup(k - 1)
        self._side_forward(0, None)
        for i in range(k - 1):
            last = i + 1 == k - 1
            if not last:
                self.net.mark_conv_backward(side[5][i])
            try:
                self.update(rows=self._index_multi[i], target_ready=(side[4][i], side[3][i]))
            finally:
                self.net.mark_conv_backward(None)
            if not last:
                self._side_forward(i + 1, side[5][i])

    def _side_setup(self, n):
        B, net = self.batch_size, self.net
        if self._side is None:
            net2 = _q.QNetwork(net.input_shape, net.c_layers, net.ff_layers, net.n_actions, dueling=net.dueling, max_batch=net.max_batch, device=self.device)
            if self.compact:
                net2.set_patch_input(self.env.volume_depth, self.env.patch_stride)
            # [network handle with its own per-job scratch, stream, event behind the rows' draw, per-update "Q_target ready" events, Q_target(s1) rows,
            #  per-update marks inside the backward]
            self._side = [net2, torch.cuda.Stream(device=self.device), torch.cuda.Event(), [], None, []]
        side = self._side
        if side[4] is None or side[4].shape[0] < n:
            side[4] = torch.empty((n, B, self.A), dtype=torch.float32, device=self.device)
        while len(side[3]) < n:
            side[3].append(torch.cuda.Event())
            side[5].append(torch.cuda.Event())
        side[2].record(torch.cuda.current_stream(self.device))      # behind the launch that drew the rows (and behind every reader of last step's Q_target rows)
        return side

    def _side_forward(self, i, after):
        """Q_target(s1) of extra update i (rows self._index_multi[i]) on the side stream, behind `after` (an event of the main stream) or, the first one,
        behind the rows' draw."""
        net2, stream, e_rows, ready, q1, _ = self._side
        with torch.cuda.stream(stream):
            stream.wait_event(e_rows if after is None else after)
            net2.forward_multi([self._obs_job(params=self.target, batch=self.batch_size, index=self._index_multi[i], index_off=self.N,
                                              index_mod=self.T * self.N, out=q1[i], packed=self.target_pk)])
            ready[i].record(stream)

    def _take_minibatch(self, t, head, filled, sample_base):
        """self.index <- rows of update t on a ring with `head` / `filled`: the look-ahead draw if an environment launch made exactly
        that one, else a launch of its own."""
        if self._presampled == (t, head, filled):
            self.index, self._index_next = self._index_next, self.index
        else:
            _q.replay_sample(self.terminal_ring, self.N, self.T, head, filled, self.batch_size, self.seed, t, sample_base=sample_base,
                             out=self.index)
        self._presampled = None

    def _update_jobs(self, t, sample_base, with_target=True):
        # Q_online(s1) picks the action, Q_target(s1) values it (double DQN; without it Q_target does both); the training forward
        # on s0 is independent of both, so the three share one pair of launches
        B, N = self.batch_size, self.N
        rows = self.T * N
        jobs = []
        if with_target:
            jobs.append(self._obs_job(params=self.target, batch=B, index=self.index, index_off=N, index_mod=rows, out=self.q1_target, packed=self.target_pk))
        if self.enable_double_dqn:
            jobs.append(self._obs_job(params=self.params, batch=B, index=self.index, index_off=N, index_mod=rows, out=self.q1_online,
                                      packed=self.params_pk))
        jobs.append(self._obs_job(params=self.params, batch=B, index=self.index, training=True, seed=self.seed, t=t, sample_base=sample_base,
                                  out=self.q0, packed=self.params_pk))
        return jobs

    def _td_job(self, step_stats=None):
        q_sel = self.q1_online if self.enable_double_dqn else self.q1_target
        return dict(q_online_s1=q_sel, q_target_s1=self.q1_target, q_s0=self.q0, reward=self.reward_ring, terminal=self.terminal_ring,
                    action=self.action_ring, gamma=self.gamma, grad_scale=_dist.grad_scale(self.batch_size, self.world_size),
                    index=self.index, y=self.y, dq=self.dq, metrics=self.metrics, step_stats=step_stats, auto_scale=self.auto_scale)

    def local_gradient(self, index=None):
        """This rank's contribution to the NEXT update's gradient, without the all-reduce and without the optimizer step: minibatch
        draw (or the given ring rows), the three forwards, TD step, backward; scaled by 1 / (batch_size * world_size), so that the SUM
        over the ranks is the gradient of the global-minibatch mean loss.  Leaves counters and parameters untouched (tests, diagnostics:
        tests/test_agent_gpu.py compares the sum of eight shard gradients with the gradient of one eight-times-larger lattice batch)."""
        assert self.filled >= MIN_FILLED
        self._join_env()
        t = self.updates + 1
        _, sample_base = _dist.shard(self.rank, self.N, self.batch_size)
        if index is None:
            _q.replay_sample(self.terminal_ring, self.N, self.T, self.cur, self.filled, self.batch_size, self.seed, t, sample_base=sample_base,
                             out=self.index)
        else:
            self.index.copy_(index)
        self._presampled = None
        self.net.forward_multi(self._update_jobs(t, sample_base))
        self.net.td_backward_phase0(self.params, self._td_job(), self.grads)
        self.net.backward_phase(self.params, self.dq, self.grads, 1)
        self._metrics_stale = True
        return self.grads

    def _learn(self, t, ride=None):
        """TD step, backward, optimizer step, repack -- everything of an update behind its forwards.  ride: the vector step's
        environment launch (qnet._env_step_job dict) to be carried by the dense backward's first kernel."""
        B, N, net = self.batch_size, self.N, self.net
        step_stats = None
        if self._stats_pending is not None:          # the pending episode bookkeeping rides on the TD launch
            assert ride is None
            (slot,), env = self._stats_pending, self.env
            step_stats = (self.terminal_ring[slot], env.was_reset, env.lifetime, self.reward_ring[slot], N, self.stats)
            self._stats_pending = None
        td = self._td_job(step_stats)
        self._metrics_stale = True
        if _dist.dist_path(self.world_size) and os.environ.get("DQ_DIST_MODE", "single") == "single":
            # ONE all-reduce of the whole flat gradient behind the backward, on THIS stream through the learner's own RCCL communicator
            # (dist.RcclComm; through torch.distributed for gloo groups): the communicator-stream hand-offs of the split form below cost
            # more on a one-rank measurement than the 0.7 MB they hide
            if ride is not None:                          # the whole backward, gradient only (m = v = None): no optimizer step on the reduction
                net.td_backward_adam_env(self.params, td, self.grads, None, None, t, self.lr, self.beta_1, self.beta_2, self.epsilon, self.env._h, ride)
            else:
                net.td_backward_adam(self.params, td, self.grads, None, None, t, self.lr, self.beta_1, self.beta_2, self.epsilon)
            probe = self.ar_events is not None and self._ar_seen % self.ar_stride == 0
            self._ar_seen += 1
            if probe:
                e0, e1 = self.ar_pool.pop() if self.ar_pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                e0.record()
            self.ensure_comm()
            if self._rccl is not None:
                self._rccl.allreduce_sum_(self.grads)
            else:
                _dist.allreduce_sum_(self.grads, group=self.pg)
            if probe:
                e1.record()
                self.ar_events.append((e0, e1))
            net.adam_step(self.params, self.grads, self.m, self.v, t, self.lr, self.beta_1, self.beta_2, self.epsilon)     # (skips AND flags non-finite elements: every rank alike)
        elif _dist.dist_path(self.world_size) and os.environ.get("DQ_DIST_MODE") == "overlap" and self._comm_ready():
            # DQ_DIST_MODE=overlap: the dense layers' gradient (0.7 of the 0.77 MB at c3) is all-reduced on a SECOND stream through the learner's own
            # RCCL communicator while the convolutional backward (32 us) runs on the step's stream -- one event each way, no torch.distributed
            # hand-off --; the convolutional range (66 KB) follows in-stream, then the guarded Adam.  The same communicator serves both streams: the
            # two collectives are issued in the same order on every rank.  What it costs beside the default: the backward in two phases (two final
            # reductions instead of one) and the two events; what it hides: the wire time of the dense range.
            nconv = net.n_conv_params
            if ride is not None:
                net.td_backward_phase0_env(self.params, td, self.grads, self.env._h, ride)
            else:
                net.td_backward_phase0(self.params, td, self.grads)
            main = torch.cuda.current_stream(self.device)
            if self._ar_stream is None:
                self._ar_stream = torch.cuda.Stream(device=self.device)
                self._e_dense, self._e_ar = torch.cuda.Event(), torch.cuda.Event()
            self._e_dense.record(main)
            with torch.cuda.stream(self._ar_stream):
                self._ar_stream.wait_event(self._e_dense)
                (self._rccl2 or self._rccl).allreduce_sum_(self.grads[nconv:])      # (the side stream's OWN communicator: two collectives of one communicator
                self._e_ar.record(self._ar_stream)                                 #  in flight on two streams rest on RCCL's implicit serialisation)
            net.backward_phase(self.params, self.dq, self.grads, 1)
            probe = self.ar_events is not None and self._ar_seen % self.ar_stride == 0
            self._ar_seen += 1
            if probe:
                e0, e1 = self.ar_pool.pop() if self.ar_pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                e0.record()
            self._rccl.allreduce_sum_(self.grads[:nconv])
            main.wait_event(self._e_ar)
            if probe:
                e1.record()
                self.ar_events.append((e0, e1))
            net.adam_step(self.params, self.grads, self.m, self.v, t, self.lr, self.beta_1, self.beta_2, self.epsilon)
        elif _dist.dist_path(self.world_size):
            # the dense layers' gradient (most of the bytes) is all-reduced while the convolutional backward runs (DQ_DIST_MODE=split; =overlap on
            # a process group without a RCCL communicator of our own: gloo)
            nconv = net.n_conv_params
            if ride is not None:
                net.td_backward_phase0_env(self.params, td, self.grads, self.env._h, ride)
            else:
                net.td_backward_phase0(self.params, td, self.grads)       # TD step + dueling + dense layers
            work = _dist.allreduce_sum_async(self.grads[nconv:], group=self.pg)
            net.backward_phase(self.params, self.dq, self.grads, 1)
            # what the step WAITS for: the convolutional range's all-reduce (critical path) + whatever is left of the dense range's
            probe = self.ar_events is not None and self._ar_seen % self.ar_stride == 0
            self._ar_seen += 1
            if probe:
                e0, e1 = self.ar_pool.pop() if self.ar_pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                e0.record()
            _dist.allreduce_sum_(self.grads[:nconv], group=self.pg)
            if work is not None:
                work.wait()
            if probe:
                e1.record()
                self.ar_events.append((e0, e1))
            net.adam_step(self.params, self.grads, self.m, self.v, t, self.lr, self.beta_1, self.beta_2, self.epsilon)     # (skips AND flags non-finite elements: every rank alike)
        elif ride is not None:
            net.td_backward_adam_env(self.params, td, self.grads, self.m, self.v, t, self.lr, self.beta_1, self.beta_2, self.epsilon,
                                     self.env._h, ride)
        else:
            # TD step in the backward's first launch, Adam on its last
            net.td_backward_adam(self.params, td, self.grads, self.m, self.v, t, self.lr, self.beta_1, self.beta_2, self.epsilon)
        self.repack()

    def step_and_update(self, eps, masked_greedy=False, record_stats=True, presample_next=True, extra_updates=0):
        """act_and_step() followed by update(), with the acting forward and the update's forwards in ONE pair of launches -- same
        results as the two calls.  extra_updates: that many further update() calls on the same ring state (DQNAgent's
        updates_per_vector_step - 1: the reference trains one 32-sample minibatch per environment step, TRAIN:119-127, i.e. N minibatches
        per vector step of N lattices).  Possible because the update's minibatch never contains the newest transition (keras-rl's range:
        its successor observation is not in the memory yet), so it does not depend on this step's environment results -- the newest
        row it can hold is the previous step's, whose successor is the observation this step acts on: the parameters are the same
        for all four forwards.
        With presample_next the environment launch also draws the NEXT step's minibatch (if that step does not update, update()
        notices the stale draw and redraws)."""
        self._flush_stats()
        env, cur, B, N, T = self.env, self.cur, self.batch_size, self.N, self.T
        nxt = cur + 1 if cur + 1 < T else 0
        filled = min(T, self.filled + 1)
        assert filled >= MIN_FILLED, "fewer than three complete transitions in the replay ring"
        t = self.updates + 1
        _, sample_base = _dist.shard(self.rank, N, B)
        self._take_minibatch(t, nxt, filled, sample_base)        # (drawn from the ring as it WILL be after this step)
        self.last_index = self.index
        jobs = self._update_jobs(t, sample_base)
        jobs.append(self._obs_job(params=self.params, slot=cur, batch=N, out=self.q_act, packed=self.params_pk))
        self.net.forward_multi(jobs)
        seed = (ctypes.c_uint32 * 2)(*env.seed)
        sj = None
        if extra_updates > 0:                    # the next update runs on THIS step's ring: the environment launch draws its minibatch
            sj = self._sample_job(t + 1, nxt, filled)
        elif presample_next:
            nxt2 = nxt + 1 if nxt + 1 < T else 0
            sj = self._sample_job(t + 1, nxt2, min(T, filled + 1))
        if self.ride_env and self._env_stream is None and self.net.fused_supported and self.net.fused_enabled and not getattr(env, "wide", False) \
                and not getattr(env, "mlp_referee", False):
            # one launch fewer per step: the environment step (+ look-ahead sampling + this step's episode bookkeeping) rides on the
            # dense backward's first kernel (dq_qnet_td_backward_adam_env / _phase0_env): neither needs the other's results
            if self.compact:
                env.arm_patch_output(self.patch_ring[nxt])           # (consumed by the riding step's launch)
            step = dict(q=self.q_act, eps=eps, masked_greedy=masked_greedy, seed=env.seed, t=self.vector_steps, action=self.action_ring[cur],
                        auto_reset=1, obs=None if self.compact else self._obs_ring[nxt], reward=self.reward_ring[cur], done=self.terminal_ring[cur],
                        legal=env.legal, lifetime=env.lifetime, was_reset=env.was_reset, sample=sj, stats=self.stats if record_stats else None)
            self._stats_pending = None
            self.cur, self.filled = nxt, filled
            self.vector_steps += 1
            self.updates = t
            try:
                self._learn(t, ride=step)
            except BaseException:
                env.disarm_patch_output()                           # (an error in front of the riding launch: no armed ring slot outlives the call)
                raise
            self._extra_updates(extra_updates)
            return
        args = (env._h, ptr(self.q_act), float(eps), int(masked_greedy), seed, int(self.vector_steps), ptr(self.action_ring[cur]), 1,
                self._obs_slot(nxt), ptr(self.reward_ring[cur]), ptr(self.terminal_ring[cur]), ptr(env.legal),
                ptr(env.lifetime), ptr(env.was_reset))
        if self._env_stream is not None:
            main = torch.cuda.current_stream(self.device)
            self._e_fwd.record(main)
            with torch.cuda.stream(self._env_stream):
                self._env_stream.wait_event(self._e_fwd)     # needs the acting forward's Q-values (and everything before it)
                self._launch_env(args, sj)
                if record_stats:                             # the bookkeeping needs this step's results: it stays with the environment
                    check(self.L.dq_episode_stats(ptr(self.terminal_ring[cur]), ptr(env.was_reset), ptr(env.lifetime), ptr(self.reward_ring[cur]),
                                                  N, ptr(self.stats), self._stream()))
                self._e_env.record(self._env_stream)
            self._env_inflight = True
            self._stats_pending = None
        else:
            self._launch_env(args, sj)
            self._stats_pending = (cur,) if record_stats else None
        self.cur, self.filled = nxt, filled
        self.vector_steps += 1
        self.updates = t
        self._learn(t)
        self._extra_updates(extra_updates)

    def read_metrics(self):
        """(loss, mean_q) of the last update on this rank; reduces the per-block partials first (syncs)."""
        if self._metrics_stale:
            _q.td_metrics(self.metrics, self.batch_size)
            self._metrics_stale = False
        m = self.metrics[:2].cpu().numpy()
        try:
            self.net.check_range()   # (already synchronised) a gradient beyond the fused backward's range is never silent ...
        except _lib.DeepQError as e:
            # ... and with the default settings it is not fatal either: an update whose TD step met such a sample was discarded WHOLE on every rank (no
            # parameter moved; `updates` and Adam's t advanced past it -- the reference, fp32 with delta_clip = inf, would have applied it: the count is
            # kept in self.discarded_updates and logged), and from here on the gradient scale is MEASURED per minibatch (dq_td_job.auto_scale, +3 us per
            # update), which carries any finite TD error.  Sticky.  DQ_TD_AUTOSCALE=0 keeps the host-known scale and raises; DQ_TD_AUTOSCALE=1 measures
            # from the first update (nothing is ever discarded).
            n_lost = self.net.range_discarded()
            self.discarded_updates += n_lost
            # (the FORWARD's guard -- a parameter or activation outside the f16 pieces' range: a diverged run -- is always fatal: no gradient scale helps)
            if e.status != -6 or "[forward]" in str(e) or self.auto_scale or os.environ.get("DQ_TD_AUTOSCALE") is not None:
                raise
            self.auto_scale = True
            self.range_switches = getattr(self, "range_switches", 0) + 1
            import warnings
            warnings.warn(f"a TD error beyond the fused backward's host-known gradient scale: {n_lost} of the {self.updates - self._range_ok_at} updates since "
                          f"the last synchronisation were discarded (whole, on every rank; {self.discarded_updates} in this run: the reference would have applied "
                          "them); the loop now measures the gradient scale per minibatch (DQNCore.auto_scale = True) -- set DQ_TD_AUTOSCALE=1 to start that way")
        self._range_ok_at = self.updates
        return float(m[0]), float(m[1])

    def repack(self):
        """Must follow every change of self.params (Adam step, weight loading): refreshes the packed conv kernels."""
        if self.params_pk is not None:
            self.net.pack(self.params, out=self.params_pk)

    def update_target_hard(self):
        self.target.copy_(self.params)
        if self.target_pk is not None:
            self.target_pk.copy_(self.params_pk)

    def ensure_comm(self, with_overlap=False):
        """Creates the learner's own RCCL communicator if the several-GPU branch will use one (dist.make_rccl: a collective call -- every rank, at the
        same point; None for gloo groups / DQ_DIST_NATIVE=0).  Called by the first several-GPU update; bench.py calls it in front of its warm-up
        steps so that the rendezvous (~1 s) can never fall into a timed region.  with_overlap: also the side stream's communicator, whatever DQ_DIST_MODE
        says now (bench.py times both forms of the exchange and runs the faster: every rank calls this alike)."""
        if self._rccl is None and not self._rccl_tried and _dist.dist_path(self.world_size):
            self._rccl_tried = True
            self._rccl = _dist.make_rccl(self.rank, self.world_size, self.device, self.pg)
        # DQ_DIST_MODE=overlap: a second communicator for the dense range's all-reduce on the side stream (the same collective call on every rank)
        if self._rccl is not None and self._rccl2 is None and (with_overlap or os.environ.get("DQ_DIST_MODE") == "overlap"):
            self._rccl2 = _dist.make_rccl(self.rank, self.world_size, self.device, self.pg)

    def _comm_ready(self):
        self.ensure_comm()
        return self._rccl is not None

    def close_comm(self, abort=False):
        """Destroys the learner's own RCCL communicator (dist.RcclComm) -- every rank, behind a synchronisation, BEFORE the process group
        that did its rendezvous is destroyed; a later several-GPU update would create a new one.  abort: the caller is leaving through an exception
        (DQNAgent.fit's finally): ncclCommAbort instead of ncclCommDestroy, which could wait for peers inside a collective this rank never joins."""
        if self._rccl is not None:
            if abort:
                # the abort path exists for a collective this rank has queued and its peers will never join: a device synchronisation IN FRONT of
                # ncclCommAbort would wait for exactly that collective, for ever.  Abort first; then drain what is left, and never let a (sticky) HIP
                # error raised by that drain replace the exception that is unwinding through fit()
                try:
                    if self._rccl2 is not None:
                        self._rccl2.close(abort=True)
                    self._rccl.close(abort=True)
                finally:
                    self._rccl = self._rccl2 = None
                    try:
                        torch.cuda.synchronize(self.device)
                    except Exception:
                        pass
            else:
                try:
                    torch.cuda.synchronize(self.device)
                finally:
                    if self._rccl2 is not None:
                        self._rccl2.close(abort=False)
                    self._rccl.close(abort=False)
                    self._rccl = None
        self._rccl, self._rccl2, self._rccl_tried = None, None, False

    def close(self):
        """Releases what outlives the Python objects: the learner's RCCL communicator (before torch.distributed.destroy_process_group)."""
        self.close_comm()

    def __del__(self):
        try:
            self.close_comm()
        except Exception:
            pass

    def read_stats(self, reset=True, all_ranks=False):
        """(episodes ended, sum of their lifetimes, rewards earned, lattices stepped) since the last reset; syncs.  all_ranks: summed
        over the process group, so that every rank takes the same decisions from them (early stopping in DQNAgent.fit: a rank that left
        the loop alone would leave the others waiting in the gradient all-reduce)."""
        self._flush_stats()
        st = self.stats
        self.local_stats = None
        if all_ranks and self.world_size > 1:
            self.local_stats = [int(x) for x in st.cpu().tolist()]       # this rank's own share (DQNAgent.fit: the single-lattice episode end)
            st = st.clone()
            _dist.allreduce_sum_(st, group=self.pg)
        s = [int(x) for x in st.cpu().tolist()]
        if self.local_stats is None:
            self.local_stats = list(s)
        if reset:
            self.stats.zero_()
        if self._inexact_acc is not None:
            # wide environment: lattice-steps whose reward / done came from the matching referee's FALLBACK (more than 14 defects in a
            # component: the rest go to their nearer boundary, csrc/match_dev.h) -- heuristic referee decisions must not enter the
            # replay memory or the lifetime statistics silently
            n = int(self._inexact_acc.item())
            if n:
                self.inexact_total += n
                self._inexact_acc.zero_()
                import warnings
                warnings.warn(f"{n} lattice-steps since the last synchronisation were refereed by the matching decoder's inexact fallback "
                              f"(a cluster of more than 20 defects, or more than 32 defects in one component; {self.inexact_total} in total): their reward / done are heuristic")
        return s

    # -- evaluation on a scratch ring ---------------------------------------------------------------------------------------
    def begin_eval(self):
        """keras-rl stores nothing in test mode (memory.append(training=False) is a no-op): greedy evaluation steps run on a scratch
        three-slot ring, so that fit -> test -> fit (or pickling the memory after test()) never trains on evaluation transitions."""
        assert getattr(self, "_train_ring", None) is None
        self._flush_stats()
        self._join_env()
        self.env.disarm_patch_output()                               # (nothing armed for the training ring survives the switch of rings)
        self._train_ring = (self.patch_ring if self.compact else self._obs_ring, self.action_ring, self.reward_ring, self.terminal_ring, self.T,
                            self.cur, self.filled, self._presampled, self.started)
        T = 3
        dev = self.device
        if self.compact:
            self.patch_ring = torch.zeros((T,) + tuple(self.patch_ring.shape[1:]), dtype=torch.int32, device=dev)
        else:
            self._obs_ring = torch.zeros((T,) + tuple(self._obs_ring.shape[1:]), dtype=torch.uint8, device=dev)
        self.action_ring = torch.zeros((T, self.N), dtype=torch.int32, device=dev)
        self.reward_ring = torch.zeros((T, self.N), dtype=torch.float32, device=dev)
        self.terminal_ring = torch.zeros((T, self.N), dtype=torch.uint8, device=dev)
        self.T, self.cur, self.filled, self._presampled = T, 0, 0, None

    def end_eval(self):
        self._flush_stats()
        self._join_env()
        self.env.disarm_patch_output()
        (ring, self.action_ring, self.reward_ring, self.terminal_ring, self.T, self.cur, self.filled, self._presampled,
         self.started) = self._train_ring
        if self.compact:
            self.patch_ring = ring
        else:
            self._obs_ring = ring
        self._train_ring = None
        # the lattices were reset and stepped by the evaluation: the ring's newest observation no longer describes them, so the next
        # fit() (which calls reset_env(): previous entry marked terminal, fresh observation into slot `cur`) must not skip its reset
        self.started = False
