"""keras-rl-compatible agent surface over the device loop (DQNCore).

Mirrors what the reference's driver scripts and notebooks use from the un-vendored keras-rl fork
(github.com/R-Sweke/keras-rl; call sites in
/root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py, "TRAIN"):

    SequentialMemory(limit, window_length=1)                                   TRAIN:109
    LinearAnnealedPolicy(EpsGreedyQPolicy(masked_greedy=..), attr='eps', ...)   TRAIN:110-114
    GreedyQPolicy(masked_greedy=True)                                           TRAIN:115
    DQNAgent(model, nb_actions, memory, nb_steps_warmup, target_model_update,
             policy, test_policy, gamma, enable_dueling_network)                TRAIN:119-127
    dqn.compile(Adam(lr=...))                                                   TRAIN:130
    dqn.fit(env, nb_steps, ..., episode_averaging_length, success_threshold,
            stopping_patience, min_nb_steps, single_cycle)                      TRAIN:138-152
    dqn.test(env, nb_episodes, visualize, verbose, interval, single_cycle)      TRAIN:206
    dqn.save_weights / dqn.model.load_weights / dqn.memory / dqn.forward        TRAIN:157-160,183; notebook 3 cell 24
    FileLogger(filepath, interval)                                              TRAIN:94-95

Semantics follow upstream keras-rl 0.4.x where the fork's source is unavailable; the
fork-only keywords (episode_averaging_length, success_threshold, stopping_patience, min_nb_steps, single_cycle,
masked_greedy) follow the README's description and log output (README.md:168,262,408-480,638-672).

`env` may be the drop-in single-lattice class or a VectorEnv of N lattices; both run the same device loop.
With N lattices one "step" of the loop advances all of them, `self.step` advances by N, and episode statistics
are gathered on the device and read back every `sync_interval` vector steps.
"""
import json
import os
import time
import timeit
import warnings
from collections import deque

import numpy as np
import torch

from .core import DQNCore, MIN_FILLED
from .env import Surface_Code_Environment_Multi_Decoding_Cycles, VectorEnv
from .qnet import QNetwork


# ----------------------------------------------------------------------------------------------------------
# model description (stands in for the keras Sequential returned by build_convolutional_nn)
# ----------------------------------------------------------------------------------------------------------
class ConvQModel:
    """Layer description + Keras-ordered weight list.  Bound to a device QNetwork by the agent."""

    def __init__(self, cc_layers, ff_layers, input_shape, num_actions):
        self.c_layers = [[int(x) for x in l] for l in cc_layers]
        self.ff_layers = [[int(l[0]), float(l[1])] for l in ff_layers]
        self.input_shape = tuple(int(x) for x in input_shape)
        self.num_actions = int(num_actions)
        self._weights = None          # host copy (list of numpy arrays) until bound
        self._agent = None

    @property
    def output_shape(self):
        return (None, self.num_actions)

    def get_weights(self):
        if self._agent is not None and self._agent._core is not None:
            return self._agent._net.get_weights(self._agent._core.params)
        return None if self._weights is None else [w.copy() for w in self._weights]

    def set_weights(self, weights):
        weights = [np.asarray(w, dtype=np.float32) for w in weights]
        if self._agent is not None and self._agent._core is not None:
            self._agent._net.set_weights(self._agent._core.params, weights)
            self._agent._core.repack()
            self._agent._core.update_target_hard()
        self._weights = weights

    def save_weights(self, filepath, overwrite=True):
        if os.path.exists(filepath) and not overwrite:
            raise IOError(f"{filepath} exists and overwrite=False")
        w = self.get_weights()
        if w is None:
            raise RuntimeError("model has no weights yet (fit/compile against an environment first)")
        from .weights_io import save_weights_file
        dueling = len(w) // 2 == len(self.c_layers) + len(self.ff_layers) + 2      # Dense(|A|) + keras-rl's Dense(|A|+1)
        save_weights_file(filepath, w, self.layer_names(dueling), dueling=dueling)

    def load_weights(self, filepath):
        from .weights_io import load_weights_file
        self.set_weights(load_weights_file(filepath))

    def layer_names(self, dueling=True):
        names = [f"conv2d_{i + 1}" for i in range(len(self.c_layers))]
        names += [f"dense_{i + 1}" for i in range(len(self.ff_layers) + 1 + (1 if dueling else 0))]
        return names

    def summary(self):
        print(f"ConvQModel input={self.input_shape} conv={self.c_layers} dense={self.ff_layers} -> {self.num_actions} actions")


def build_convolutional_nn(cc_layers, ff_layers, input_shape, num_actions):
    """Function_Library.py:338-377 / TRAIN:61-90: returns the model description the agent builds on the GPU."""
    return ConvQModel(cc_layers, ff_layers, input_shape, num_actions)


class Adam:
    """keras.optimizers.Adam stand-in (TRAIN:130): hyper-parameters only; the update is dq_adam_step."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0.0, amsgrad=False, **kwargs):
        if decay or amsgrad:
            raise NotImplementedError("Adam(decay/amsgrad) is not used by the reference and not implemented")
        self.lr, self.beta_1, self.beta_2 = float(lr), float(beta_1), float(beta_2)
        self.epsilon = 1e-7 if epsilon is None else float(epsilon)     # K.epsilon()


# ----------------------------------------------------------------------------------------------------------
# memory and policies
# ----------------------------------------------------------------------------------------------------------
class SequentialMemory:
    """Capacity holder; the storage is the device replay ring of DQNCore (see core.py)."""

    def __init__(self, limit, window_length=1, ignore_episode_boundaries=False):
        if window_length != 1:
            raise NotImplementedError("window_length != 1 is not used by the reference and not implemented")
        self.limit, self.window_length = int(limit), 1
        self._core = None
        self._saved = None

    @property
    def nb_entries(self):
        if self._core is not None:
            return min(self._core.filled, self._core.T) * self._core.N
        if self._saved is not None:                 # unpickled, not bound to an agent yet
            T, N = self._saved["obs_shape"][:2]
            return min(int(self._saved["filled"]), T) * N
        return 0

    def get_config(self):
        return dict(limit=self.limit, window_length=1)

    # pickling (TRAIN:156-157 pickles dqn.memory; the Continue script reloads it)
    def __getstate__(self):
        st = dict(limit=self.limit, window_length=1, _saved=None)
        c = self._core
        if c is not None:
            st["_saved"] = dict(obs_shape=tuple(c.obs_ring.shape), action=c.action_ring.cpu().numpy(), reward=c.reward_ring.cpu().numpy(),
                                terminal=c.terminal_ring.cpu().numpy(), cur=c.cur, filled=c.filled)
            if getattr(c, "compact", False):     # the ring holds patch words (core.py): pickled as they are, d * d words per observation
                st["_saved"]["patch"] = c.patch_ring.cpu().numpy()
            else:
                st["_saved"]["obs"] = np.packbits(c.obs_ring.cpu().numpy(), axis=None)
        elif self._saved is not None:
            st["_saved"] = self._saved
        return st

    def __setstate__(self, st):
        self.limit, self.window_length, self._saved, self._core = st["limit"], 1, st.get("_saved"), None

    def _restore_into(self, core):
        s = self._saved
        if s is None or tuple(s["obs_shape"]) != tuple(core.obs_ring.shape):
            return False
        if "patch" in s:                            # pickled from a compact ring
            patch = torch.from_numpy(s["patch"]).to(core.device)
            if getattr(core, "compact", False) and tuple(patch.shape) == tuple(core.patch_ring.shape):
                core.patch_ring.copy_(patch)
            else:
                core.obs_ring.copy_(core.env.patch_to_obs(patch))
        else:
            n = int(np.prod(s["obs_shape"]))
            core.obs_ring.copy_(torch.from_numpy(np.unpackbits(s["obs"], count=n).reshape(s["obs_shape"])))
        core.action_ring.copy_(torch.from_numpy(s["action"]))
        core.reward_ring.copy_(torch.from_numpy(s["reward"]))
        core.terminal_ring.copy_(torch.from_numpy(s["terminal"]))
        core.cur, core.filled = int(s["cur"]), int(s["filled"])
        return True


class Policy:
    def _set_agent(self, agent):
        self.agent = agent

    @property
    def metrics_names(self):
        return []

    @property
    def metrics(self):
        return []


class EpsGreedyQPolicy(Policy):
    """With probability eps a uniformly random LEGAL action, else argmax Q (over the legal set when masked_greedy)."""

    def __init__(self, eps=.1, masked_greedy=False):
        self.eps, self.masked_greedy = eps, masked_greedy

    def current(self, training=True):
        return float(self.eps), bool(self.masked_greedy)


class GreedyQPolicy(Policy):
    def __init__(self, masked_greedy=False):
        self.masked_greedy = masked_greedy

    def current(self, training=True):
        return 0.0, bool(self.masked_greedy)


class BoltzmannQPolicy(Policy):
    def __init__(self, *a, **k):
        raise NotImplementedError("BoltzmannQPolicy is imported but never used by the reference (TRAIN:16)")


class LinearAnnealedPolicy(Policy):
    """keras-rl LinearAnnealedPolicy: linearly anneals inner_policy.<attr> with the agent's step counter."""

    def __init__(self, inner_policy, attr, value_max, value_min, value_test, nb_steps):
        if not hasattr(inner_policy, attr):
            raise ValueError(f'Policy does not have attribute "{attr}".')
        self.inner_policy, self.attr = inner_policy, attr
        self.value_max, self.value_min, self.value_test, self.nb_steps = value_max, value_min, value_test, nb_steps

    def get_current_value(self, training=True):
        if training:
            a = -float(self.value_max - self.value_min) / float(self.nb_steps)
            return max(self.value_min, a * float(self.agent.step) + float(self.value_max))
        return self.value_test

    def current(self, training=True):
        setattr(self.inner_policy, self.attr, self.get_current_value(training))
        return self.inner_policy.current(training)

    @property
    def metrics_names(self):
        return [f"mean_{self.attr}"]


# ----------------------------------------------------------------------------------------------------------
# callbacks / history
# ----------------------------------------------------------------------------------------------------------
class History:
    def __init__(self):
        self.history = {}

    def append(self, logs):
        for k, v in logs.items():
            self.history.setdefault(k, []).append(v)


class Callback:
    def on_train_begin(self, logs=None): pass
    def on_train_end(self, logs=None): pass
    def on_episode_end(self, episode, logs=None): pass


class FileLogger(Callback):
    """keras-rl FileLogger: per-episode metrics dumped as one JSON dict of lists every `interval` episodes."""

    def __init__(self, filepath, interval=None):
        self.filepath, self.interval = filepath, interval
        self.data = {}

    def on_episode_end(self, episode, logs=None):
        logs = dict(logs or {})
        duration = logs.pop("duration", None)
        logs["episode"] = episode                       # keras-rl appends ('episode', ...) and ('duration', ...) behind the agent's log
        if duration is not None:
            logs["duration"] = duration
        for k, v in logs.items():
            self.data.setdefault(k, []).append(v)
        if self.interval is not None and episode % self.interval == 0:
            self.save_data()

    def on_train_end(self, logs=None):
        self.save_data()

    def save_data(self):
        if not self.data.get("episode"):
            return
        order = np.argsort(self.data["episode"])
        out = {k: [_jsonable(v[i]) for i in order] for k, v in self.data.items()}
        with open(self.filepath, "w") as f:
            json.dump(out, f)


def _jsonable(v):
    if isinstance(v, (np.floating, np.integer)):
        v = v.item()
    if isinstance(v, float) and v != v:
        return float("nan")
    return v


# ----------------------------------------------------------------------------------------------------------
# the agent
# ----------------------------------------------------------------------------------------------------------
def stopping_flags(has_succeeded, stopped_improving, rolling, time_since_best, steps_done, success_threshold, stopping_patience, min_nb_steps):
    """The fork's early-stopping flags after an episode (fit keywords of TRAIN:138-152; the fork itself is not in the tree).  Pinned by the
    fourteen training_history.json files the reference ships (tests/golden/training_history_tails.npz, tests/test_host_logic.py): `stopped_improving`
    turns True at the first episode whose time_since_best EXCEEDS stopping_patience (1001 with patience 1000) once min_nb_steps steps have been
    taken -- and that episode is the run's last; before min_nb_steps the counter runs far past the patience without effect (9426 in
    d5_dp/0.001) and sets nothing.  `has_succeeded` (never True in any shipped record): rolling average >= success_threshold, as README.md:408-480
    describes; the run ends once either flag is up and min_nb_steps have passed."""
    if success_threshold is not None and rolling >= success_threshold:
        has_succeeded = True
    if stopping_patience is not None and time_since_best > stopping_patience and steps_done >= min_nb_steps:
        stopped_improving = True
    return has_succeeded, stopped_improving


class DQNAgent:
    def __init__(self, model, nb_actions, memory, nb_steps_warmup=1000, target_model_update=10000, policy=None, test_policy=None,
                 gamma=.99, enable_dueling_network=False, enable_double_dqn=True, dueling_type='avg', batch_size=32,
                 train_interval=1, memory_interval=1, delta_clip=np.inf, custom_model_objects=None, seed=None, updates_per_vector_step=1,
                 **kwargs):
        """updates_per_vector_step (no keras-rl counterpart): minibatch updates per vector step of the N lattices.  keras-rl trains one
        `batch_size` minibatch per ENVIRONMENT step (train_interval=1, TRAIN:119-127), i.e. batch_size samples per step; a vector step
        advances N lattices, so the reference's replay ratio is updates_per_vector_step * batch_size / N = batch_size, reached with
        updates_per_vector_step = N (batch_size 32) or N * 32 / batch_size for a larger minibatch.  Default 1: one update per vector step."""
        if hasattr(model, "_built"):                # a keras.models.Sequential stand-in (dropin/keras): resolve to the model description
            model = model._built()
        if model.output_shape != (None, nb_actions):
            raise ValueError(f'Model output "{model.output_shape}" has invalid shape. DQN expects a model that has one dimension for each action, in this case {nb_actions}.')
        if dueling_type != 'avg':
            raise NotImplementedError("only dueling_type='avg' (the keras-rl default the reference uses) is implemented")
        if not np.isinf(delta_clip):
            raise NotImplementedError("delta_clip != inf is not used by the reference and not implemented")
        if target_model_update < 1:
            raise NotImplementedError("soft target updates (target_model_update < 1) are not used by the reference")
        if memory_interval != 1:
            raise NotImplementedError("memory_interval != 1 is not implemented")
        self.model, self.nb_actions, self.memory = model, int(nb_actions), memory
        self.nb_steps_warmup, self.target_model_update = int(nb_steps_warmup), int(target_model_update)
        self.policy = policy if policy is not None else EpsGreedyQPolicy()
        self.test_policy = test_policy if test_policy is not None else GreedyQPolicy()
        self.policy._set_agent(self)
        self.test_policy._set_agent(self)
        self.gamma, self.enable_dueling_network, self.enable_double_dqn = gamma, bool(enable_dueling_network), bool(enable_double_dqn)
        self.batch_size, self.train_interval = int(batch_size), int(train_interval)
        self.updates_per_vector_step = int(updates_per_vector_step)
        if self.updates_per_vector_step < 1:
            raise ValueError("updates_per_vector_step must be at least 1")
        self.seed = seed
        self.optimizer = None
        self.compiled = False
        self.training = False
        self.step = 0
        self._core = self._net = self._env = None
        self._last_target_sync = 0
        model._agent = self

    # -- keras-rl surface ---------------------------------------------------------------------------------------
    def compile(self, optimizer, metrics=[]):
        self.optimizer = optimizer
        self.compiled = True

    @property
    def metrics_names(self):
        return ["loss", "mean_q"] + self.policy.metrics_names

    def reset_states(self):
        pass

    def save_weights(self, filepath, overwrite=True):
        self.model.save_weights(filepath, overwrite=overwrite)

    def load_weights(self, filepath):
        self.model.load_weights(filepath)

    def get_config(self):
        return dict(nb_actions=self.nb_actions, gamma=self.gamma, batch_size=self.batch_size, nb_steps_warmup=self.nb_steps_warmup,
                    train_interval=self.train_interval, target_model_update=self.target_model_update,
                    enable_double_dqn=self.enable_double_dqn, enable_dueling_network=self.enable_dueling_network, dueling_type='avg',
                    updates_per_vector_step=self.updates_per_vector_step)

    # -- binding to an environment ------------------------------------------------------------------------------
    def _bind(self, env):
        venv = env._v if isinstance(env, Surface_Code_Environment_Multi_Decoding_Cycles) else env
        if not isinstance(venv, VectorEnv):
            raise TypeError("env must be the drop-in Surface_Code_Environment_Multi_Decoding_Cycles or a VectorEnv")
        if self._core is not None and self._env is venv:
            return venv
        if not self.compiled:
            raise RuntimeError("Your tried to fit your agent but it hasn't been compiled yet. Please call `compile()` before `fit()`.")
        if venv.num_actions != self.nb_actions or tuple(venv.obs_shape) != tuple(self.model.input_shape):
            raise ValueError("environment and model shapes disagree")
        old = self.model.get_weights()
        prev = self._core
        max_batch = max(self.batch_size, venv.n_envs)
        self._net = QNetwork(self.model.input_shape, self.model.c_layers, self.model.ff_layers, self.nb_actions,
                             dueling=self.enable_dueling_network, max_batch=max_batch, device=venv.device)
        opt = self.optimizer
        world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
        rank = torch.distributed.get_rank() if world > 1 else 0
        self._core = DQNCore(venv, self._net, batch_size=self.batch_size, memory_limit=self.memory.limit, gamma=self.gamma,
                             lr=opt.lr, beta_1=opt.beta_1, beta_2=opt.beta_2, epsilon=opt.epsilon,
                             target_model_update=self.target_model_update, enable_double_dqn=self.enable_double_dqn,
                             seed=self.seed if self.seed is not None else venv.seed, rank=rank, world_size=world)
        self._env = venv
        if old is not None:
            self._net.set_weights(self._core.params, old)
            self._core.repack()
            self._core.update_target_hard()
        if prev is not None:
            # a different environment object (e.g. test() on a lattice with other rates): the learner's state moves over -- Adam
            # moments, update counter, target network, and the replay ring when its shape still fits
            c = self._core
            c.m.copy_(prev.m); c.v.copy_(prev.v); c.target.copy_(prev.target)
            if c.target_pk is not None and prev.target_pk is not None:
                c.target_pk.copy_(prev.target_pk)
            c.updates, c.vector_steps = prev.updates, prev.vector_steps
            if tuple(prev.obs_ring.shape) == tuple(c.obs_ring.shape):
                for a, b in ((c.obs_ring, prev.obs_ring), (c.action_ring, prev.action_ring), (c.reward_ring, prev.reward_ring),
                             (c.terminal_ring, prev.terminal_ring)):
                    a.copy_(b)
                c.cur, c.filled = prev.cur, prev.filled
            elif prev.filled > 1:
                warnings.warn("the replay memory of the previous environment does not fit the new one (other lattice count or "
                              "observation shape) and was dropped")
        self.memory._core = self._core
        if prev is None and self.memory._saved is not None and not self.memory._restore_into(self._core):
            warnings.warn("the unpickled replay memory does not fit this environment (ring shape "
                          f"{tuple(self.memory._saved['obs_shape'])} vs {tuple(self._core.obs_ring.shape)}) and was dropped")
        return venv

    # -- single-observation API (notebook 3 cell 24: action = dqn.forward(input_state)) ----------------------------
    def forward(self, observation, legal_actions=None):
        if self._core is None:
            raise RuntimeError("forward() needs the agent bound to an environment: call fit()/test() first or agent._bind(env)")
        obs = torch.as_tensor(np.asarray(observation), dtype=torch.uint8, device=self._core.device).reshape((1,) + tuple(self.model.input_shape)).contiguous()
        q = self._net.forward(self._core.params, obs, batch=1)[0].cpu().numpy()
        eps, masked = (self.policy if self.training else self.test_policy).current(self.training)
        if legal_actions is not None and masked:
            legal = sorted(legal_actions)
            return int(max(legal, key=lambda a: (q[a], -a)))
        return int(np.argmax(q))

    def compute_q_values(self, observation):
        obs = torch.as_tensor(np.asarray(observation), dtype=torch.uint8, device=self._core.device).reshape((1,) + tuple(self.model.input_shape)).contiguous()
        return self._net.forward(self._core.params, obs, batch=1)[0].cpu().numpy()

    # -- training ------------------------------------------------------------------------------------------------
    # keras-rl's step arithmetic (DQNAgent.backward runs with self.step = s, the 0-based number of the step just taken, and Agent.fit increments it
    # afterwards): the update of step s happens iff s > nb_steps_warmup (and s % train_interval == 0), the hard target copy iff
    # s % target_model_update == 0, behind that step's update.  Pinned by the reference's own records: every `mean_eps` entry of
    # trained_models/d5_x/0.001/training_history.json equals, to 3e-16, the mean of the annealed epsilon over the episode's steps s > 1000 --
    # its first trained step is s = 1001 (tests/test_host_logic.py::test_reference_mean_eps_records_pin_the_step_arithmetic).  Here self.step is
    # incremented by N per vector step BEFORE these checks, so the vector step just taken covers s in [self.step - N, self.step).
    def _will_train(self, step_after):
        """Whether _maybe_train() will update once self.step has become step_after (the ring then holds one more slot)."""
        core = self._core
        return step_after - 1 > self.nb_steps_warmup and min(core.T, core.filled + 1) >= MIN_FILLED and ((step_after - 1) // core.N) % self.train_interval == 0

    def _maybe_train(self):
        core = self._core
        did = False
        if self.step - 1 > self.nb_steps_warmup and core.filled >= MIN_FILLED and ((self.step - 1) // core.N) % self.train_interval == 0:
            for _ in range(self.updates_per_vector_step):
                core.update()
            did = True
        self._sync_target()
        return did

    def _sync_target(self):
        """Hard target copy when one of the steps just taken, s in [self.step - N, self.step), is a multiple of target_model_update."""
        tmu, lo = self.target_model_update, self.step - self._core.N
        if -(-lo // tmu) * tmu < self.step and self.step > self._last_target_sync:
            self._core.update_target_hard()
            self._last_target_sync = self.step

    def fit(self, env, nb_steps, action_repetition=1, callbacks=None, verbose=1, visualize=False, nb_max_start_steps=0,
            start_step_policy=None, log_interval=10000, nb_max_episode_steps=None, episode_averaging_length=10,
            success_threshold=None, stopping_patience=None, min_nb_steps=500, single_cycle=True, sync_interval=None):
        if action_repetition != 1 or nb_max_start_steps != 0 or nb_max_episode_steps is not None:
            raise NotImplementedError("action_repetition / nb_max_start_steps / nb_max_episode_steps are not used by the reference")
        if single_cycle and getattr(env, "multi_cycle", True):
            warnings.warn("single_cycle=True is ignored: the environment is a multi-cycle one (Environments.py:97)")
        venv = self._bind(env)
        core, N = self._core, venv.n_envs
        self.training = True
        # Several ranks (one per GPU): every step-counted hyper-parameter -- nb_steps, nb_steps_warmup, the epsilon schedule, min_nb_steps,
        # target_model_update -- counts THIS RANK's environment steps (self.step); what is logged and printed ("Step: a/b", nb_steps in
        # training_history.json, "Final Step") is global on both sides: self.step * world_size of nb_steps * world_size.
        # several ranks: episode statistics are summed over the ranks at every synchronisation point so that all of them stop on the
        # same step (a rank leaving alone would hang the others in the gradient all-reduce); callbacks and printing on rank 0 only
        lead = core.rank == 0
        callbacks = list(callbacks or []) if lead else []
        if not lead:
            verbose = 0
        history = History()
        for cb in callbacks:
            cb.on_train_begin()
        if verbose >= 1:
            print(f"Training for {nb_steps} steps ...")
        core.reset_env()
        if sync_interval is None:
            sync_interval = 1 if N == 1 else 64
        core.ensure_comm()              # several ranks: the learner's own RCCL communicator is created here, not inside the first update (~1 s rendezvous)
        t_start = timeit.default_timer()
        lifetimes = deque(maxlen=int(episode_averaging_length))     # (episodes, lifetime_sum) chunks, newest last
        best_avg, best_episode, episode = -np.inf, 0, 0
        has_succeeded = stopped_improving = False
        ep_start, ep_steps, ep_reward = t_start, 0, 0.0
        losses, qs, epss = [], [], []
        start_step = self.step
        stop = False
        loop_iter = 0                   # counted steps of THIS fit() -- the same on every rank (core.vector_steps is not: with one lattice per
                                        # rank only the ranks whose own lattice ended take the uncounted reset step), so the host synchronisations,
                                        # which contain collectives, are gated on it
        try:
            while self.step - start_step < nb_steps and not stop:
                loop_iter += 1
                eps, masked = self.policy.current(True)
                if self._will_train(self.step + N):
                    # acting forward + the update's forwards in one pair of launches; the environment launch draws the next minibatch
                    core.step_and_update(eps, masked_greedy=masked, presample_next=self._will_train(self.step + 2 * N),
                                         extra_updates=self.updates_per_vector_step - 1)
                    self.step += N
                    trained = True
                    self._sync_target()
                else:
                    core.act_and_step(eps, masked_greedy=masked)
                    self.step += N
                    trained = self._maybe_train()
                if trained:                     # keras-rl: a step without an update contributes NaN to every metric, mean_eps included (nanmean per episode)
                    epss.append(eps)
                if loop_iter % sync_interval != 0:
                    continue
                # ---- host sync: episode bookkeeping ---------------------------------------------------------------
                n_ep, life_sum, n_rew, n_stepped = core.read_stats(all_ranks=True)
                if trained:
                    loss_v, mean_q_v = core.read_metrics()
                    losses.append(loss_v)
                    qs.append(mean_q_v)
                ep_steps += n_stepped
                ep_reward += n_rew
                if n_ep == 0:
                    continue
                if N == 1:
                    # keras-rl spends one more forward/backward on the terminal observation before env.reset();
                    # here that is the auto-reset vector step, taken now so that the episode boundary is exact.  Several ranks: n_ep is
                    # the all-reduced count -- only the ranks whose OWN lattice finished take the (uncounted) reset step, a rank whose
                    # lattice is still alive must not step it; the update is collective (gradient all-reduce), so every rank trains.
                    if core.local_stats[0] > 0:
                        core.act_and_step(eps, masked_greedy=masked, record_stats=False)
                    self._maybe_train()
                now = timeit.default_timer()
                # one log record per finished episode (N == 1) or per synchronisation chunk (N > 1)
                episode += n_ep
                self._append_lifetimes(lifetimes, n_ep, life_sum, int(episode_averaging_length))
                tot_ep = sum(c for c, _ in lifetimes)
                rolling = sum(s for _, s in lifetimes) / tot_ep
                if rolling > best_avg:
                    best_avg, best_episode = rolling, episode - 1
                time_since_best = episode - 1 - best_episode
                has_succeeded, stopped_improving = stopping_flags(has_succeeded, stopped_improving, rolling, time_since_best, self.step - start_step,
                                                                  success_threshold, stopping_patience, min_nb_steps)
                # key order and value types of the reference's training_history.json (trained_models/*/*/training_history.json):
                # the three metrics first, then the fork's episode log, then FileLogger's own 'episode' and 'duration'
                logs = {
                    "loss": float(np.mean(losses)) if losses else float("nan"), "mean_q": float(np.mean(qs)) if qs else float("nan"),
                    "mean_eps": float(np.mean(epss)) if epss else float("nan"),
                    "episode_reward": float(ep_reward / n_ep), "nb_episode_steps": int(round(ep_steps / n_ep)),
                    "nb_steps": int(self.step * core.world_size),
                    "episode_lifetimes_rolling_avg": float(rolling), "best_rolling_avg": float(best_avg), "best_episode": int(best_episode),
                    "time_since_best": int(time_since_best), "has_succeeded": has_succeeded, "stopped_improving": stopped_improving,
                }
                history.append(dict(logs, episode=episode - 1, duration=now - ep_start))
                for cb in callbacks:
                    cb.on_episode_end(episode - 1, dict(logs, duration=now - ep_start))
                logs["duration"] = now - ep_start
                if verbose >= 2 and (episode // max(1, log_interval)) != ((episode - n_ep) // max(1, log_interval)):
                    self._print_train_block(episode, nb_steps * core.world_size, logs, now - t_start)
                ep_start, ep_steps, ep_reward = now, 0, 0.0
                losses, qs, epss = [], [], []
                if (has_succeeded or stopped_improving) and (self.step - start_step) >= min_nb_steps:
                    stop = True
        except KeyboardInterrupt:
            pass
        except BaseException:
            # (a DeepQError -- e.g. DQ_ERR_RANGE, raised on every rank at the same synchronisation --, or anything else: the communicator does not outlive
            # fit(); aborted, not destroyed: a destroy may wait for peers that are inside a collective this rank will never join)
            core.close_comm(abort=True)
            self.training = False
            raise
        torch.cuda.synchronize(core.device)
        core.close_comm()               # (every rank, everything drained: before the process group that did its rendezvous can be destroyed)
        dt = timeit.default_timer() - t_start
        for cb in callbacks:
            cb.on_train_end()
        if verbose >= 1:
            final = history.history.get("episode_lifetimes_rolling_avg", [float("nan")])[-1]
            print(f"Training Finished in {dt:.3f} seconds\n        \nFinal Step: {self.step * core.world_size}\nSucceeded: {has_succeeded}\n"
                  f"Stopped_Improving: {stopped_improving}\nFinal Episode Lifetimes Rolling Avg: {final:.3f}")
        self.training = False
        return history

    @staticmethod
    def _append_lifetimes(dq_, n_ep, life_sum, L):
        """Keep (count, sum) chunks covering the most recent L episodes (chunks are split so the window is exact
        for single episodes and chunk-granular for vector runs)."""
        dq_.append((n_ep, life_sum))
        tot = sum(c for c, _ in dq_)
        while tot > L and len(dq_) > 1:
            c, _ = dq_[0]
            if tot - c >= L:
                dq_.popleft()
                tot -= c
            else:
                break

    @staticmethod
    def _print_train_block(episode, nb_steps, logs, total):
        # README.md:411-427, character for character (the separator is followed by a line of 16 blanks)
        print("-----------------\n                ")
        print(f"Episode: {episode}\nStep: {logs['nb_steps']}/{nb_steps}\nThis Episode Steps: {logs['nb_episode_steps']}\n"
              f"This Episode Reward: {logs['episode_reward']}\nThis Episode Duration: {logs['duration']:.3f}s\n"
              f"Rolling Lifetime length: {logs['episode_lifetimes_rolling_avg']:.3f}\nBest Lifetime Rolling Avg: {logs['best_rolling_avg']}\n"
              f"Best Episode: {logs['best_episode']}\nTime Since Best: {logs['time_since_best']}\nHas Succeeded: {logs['has_succeeded']}\n"
              f"Stopped Improving: {logs['stopped_improving']}\nMetrics: loss: {logs['loss']:.6f}, mean_q: {logs['mean_q']:.6f}, mean_eps: {logs['mean_eps']:.6f}\n"
              f"Total Training Time: {total:.3f}s\n")

    # -- evaluation ------------------------------------------------------------------------------------------------
    def test(self, env, nb_episodes=1, action_repetition=1, callbacks=None, visualize=True, nb_max_episode_steps=None,
             nb_max_start_steps=0, start_step_policy=None, verbose=1, episode_averaging_length=None, interval=100, single_cycle=True):
        """Greedy episodes (TRAIN:206).  With N lattices, lattice i contributes its first ceil-share of episodes, so the
        sample is not biased towards short episodes.  history keys: episode_reward, nb_steps, episode_lifetime,
        episode_lifetimes_rolling_avg."""
        venv = self._bind(env)
        core, N = self._core, venv.n_envs
        self.training = False
        history = History()
        if verbose >= 1:
            print(f"Testing for {nb_episodes} episodes ...")
        quota = np.full(N, nb_episodes // N, dtype=np.int64)
        quota[:nb_episodes % N] += 1
        eps, masked = self.test_policy.current(False)
        core.begin_eval()                           # nothing of the evaluation reaches the replay memory (keras-rl: training=False)
        try:
            return self._test_loop(core, venv, N, quota, eps, masked, history, verbose, interval)
        finally:
            core.end_eval()

    def _test_loop(self, core, venv, N, quota, eps, masked, history, verbose, interval, sync_interval=None):
        """Device-resident: the episode records are appended on the device (dq_test_bookkeeping, one small launch per vector step); the host
        looks at the record counter every `sync_interval` steps only, so a batched evaluation costs what its kernels cost (rounds 1-2 did
        four device-to-host copies and a Python loop over the finished lattices per vector step)."""
        from ._lib import check, ptr
        dev = core.device
        total = int(quota.sum())
        quota_d = torch.from_numpy(quota.astype(np.int32)).to(dev)
        ep_reward = torch.zeros(N, dtype=torch.float32, device=dev)
        ep_len = torch.zeros(N, dtype=torch.int32, device=dev)
        records = torch.zeros((max(total, 1), 5), dtype=torch.int32, device=dev)
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        core.reset_env()
        L = core.L
        step = 0
        if sync_interval is None:
            sync_interval = 1 if N == 1 else 64
        while total > 0:
            slot = core.cur
            core.act_and_step(eps, masked_greedy=masked, record_stats=False)
            check(L.dq_test_bookkeeping(ptr(core.terminal_ring[slot]), ptr(venv.was_reset), ptr(core.reward_ring[slot]), ptr(venv.lifetime), N, step,
                                        ptr(quota_d), ptr(ep_reward), ptr(ep_len), ptr(records), total, ptr(counter), core._stream()))
            step += 1
            if step % sync_interval == 0 and int(counter.item()) >= total:
                break
        rec = records.cpu().numpy()[:total]
        rec = rec[np.lexsort((rec[:, 1], rec[:, 0]))]                     # by vector step, then lattice: the serial loop's order
        lifetimes, run = [], 0.0
        for episode, (t, i, rbits, length, life) in enumerate(rec, 1):
            run += int(life)
            logs = {"episode_reward": float(np.int32(rbits).view(np.float32)), "nb_steps": int(length), "episode_lifetime": int(life),
                    "episode_lifetimes_rolling_avg": run / episode}
            history.append(logs)
            if verbose >= 2 and (episode - 1) % max(1, interval) == 0:
                print(f"-----------------\nEpisode: {episode}\nThis Episode Length: {logs['nb_steps']}\n"
                      f"This Episode Reward: {logs['episode_reward']}\nThis Episode Lifetime: {logs['episode_lifetime']}\n\n"
                      f"Episode Lifetimes Avg: {logs['episode_lifetimes_rolling_avg']:.3f}\n")
        core.read_stats()
        return history
