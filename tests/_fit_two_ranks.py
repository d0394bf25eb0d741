"""Helper of tests/test_distributed_gpu.py (launched under torch.distributed.run, 2 ranks, gloo, both on the one GPU): DQNAgent.fit with
the reference's early-stopping keywords on ranks whose LOCAL episode statistics differ wildly -- every rank must leave the loop on the
same step (the decision is taken from the all-reduced statistics), with identical parameters; only rank 0 logs."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


SYNC_INTERVAL = None


def main():
    out_dir = sys.argv[1]
    single = len(sys.argv) > 2 and sys.argv[2].startswith("single")
    global SYNC_INTERVAL
    SYNC_INTERVAL = 4 if len(sys.argv) > 2 and sys.argv[2] == "single4" else None
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    if os.environ.get("DQ_DIST_FORCE") == "1":                      # (one rank through the REAL backend: the learner gets a RCCL communicator of its own)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    dq = importlib.import_module("deepq-decoding_amd")
    if len(sys.argv) > 2 and sys.argv[2] == "range":
        return main_range(dq, out_dir, rank, world)
    if len(sys.argv) > 2 and sys.argv[2] == "rangeauto":
        return main_range(dq, out_dir, rank, world, default_path=True)
    if len(sys.argv) > 2 and sys.argv[2] == "fitrange":
        return main_fit_range(dq, out_dir, rank, world)
    if single:
        return main_single(dq, out_dir, rank)
    N = 64
    # rank 1's lattices are almost noiseless: its episodes are ~100x rarer than rank 0's, so a patience counted in LOCAL episodes would run out
    # on rank 0 long before rank 1
    p = 0.011 if rank == 0 else 0.0005
    env = dq.VectorEnv(n_envs=N, env_id_base=rank * N, d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=p, p_meas=p)
    model = dq.build_convolutional_nn([[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]], env.obs_shape, env.num_actions)
    policy = dq.LinearAnnealedPolicy(dq.EpsGreedyQPolicy(masked_greedy=False), attr="eps", value_max=1.0, value_min=0.02, value_test=0.0,
                                     nb_steps=5000)
    agent = dq.DQNAgent(model=model, nb_actions=env.num_actions, memory=dq.SequentialMemory(limit=N * 64, window_length=1),
                        nb_steps_warmup=N * 6, target_model_update=N * 50, policy=policy, test_policy=dq.GreedyQPolicy(masked_greedy=True),
                        gamma=0.99, enable_dueling_network=True, batch_size=32, seed=(1, 2))
    agent.compile(dq.Adam(lr=1e-4))
    log = dq.FileLogger(os.path.join(out_dir, "training_history.json"), interval=1)
    hist = agent.fit(env, nb_steps=N * 400, callbacks=[log], verbose=0, log_interval=100, episode_averaging_length=30, success_threshold=None,
                     stopping_patience=150, min_nb_steps=N * 20, single_cycle=False, sync_interval=4)
    chk = int(agent._core.params.view(torch.int32).to(torch.int64).sum().item())
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(dict(step=agent.step, updates=agent._core.updates, params=chk, stopped=hist.history["stopped_improving"][-1],
                       records=len(hist.history["episode"])), f)
    dist.barrier()
    dist.destroy_process_group()


def main_range(dq, out_dir, rank, world, default_path=False):
    """The range guard under several ranks: ONE rank's minibatch holds TD errors beyond the fused backward's range.  The whole update must
    be discarded on EVERY rank (the overflowing rank turns its gradient into NaNs, the all-reduce carries them to the others, the guarded Adam step skips
    and flags them), the replicas stay bit-identical, and every rank reports DQ_ERR_RANGE at the same synchronisation point."""
    N = 64
    cfg = dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011)
    env = dq.VectorEnv(n_envs=N, env_id_base=rank * N, **cfg)
    net = dq.QNetwork(env.obs_shape, [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]], env.num_actions, max_batch=N)
    core = dq.DQNCore(env, net, batch_size=N, memory_limit=N * 8, gamma=0.99, lr=1e-3, seed=(5, 6), rank=rank, world_size=world)
    core.reset_env()
    for _ in range(4):
        core.act_and_step(1.0, use_q=False)
    for _ in range(3):
        core.step_and_update(0.5)
    core.read_metrics()                                             # healthy so far: no error
    before = core.params.clone()
    if rank == 0:
        core.reward_ring.fill_(1e6)                                 # TD errors of ~1e6 on this rank only
    core.step_and_update(0.5)
    torch.cuda.synchronize()
    unchanged = bool(torch.equal(core.params, before))
    raised, warned = False, False
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        try:
            core.read_metrics()
        except dq.DeepQError as e:
            raised = e.status == -6
        warned = any("measures the gradient scale" in str(x.message) for x in w)
    if default_path:
        # DQ_TD_AUTOSCALE unset: no error -- every rank warns at the SAME synchronisation, counts the one discarded update, and switches to the
        # measured scale; the poisoned memory is then CARRIED (rank 0 keeps its 1e6 rewards), the replicas stay bit-identical
        if rank == 0:
            core.reward_ring.fill_(1e6)
        for _ in range(3):
            core.step_and_update(0.5)
    else:
        if rank == 0:
            core.reward_ring.zero_()
        core.step_and_update(0.5)                                   # the loop carries on
    core.read_metrics()
    chk = int(core.params.view(torch.int32).to(torch.int64).sum().item())
    finite = bool(torch.isfinite(core.params).all())
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(dict(unchanged=unchanged, raised=raised, warned=warned, auto_scale=bool(core.auto_scale), discarded=int(core.discarded_updates), params=chk,
                       finite=finite, moved=not torch.equal(core.params, before)), f)
    dist.barrier()
    dist.destroy_process_group()


def main_fit_range(dq, out_dir, rank, world):
    """A DQ_ERR_RANGE raised INSIDE DQNAgent.fit: rank 0's callback poisons ITS replay rewards in the middle of the run; the update
    that samples them is discarded on every rank and every rank's fit() raises DeepQError(DQ_ERR_RANGE) at the same synchronisation -- and must leave
    with the learner's communicator closed (core._rccl None, a later fit() would create a new one) and the process group still usable."""
    N = 64
    cfg = dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011)
    env = dq.VectorEnv(n_envs=N, env_id_base=rank * N, **cfg)
    model = dq.build_convolutional_nn([[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]], env.obs_shape, env.num_actions)
    agent = dq.DQNAgent(model=model, nb_actions=env.num_actions, memory=dq.SequentialMemory(limit=N * 16, window_length=1), nb_steps_warmup=N * 4,
                        target_model_update=N * 50, policy=dq.EpsGreedyQPolicy(eps=0.3, masked_greedy=True), test_policy=dq.GreedyQPolicy(masked_greedy=True),
                        gamma=0.99, enable_dueling_network=True, batch_size=N, seed=(1, 2))
    agent.compile(dq.Adam(lr=1e-4))
    state = dict(comm_seen=None, episodes=0)

    class Poison:                                                   # (callbacks run on rank 0 only)
        def on_train_begin(self): pass
        def on_train_end(self): pass
        def on_episode_end(self, episode, logs):
            state["episodes"] += 1
            if state["episodes"] == 3:
                agent._core.reward_ring.fill_(1e6)

    raised, status = False, None
    try:
        agent.fit(env, nb_steps=N * 400, callbacks=[Poison()] if rank == 0 else [], verbose=0, episode_averaging_length=30, success_threshold=None,
                  stopping_patience=None, min_nb_steps=0, single_cycle=False, sync_interval=4)
    except dq.DeepQError as e:
        raised, status = True, e.status
    core = agent._core
    closed = core._rccl is None and not core._rccl_tried
    t = torch.tensor([rank + 1.0])
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t)                                              # the process group survived
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(dict(raised=raised, status=status, closed=closed, step=agent.step, group_sum=float(t.item()), finite=bool(torch.isfinite(core.params).all()),
                       params=int(core.params.view(torch.int32).to(torch.int64).sum().item()), training=bool(agent.training)), f)
    dist.barrier()
    dist.destroy_process_group()


def main_single(dq, out_dir, rank):
    """ONE lattice per rank (keras-rl's own shape) under two ranks: a rank whose lattice is still alive must not spend the extra
    'forward/backward on the terminal observation' step when only the OTHER rank's episode ended -- its lattice's steps are all counted
    (lifetime statistics) -- while the update stays collective.  Rank 1 is nearly noiseless, so its episodes practically never end."""
    p = 0.05 if rank == 0 else 0.0002
    env = dq.VectorEnv(n_envs=1, env_id_base=rank, d=3, error_model="X", use_Y=False, volume_depth=3, p_phys=p, p_meas=p)
    model = dq.build_convolutional_nn([[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]], env.obs_shape, env.num_actions)
    agent = dq.DQNAgent(model=model, nb_actions=env.num_actions, memory=dq.SequentialMemory(limit=500, window_length=1), nb_steps_warmup=20,
                        target_model_update=50, policy=dq.EpsGreedyQPolicy(eps=0.3, masked_greedy=True), test_policy=dq.GreedyQPolicy(masked_greedy=True),
                        gamma=0.99, enable_dueling_network=True, batch_size=8, seed=(1, 2))
    agent.compile(dq.Adam(lr=1e-4))
    # sync_interval > 1 with one lattice per rank: the ranks' launch counters diverge (only the rank whose lattice ended takes the uncounted
    # reset step), so the host synchronisations -- which contain collectives -- must be gated on a rank-independent counter
    hist = agent.fit(env, nb_steps=300, verbose=0, episode_averaging_length=10, success_threshold=None, stopping_patience=None,
                     min_nb_steps=0, single_cycle=False, **({} if SYNC_INTERVAL is None else dict(sync_interval=SYNC_INTERVAL)))
    core = agent._core
    chk = int(core.params.view(torch.int32).to(torch.int64).sum().item())
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        # vector_steps counts every environment launch of this rank: the 300 counted steps + one uncounted reset step per LOCAL episode end
        json.dump(dict(step=agent.step, updates=core.updates, params=chk, vector_steps=core.vector_steps,
                       episodes_global=(hist.history["episode"][-1] + 1) if hist.history.get("episode") else 0), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
