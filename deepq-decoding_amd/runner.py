"""Config-dict driven runs and the one-node grid runner.

The reference drives training from two pickled dicts per grid point --
    ../fixed_config.p           (Generate_Base_Configs_and_Simulation_Scripts.py:10-29, "GEN")
    config_N/variable_config_N.p (GEN:54-63)
-- through `python Single_Point_Training_Script.py N` ("TRAIN") or, for a grid point that continues from an earlier error rate,
`Single_Point_Continue_Training_Script.py N <dir>` ("CONT": unpickles config_N/memory.p and loads config_N/initial_dqn_weights.h5f
first), one SLURM job of four CPU cores each (GEN:80-109); `Controller.py` ("CTRL") collects `results.p`, keeps the best point
above a threshold and spawns the next error rate's grid from it (CTRL:117-274).

Here the same directory layout, file names and dict schema run on the GPU path:

    train_single_point(config_dir)     what TRAIN / CONT do for one grid point (same output files: started_at.p,
                                       training_history.json, memory.p, final_dqn_weights.h5f, results.p, all_results.p)
    write_grid(...)                    GEN's fixed_config.p + config_N/variable_config_N.p tree (the dicts only: no SLURM scripts)
    run_grid(p_dir, gpus)              every config_N of an error-rate directory, ONE GRID POINT PER GPU at a time (a process per
                                       point, HIP_VISIBLE_DEVICES pins it): the node's 8 GPUs take the place of 8 SLURM jobs
    collect_results / select_best / spawn_next / run_error_rate_ladder
                                       CTRL's bookkeeping: results.p -> best point above the threshold -> next error rate's grid
                                       seeded with its weights and memory
`n_envs` lattices per grid point is this build's extension (1 = the reference's loop, step for step): with N lattices one vector
step advances all of them and `batch_size` / the step-counted hyper-parameters keep their meaning per environment step.
"""
import datetime
import glob
import importlib
import itertools
import json
import os
import pickle
import shutil
import subprocess
import sys

FIXED_KEYS = ("d", "use_Y", "train_freq", "batch_size", "print_freq", "rolling_average_length", "stopping_patience", "error_model",
              "c_layers", "ff_layers", "max_timesteps", "volume_depth", "testing_length", "buffer_size", "dueling", "masked_greedy",
              "static_decoder")                                                                            # GEN:10-27
VARIABLE_KEYS = ("p_phys", "p_meas", "success_threshold", "learning_starts", "learning_rate", "exploration_fraction", "max_eps",
                 "target_network_update_freq", "gamma", "final_eps")                                       # GEN:54-63
# CTRL:16-32
THRESHOLD_DICT = {"0.001": 1000, "0.003": 334, "0.005": 200, "0.007": 142, "0.009": 112, "0.011": 91, "0.013": 77, "0.015": 67, "0.017": 59}
P_PHYS_LIST = [0.001, 0.003, 0.005, 0.007, 0.009, 0.011, 0.013, 0.015, 0.017]
CONTINUE_GRID = dict(learning_starts=[1000], learning_rate=[0.0001, 0.00005, 0.00001, 0.000005], exploration_fraction=[100000, 200000],
                     max_eps=[1.0, 0.5, 0.25], target_network_update_freq=[2500, 5000], gamma=[0.99], final_eps=[0.04, 0.02, 0.001])
INITIAL_GRID = dict(learning_starts=[1000], learning_rate=[0.0001, 0.00005, 0.00001], exploration_fraction=[100000, 200000],
                    max_eps=[1.0], target_network_update_freq=[2500, 5000], gamma=[0.99], final_eps=[0.04, 0.02, 0.001])   # GEN:36-43
GRID_ORDER = ("learning_starts", "learning_rate", "exploration_fraction", "max_eps", "target_network_update_freq", "gamma", "final_eps")


def load_configs(config_dir, fixed_config_path=None):
    """TRAIN:36-53: (all_configs, number) of a config_N directory; fixed_config.p is looked up one level above the error-rate
    directory like TRAIN:41 unless given."""
    config_dir = os.path.abspath(config_dir)
    number = os.path.basename(config_dir.rstrip("/")).split("_")[-1]
    if fixed_config_path is None:
        fixed_config_path = os.path.join(os.path.dirname(config_dir), "..", "fixed_config.p")
    with open(fixed_config_path, "rb") as f:
        fixed = pickle.load(f)
    with open(os.path.join(config_dir, f"variable_config_{number}.p"), "rb") as f:
        variable = pickle.load(f)
    all_configs = dict(fixed)
    all_configs.update(variable)
    missing = [k for k in FIXED_KEYS + VARIABLE_KEYS if k not in all_configs]
    if missing:
        raise KeyError(f"configuration keys missing from {config_dir}: {missing}")
    return all_configs, number


def train_single_point(config_dir, fixed_config_path=None, n_envs=1, verbose=2, device=None, seed=None, test_rates=None,
                       batch_size=None, sync_interval=None):
    """One grid point, TRAIN:92-222 (CONT when config_dir holds memory.p + initial_dqn_weights.h5f).  Returns all_results."""
    dq = importlib.import_module(__package__)
    import torch
    cfg, number = load_configs(config_dir, fixed_config_path)
    if device is not None:
        torch.cuda.set_device(device)
    # static_decoder: the reference loads a Keras referee from ../static_decoder (TRAIN:55-58); that blob is not shipped and a Keras
    # model cannot run inside the kernel, so True selects the built-in look-up referee
    kw = dict(d=cfg["d"], p_phys=cfg["p_phys"], p_meas=cfg["p_meas"], error_model=cfg["error_model"], use_Y=cfg["use_Y"],
              volume_depth=cfg["volume_depth"])
    if n_envs == 1:
        env = dq.Surface_Code_Environment_Multi_Decoding_Cycles(static_decoder=None, **({} if seed is None else dict(seed=seed)), **kw)
        shape, n_actions = env.observation_space.shape, env.num_actions
    else:
        env = dq.VectorEnv(n_envs=n_envs, **({} if seed is None else dict(seed=seed)), **kw)
        shape, n_actions = env.obs_shape, env.num_actions

    def agent(policy, memory):
        model = dq.build_convolutional_nn(cfg["c_layers"], cfg["ff_layers"], shape, n_actions)
        a = dq.DQNAgent(model=model, nb_actions=n_actions, memory=memory, nb_steps_warmup=cfg["learning_starts"],
                        target_model_update=cfg["target_network_update_freq"], policy=policy,
                        test_policy=dq.GreedyQPolicy(masked_greedy=True), gamma=cfg["gamma"], enable_dueling_network=cfg["dueling"],
                        batch_size=batch_size or cfg["batch_size"], train_interval=cfg["train_freq"], seed=seed)
        a.compile(dq.Adam(lr=cfg["learning_rate"]))
        return a
    memory_file = os.path.join(config_dir, "memory.p")
    initial_weights = os.path.join(config_dir, "initial_dqn_weights.h5f")
    continuing = os.path.exists(memory_file) and os.path.exists(initial_weights)
    if continuing:                                                                                 # CONT:109-110
        with open(memory_file, "rb") as f:
            memory = pickle.load(f)
    else:
        memory = dq.SequentialMemory(limit=cfg["buffer_size"], window_length=1)
    policy = dq.LinearAnnealedPolicy(dq.EpsGreedyQPolicy(masked_greedy=cfg["masked_greedy"]), attr="eps", value_max=cfg["max_eps"],
                                     value_min=cfg["final_eps"], value_test=0.0, nb_steps=cfg["exploration_fraction"])
    dqn = agent(policy, memory)
    if continuing:
        dqn.model.load_weights(initial_weights)                                                    # CONT:135-136
    logging_callback = dq.FileLogger(filepath=os.path.join(config_dir, "training_history.json"), interval=cfg["print_freq"])
    with open(os.path.join(config_dir, "started_at.p"), "wb") as f:
        pickle.dump(datetime.datetime.now(), f)                                                    # TRAIN:134-136
    fit_kw = {} if sync_interval is None else dict(sync_interval=sync_interval)
    dqn.fit(env, nb_steps=cfg["max_timesteps"], action_repetition=1, callbacks=[logging_callback], verbose=verbose, visualize=False,
            nb_max_start_steps=0, start_step_policy=None, log_interval=cfg["print_freq"], nb_max_episode_steps=None,
            episode_averaging_length=cfg["rolling_average_length"], success_threshold=cfg["success_threshold"],
            stopping_patience=cfg["stopping_patience"], min_nb_steps=cfg["exploration_fraction"], single_cycle=False, **fit_kw)
    with open(memory_file, "wb") as f:
        pickle.dump(dqn.memory, f)                                                                 # TRAIN:156-157
    final_weights_file = os.path.join(config_dir, "final_dqn_weights.h5f")
    dqn.save_weights(final_weights_file, overwrite=True)                                           # TRAIN:159-160
    # evaluation sweep, TRAIN:164-222: a fresh agent with the saved weights, error rates 0.001, 0.002, ... until the average lifetime
    # drops below 1/p (or 20 rates)
    tester = agent(dq.GreedyQPolicy(masked_greedy=True), dq.SequentialMemory(limit=cfg["buffer_size"], window_length=1))
    tester._bind(env)
    tester.model.load_weights(final_weights_file)
    trained_at = cfg["p_phys"]
    error_rates = [j * 0.001 for j in range(1, 21)] if test_rates is None else list(test_rates)
    all_results = {}
    for count, err_rate in enumerate(error_rates):
        env.p_phys = err_rate
        env.p_meas = err_rate
        th = tester.test(env, nb_episodes=cfg["testing_length"], visualize=False, verbose=verbose, interval=10, single_cycle=False)
        results = th.history["episode_lifetimes_rolling_avg"]
        final_result = results[-1:][0]
        all_results[str(err_rate)[:5]] = final_result
        if abs(trained_at - err_rate) < 1e-6:
            with open(os.path.join(config_dir, "results.p"), "wb") as f:
                pickle.dump(results, f)
        if final_result < 1.0 / err_rate or count == len(error_rates) - 1:
            break
    with open(os.path.join(config_dir, "all_results.p"), "wb") as f:
        pickle.dump(all_results, f)
    return all_results


# ---- the grid ------------------------------------------------------------------------------------------------------------------------
def write_grid(family_dir, fixed_config, p_phys, success_threshold, grid=None, spawn_from=None):
    """GEN:29-74 (spawn_from=None) / CTRL:189-263 (spawn_from = {name: config_dir of an earlier point}): family_dir/fixed_config.p and
    family_dir/<p_phys>/config_N/variable_config_N.p for the cartesian grid, N counting from 1 in the reference's loop order.  A
    spawned point also receives that earlier point's final weights as initial_dqn_weights.h5f and its memory.p (CTRL:262-268).
    Returns the list of config directories."""
    grid = dict(INITIAL_GRID if spawn_from is None else CONTINUE_GRID, **(grid or {}))
    os.makedirs(family_dir, exist_ok=True)
    with open(os.path.join(family_dir, "fixed_config.p"), "wb") as f:
        pickle.dump(dict(fixed_config), f)
    p_dir = os.path.join(family_dir, str(p_phys))
    os.makedirs(os.path.join(p_dir, "output_files"), exist_ok=True)
    dirs, counter = [], 1
    for parent in ([None] if spawn_from is None else list(spawn_from.values())):
        for values in itertools.product(*(grid[k] for k in GRID_ORDER)):
            var = dict(zip(GRID_ORDER, values))
            var.update(p_phys=p_phys, p_meas=p_phys, success_threshold=success_threshold)
            cdir = os.path.join(p_dir, f"config_{counter}")
            if os.path.exists(cdir):
                shutil.rmtree(cdir)                                                                # GEN:69 / CTRL:217
            os.makedirs(cdir)
            with open(os.path.join(cdir, f"variable_config_{counter}.p"), "wb") as f:
                pickle.dump({k: var[k] for k in VARIABLE_KEYS}, f)
            if parent is not None:
                shutil.copyfile(os.path.join(parent, "final_dqn_weights.h5f"), os.path.join(cdir, "initial_dqn_weights.h5f"))
                shutil.copyfile(os.path.join(parent, "memory.p"), os.path.join(cdir, "memory.p"))
            dirs.append(cdir)
            counter += 1
    return dirs


def _config_dirs(p_dir):
    return sorted(glob.glob(os.path.join(p_dir, "config_*")), key=lambda d: int(d.rsplit("_", 1)[1]))


def run_grid(p_dir, gpus=None, n_envs=1, max_points=None, extra_args=(), timeout=None):
    """Runs every config_N under an error-rate directory, one grid point per GPU at a time (each point is a process pinned to its GPU
    with HIP_VISIBLE_DEVICES; stdout / stderr go to output_files/out_<p>_<N>.out / err_<p>_<N>.err like GEN:81-83).  `gpus`: list
    of device ordinals (default: all visible).  Returns {N: return code}."""
    import torch
    if gpus is None:
        gpus = list(range(max(1, torch.cuda.device_count())))
    dirs = _config_dirs(p_dir)[:max_points]
    pname = os.path.basename(os.path.normpath(p_dir))
    os.makedirs(os.path.join(p_dir, "output_files"), exist_ok=True)
    pending, running, codes = list(dirs), {}, {}
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "grid_run.py")
    while pending or running:
        for g in gpus:
            if g not in running and pending:
                cdir = pending.pop(0)
                n = cdir.rsplit("_", 1)[1]
                out = open(os.path.join(p_dir, "output_files", f"out_{pname}_{n}.out"), "w")
                err = open(os.path.join(p_dir, "output_files", f"err_{pname}_{n}.err"), "w")
                env = dict(os.environ, HIP_VISIBLE_DEVICES=str(g))
                cmd = [sys.executable, script, "point", cdir, "--n-envs", str(n_envs)] + list(extra_args)
                running[g] = (subprocess.Popen(cmd, stdout=out, stderr=err, env=env), n, out, err)
        for g, (proc, n, out, err) in list(running.items()):
            try:
                rc = proc.wait(timeout=0.5 if len(running) > 1 or pending else timeout)
            except subprocess.TimeoutExpired:
                continue
            codes[int(n)] = rc
            out.close(), err.close()
            del running[g]
    return codes


def collect_results(p_dir):
    """CTRL:60-83: {N: last rolling-average lifetime at the training error rate | 'still running' | 'not started'}."""
    res = {}
    for cdir in _config_dirs(p_dir):
        n = cdir.rsplit("_", 1)[1]
        if os.path.exists(os.path.join(cdir, "results.p")):
            with open(os.path.join(cdir, "results.p"), "rb") as f:
                res[n] = pickle.load(f)[-1:][0]
        elif os.path.exists(os.path.join(cdir, "started_at.p")):
            res[n] = "still running"
        else:
            res[n] = "not started"
    return res


def select_best(results, threshold, num_best=1):
    """CTRL:117-156: the `num_best` finished points above the threshold, best first."""
    ok = sorted(((v, k) for k, v in results.items() if not isinstance(v, str) and v > threshold), reverse=True)
    return {k: v for v, k in ok[:num_best]}


def spawn_next(family_dir, fixed_config, p_from, p_to, success_threshold=100000, grid=None, num_best=1, thresholds=None):
    """One Controller pass (CTRL:117-274): results of <p_from> -> best points above the testing threshold -> the grid of <p_to>
    continuing from them.  Writes results/results_from_<p>.txt and best_results_from_<p>.txt like CTRL:86-99,136-155.  Returns the
    new config directories ([] when nothing passed)."""
    p_dir = os.path.join(family_dir, str(p_from))
    results = collect_results(p_dir)
    os.makedirs(os.path.join(family_dir, "results"), exist_ok=True)
    with open(os.path.join(family_dir, "results", f"results_from_{p_from}.txt"), "w") as f:
        for k in sorted(results, key=int):
            f.write(f"{k}: {results[k]}\n")
    best = select_best(results, (thresholds or THRESHOLD_DICT)[str(p_from)], num_best)
    with open(os.path.join(family_dir, "results", f"best_results_from_{p_from}.txt"), "w") as f:
        for k, v in best.items():
            f.write(f"{k}: {v}\n")
    if not best:
        return []
    return write_grid(family_dir, fixed_config, p_to, success_threshold, grid=grid,
                      spawn_from={k: os.path.join(p_dir, f"config_{k}") for k in best})


def run_error_rate_ladder(family_dir, fixed_config, p_list=None, gpus=None, n_envs=1, grid0=None, grid=None, max_points=None,
                          thresholds=None, extra_args=()):
    """The whole CTRL loop on one node: initial grid at p_list[0], then for each further error rate the grid spawned from the best
    point so far; stops where nothing passes the threshold.  Returns {p: results}."""
    p_list = list(p_list or P_PHYS_LIST)
    write_grid(family_dir, fixed_config, p_list[0], 100000, grid=grid0)
    summary = {}
    for i, p in enumerate(p_list):
        run_grid(os.path.join(family_dir, str(p)), gpus=gpus, n_envs=n_envs, max_points=max_points, extra_args=extra_args)
        summary[p] = collect_results(os.path.join(family_dir, str(p)))
        if i + 1 == len(p_list) or not spawn_next(family_dir, fixed_config, p, p_list[i + 1], grid=grid, thresholds=thresholds):
            break
    with open(os.path.join(family_dir, "history.json"), "w") as f:
        json.dump({str(k): v for k, v in summary.items()}, f, indent=1, default=str)
    return summary
