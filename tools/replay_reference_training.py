#!/usr/bin/env python3
"""Replays one of the reference's FROM-SCRATCH training runs through this build's DQNAgent.fit and compares the curves with the reference's
own record (GPU box; the record travels as the data fixture tests/golden/training_history_<family>_<p>.npz made by
tools/gen_weight_fixtures.py -- no reference file is read here).

    python tools/replay_reference_training.py [--family d5_x] [--p 0.001] [--max-steps N] [--lattices 1] [--out profiles/....json]

The recipe is the reference's (fixed_config.p + variable_config_35.p of trained_models/d5_x/0.001: d = 5 bit-flip noise, p = 0.001, one
lattice, batch 32, Adam 1e-5, epsilon 1 -> 0.02 over 200 000 steps, target copy every 5000, learning_starts 1000, buffer 50 000, gamma
0.99, rolling average over 1000 episodes, max 1 000 000 steps) run through runner.train_single_point, i.e. the call sequence of
cluster_scripts/<family>/<p>/Single_Point_Training_Script.py.  What differs by construction: the random numbers (the reference is unseeded)
and the referee (the reference's Keras referee is not shipped; here the minimum-weight table) -- so the comparison is STATISTICAL:
    * mean_eps as a function of the step count: exact (same annealing, same step arithmetic);
    * mean_q, loss and the rolling lifetime at the same step counts: within bands (compare());
    * the greedy lifetime of the trained agent at the training rate against all_results.p.
Prints a table and returns / writes a JSON record."""
import argparse
import importlib
import json
import os
import pickle
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ROOT))
CHECKPOINTS = (10000, 25000, 50000, 100000, 150000, 200000, 234000, 300000, 400000, 500000, 750000, 990000)


def load_record(family, p):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"training_history_{family}_{p}.npz"))
    return g, json.loads(str(g["fixed_config_json"])), json.loads(str(g["variable_config_json"]))


def curve_at(steps, values, at, window=25000):
    """Mean of the per-episode values whose episode ended within `window` steps before `at`, weighted by nothing (as FileLogger's rows)."""
    steps, values = np.asarray(steps), np.asarray(values, dtype=np.float64)
    sel = (steps > at - window) & (steps <= at) & ~np.isnan(values)
    return float(values[sel].mean()) if sel.any() else float("nan")


def last_at(steps, values, at):
    steps = np.asarray(steps)
    i = int(np.searchsorted(steps, at, side="right")) - 1
    return float(values[i]) if i >= 0 else float("nan")


def run(family="d5_x", p="0.001", max_steps=None, lattices=1, seed=(20181012, 7), verbose=0, eval_lattices=256, sweep_rates=None):
    dq = importlib.import_module("deepq-decoding_amd")
    runner = importlib.import_module("deepq-decoding_amd.runner")
    g, fixed, var = load_record(family, p)
    if max_steps:
        fixed = dict(fixed, max_timesteps=int(max_steps))
    n = str(g["variable_config_file"]).split("_")[-1].split(".")[0]
    work = tempfile.mkdtemp(prefix="dq_replay_")
    with open(os.path.join(work, "fixed_config.p"), "wb") as f:
        pickle.dump(fixed, f)
    cdir = os.path.join(work, p, f"config_{n}")
    os.makedirs(cdir)
    with open(os.path.join(cdir, f"variable_config_{n}.p"), "wb") as f:
        pickle.dump(var, f)
    t0 = time.time()
    runner.train_single_point(cdir, n_envs=lattices, verbose=verbose, seed=seed, test_rates=[])
    train_s = time.time() - t0
    ours = json.load(open(os.path.join(cdir, "training_history.json")))
    # greedy evaluation at the training rate on a batch of lattices (the reference: testing_length = 101 serial episodes, all_results.p)
    import torch
    cfg = dict(fixed, **var)
    venv = dq.VectorEnv(n_envs=eval_lattices, d=cfg["d"], p_phys=cfg["p_phys"], p_meas=cfg["p_meas"], error_model=cfg["error_model"],
                        use_Y=cfg["use_Y"], volume_depth=cfg["volume_depth"], seed=(seed[0] + 1, seed[1]))
    model = dq.build_convolutional_nn(cfg["c_layers"], cfg["ff_layers"], venv.obs_shape, venv.num_actions)
    tester = dq.DQNAgent(model=model, nb_actions=venv.num_actions, memory=dq.SequentialMemory(limit=1000, window_length=1), nb_steps_warmup=10,
                         target_model_update=10, policy=dq.GreedyQPolicy(masked_greedy=True), test_policy=dq.GreedyQPolicy(masked_greedy=True),
                         gamma=cfg["gamma"], enable_dueling_network=cfg["dueling"], batch_size=32)
    tester.compile(dq.Adam(lr=1e-5))
    tester._bind(venv)
    tester.model.load_weights(os.path.join(cdir, "final_dqn_weights.h5f"))
    t0 = time.time()
    th = tester.test(venv, nb_episodes=eval_lattices, visualize=False, verbose=0)
    lifetime = float(np.mean(th.history["episode_lifetime"]))
    # the whole test-rate sweep the reference recorded for its agent (all_results.p), for OUR trained weights and -- same lattices, same
    # referee, same evaluation -- for the weights the REFERENCE trained with this recipe (tests/golden/keras_weights_<family>_<p>.npz)
    sweep = []
    if sweep_rates is None:
        sweep_rates = [float(x) for x in g["ref_test_p"]]
    theirs = np.load(os.path.join(ROOT, "tests", "golden", f"keras_weights_{family}_{p}.npz"))
    our_w = tester.model.get_weights()
    for rate in sweep_rates:
        row = dict(rate=rate, reference_recorded=float(g["ref_lifetime"][np.argmin(np.abs(g["ref_test_p"] - rate))]))
        venv.p_phys = venv.p_meas = rate
        for tag, w in (("ours_trained", our_w), ("reference_weights", [theirs[f"w{i}"] for i in range(12)])):
            tester.model.set_weights(w)
            row[tag] = float(np.mean(tester.test(venv, nb_episodes=eval_lattices, visualize=False, verbose=0).history["episode_lifetime"]))
        sweep.append(row)
    eval_s = time.time() - t0
    return dict(family=family, p=p, ours=ours, record=g, fixed=fixed, var=var, train_seconds=train_s, eval_seconds=eval_s, eval_lifetime=lifetime,
                eval_episodes=eval_lattices, lattices=lattices, sweep=sweep)


def compare(res):
    """Rows (checkpoint step, metric, ours, reference) + the verdicts.  Bands (stated in DESIGN.md section 5): mean_eps exact to 1e-9 at every
    episode (against the annealing rule at OUR episode boundaries -- the reference's records obey the same rule to 3e-16, tests/test_host_logic.py);
    mean_q within a factor 1.25 of the reference's at every checkpoint up to the end of the run (measured over nine replays: within 10 %), loss within a
    factor 2 (a noisy per-episode mean of squared TD errors; measured: within 1.7), the rolling lifetime within a factor 1.5 while it climbs (a learning
    curve's position in time varies run to run and the reference has ONE run; measured: 0.89 - 1.02) and -- for a FULL-LENGTH run only -- the greedy
    lifetime at the end within x2.2 of all_results.p (the spread of this build's own eight seeds is x2.6).  A shortened run's agent is not compared with
    the record of the finished one: its row is listed as "not judged".  (Round 4 asserted x2 / x4 / x3.)"""
    ours, g, var = res["ours"], res["record"], res["var"]
    S = np.array(ours["nb_steps"])
    rows, ok = [], True
    hi, lo, n_anneal, warm = var["max_eps"], var["final_eps"], var["exploration_fraction"], var["learning_starts"]
    eps = lambda s: np.maximum(lo, hi - (hi - lo) * s / float(n_anneal))
    me = np.array(ours["mean_eps"], dtype=np.float64)
    worst = 0.0
    prev = 0
    for i, s_end in enumerate(S):
        s = np.arange(prev, s_end)
        s = s[s > warm]
        prev = s_end
        if len(s):
            worst = max(worst, abs(float(eps(s).mean()) - me[i]))
        else:
            ok &= bool(np.isnan(me[i]))
    rows.append(("all", "mean_eps: max |ours - annealing rule| over the episodes", worst, 0.0, worst < 1e-9))
    ok &= worst < 1e-9
    last = int(S[-1])
    for at in CHECKPOINTS:
        if at > last:
            break
        for key, band, fn in (("mean_q", 1.25, curve_at), ("loss", 2.0, curve_at), ("episode_lifetimes_rolling_avg", 1.5, last_at)):
            a = fn(S, ours[key], at)
            b = fn(g["nb_steps"], g[key], at)
            good = (not np.isnan(a)) and (not np.isnan(b)) and b / band <= a <= b * band
            if key == "episode_lifetimes_rolling_avg" and at < 50000:
                good = True                                        # (both are the random policy's lifetime there; listed, not judged)
            rows.append((at, key, a, b, good))
            ok &= good
    ref_life = float(g["ref_lifetime"][np.argmin(np.abs(g["ref_test_p"] - float(res["p"])))])
    full = last >= 0.95 * res["fixed"]["max_timesteps"] and res["fixed"]["max_timesteps"] >= 900000
    # (one run against one run: eight seeds of THIS build's replay gave 13.7 k ... 36.1 k, median 21.2 k, at p = 0.001 against the reference's 27.7 k --
    # profiles/r04_replay_d5_x_0.001_*.json --, the same recipe's agents differ by x2.6 among themselves)
    if full:
        good = ref_life / 2.2 <= res["eval_lifetime"] <= ref_life * 2.2
        rows.append((last, f"greedy lifetime at p = {res['p']} ({res['eval_episodes']} episodes; reference: all_results.p, 101 episodes)", res["eval_lifetime"], ref_life, good))
        ok &= good
    else:
        rows.append((last, f"greedy lifetime at p = {res['p']} after {last} of {res['fixed']['max_timesteps']} steps (NOT JUDGED: the record is the finished agent's)",
                     res["eval_lifetime"], ref_life, True))
    return rows, bool(ok)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", default="d5_x")
    ap.add_argument("--p", default="0.001")
    ap.add_argument("--max-steps", type=int, default=0)
    ap.add_argument("--lattices", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("--seed", default="20181012,7", help="Philox key of the run (two integers)")
    args = ap.parse_args()
    res = run(args.family, args.p, args.max_steps or None, args.lattices, seed=tuple(int(x) for x in args.seed.split(",")))
    rows, ok = compare(res)
    print(f"replay of trained_models/{args.family}/{args.p}: {res['ours']['nb_steps'][-1]} steps, {len(res['ours']['nb_steps'])} episodes in "
          f"{res['train_seconds']:.1f} s (reference: {int(res['record']['nb_steps'][-1])} steps, {len(res['record']['nb_steps'])} episodes, "
          f"{float(res['record']['duration'].sum()) / 3600:.2f} h on 4 CPU cores)")
    for at, key, a, b, good in rows:
        print(f"  {str(at):>8}  {key:<60} ours {a:12.5g}   reference {b:12.5g}   {'ok' if good else 'OUTSIDE THE BAND'}")
    for r in res["sweep"]:
        print(f"  greedy lifetime at test rate {r['rate']:.3f}: our trained agent {r['ours_trained']:10.1f}   the reference's weights on this environment "
              f"{r['reference_weights']:10.1f}   recorded by the reference {r['reference_recorded']:10.1f}")
    print("verdict:", "within the bands" if ok else "OUTSIDE")
    if args.out:
        with open(args.out, "w") as f:
            json.dump(dict(family=args.family, p=args.p, steps=int(res["ours"]["nb_steps"][-1]), episodes=len(res["ours"]["nb_steps"]),
                           train_seconds=res["train_seconds"], eval_seconds=res["eval_seconds"], eval_lifetime=res["eval_lifetime"],
                           rows=[[str(a), k, float(x), float(y), bool(gd)] for a, k, x, y, gd in rows], sweep=res["sweep"], within_bands=ok), f, indent=1)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
