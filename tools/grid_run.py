#!/usr/bin/env python3
"""Command-line front end of deepq-decoding_amd/runner.py (the reference's `python Single_Point_Training_Script.py N`, its
Start_Simulations.sh and Controller.py on one node).

    tools/grid_run.py point  <family>/<p>/config_N [--n-envs 4096] [--batch-size B] [--quiet]   one grid point on the current GPU
    tools/grid_run.py grid   <family>/<p> [--gpus 0,1,2,3,4,5,6,7] [--n-envs 4096] [--max-points K]   one grid point per GPU at a time
    tools/grid_run.py ladder <family> --fixed <fixed_config.p> [--p-list 0.001,0.003,...]        the whole error-rate ladder
    tools/grid_run.py write  <family> --fixed <fixed_config.p> --p 0.001                          the initial grid's dict files only
"""
import argparse
import importlib
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["point", "grid", "ladder", "write"])
    ap.add_argument("path")
    ap.add_argument("--n-envs", type=int, default=1)
    ap.add_argument("--batch-size", type=int, default=0)
    ap.add_argument("--sync-interval", type=int, default=0)
    ap.add_argument("--gpus", default="")
    ap.add_argument("--max-points", type=int, default=0)
    ap.add_argument("--fixed", default="")
    ap.add_argument("--p", type=float, default=0.001)
    ap.add_argument("--p-list", default="")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    runner = importlib.import_module("deepq-decoding_amd.runner")
    gpus = [int(x) for x in a.gpus.split(",")] if a.gpus else None
    extra = (["--batch-size", str(a.batch_size)] if a.batch_size else []) + (["--sync-interval", str(a.sync_interval)] if a.sync_interval else []) \
        + (["--quiet"] if a.quiet else [])
    if a.cmd == "point":
        res = runner.train_single_point(a.path, n_envs=a.n_envs, verbose=0 if a.quiet else 2, batch_size=a.batch_size or None,
                                        sync_interval=a.sync_interval or None)
        print("all_results:", res)
    elif a.cmd == "grid":
        print(runner.run_grid(a.path, gpus=gpus, n_envs=a.n_envs, max_points=a.max_points or None, extra_args=extra))
        print(runner.collect_results(a.path))
    elif a.cmd == "write":
        fixed = pickle.load(open(a.fixed, "rb"))
        print(len(runner.write_grid(a.path, fixed, a.p, 100000)), "grid points written")
    else:
        fixed = pickle.load(open(a.fixed, "rb"))
        p_list = [float(x) for x in a.p_list.split(",")] if a.p_list else None
        print(runner.run_error_rate_ladder(a.path, fixed, p_list=p_list, gpus=gpus, n_envs=a.n_envs, max_points=a.max_points or None,
                                           extra_args=extra))


if __name__ == "__main__":
    main()
