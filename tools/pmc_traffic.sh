#!/bin/bash
# HBM-side traffic per kernel launch from the L2 fabric counters, as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE and WRITE_SIZE
# in SEPARATE --pmc passes (they do not fit one pass), unit KB, gfx950 correction: FETCH_SIZE counts 128-byte requests at 64 bytes
# -> doubled.  Usage (GPU box, repo root): tools/pmc_traffic.sh [mode [config [extra bench args]]]
#   -> gpurun_out/pmc_traffic_<mode>_<config>.json, stamped with the sha256 of the kernel sources it measured (copy into profiles/:
#   bench.py prints `roofline.traffic` from it only while the sources still hash to that value).
root="${GRAFT_REPO_ROOT:-$PWD}"; cd "$root"; mkdir -p gpurun_out
mode="${1:-loop}"; config="${2:-c3}"; shift; shift
# (--mode env launches 64 agent steps at a time -- dq_env_act_steps --: 2560 steps = 40 full launches, 64 untimed ones in front)
steps=40; warmup=5; if [ "$mode" = env ]; then steps=2560; warmup=64; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "gpurun_out/pmc_$c"
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $c --kernel-trace -d "$root/gpurun_out/pmc_$c" -o pmc --output-format csv -- python "$root/bench.py" --mode "$mode" --config "$config" --steps $steps --warmup $warmup --no-cpu-baseline --ratio-steps 0 "$@" > "$root/gpurun_out/pmc_$c.log" 2>&1)
done
MODE="$mode" CONFIG="$config" EXTRA="$*" python3 - <<'PY'
import csv, glob, json, collections, importlib, os, sys
sys.path.insert(0, os.getcwd())
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg, cnt = collections.Counter(), collections.Counter()
    rows = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            rows[k].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    # the LAST 40 launches of every kernel = those of the 40 timed steps: the set-up's launches of the same kernel (ring fill: one-job forwards of
    # conv_wave_kernel / dense_chain_kernel, resets of env_kernel) come first and are not what `roofline.traffic` describes
    for k, v in rows.items():
        v.sort()
        for _, x in v[-40:]:
            agg[k] += x; cnt[k] += 1
    for k in agg:
        out[k][c + "_KB_per_launch"] = agg[k] / cnt[k]
        out[k]["launches_" + c] = cnt[k]
for k, v in out.items():
    f, w = v.get("FETCH_SIZE_KB_per_launch", 0.0), v.get("WRITE_SIZE_KB_per_launch", 0.0)
    v["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0       # gfx950: FETCH_SIZE reads half of a wide coalesced stream
    v["hbm_bytes_per_launch_raw"] = (f + w) * 1024.0
mode, config = os.environ["MODE"], os.environ["CONFIG"]
digest = importlib.import_module("deepq-decoding_amd.bench_loop").csrc_digest()
extra = os.environ.get("EXTRA", "").split()
arg = lambda name, default: int(extra[extra.index(name) + 1]) if name in extra else default
shape = {"minibatch": arg("--minibatch", 0), "updates_per_step": arg("--updates-per-step", 1), "lattices": arg("--lattices", 0)}      # 0 = the configuration's default
path = f"gpurun_out/pmc_traffic_{mode}_{config}.json"
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --mode {mode} --config {config} --steps 40 --warmup 5 --no-cpu-baseline --ratio-steps 0` (--mode env: 2560 / 64)",
           "correction": "MI355X_MICROARCH.md HBM section: unit KB; gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-B requests at 64 B -> x2; WRITE_SIZE uncalibrated (used as is)",
           "csrc_sha256": digest, "shape": shape, "launches_averaged": "the last 40 of each kernel (the timed steps)", "kernels": out}, open(path, "w"), indent=1)
print(path, digest[:16])
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_corrected"])[:12]:
    print(f"{k[:40]:40s} fetch {v.get('FETCH_SIZE_KB_per_launch', 0):10.1f} KB  write {v.get('WRITE_SIZE_KB_per_launch', 0):10.1f} KB  corrected {v['hbm_bytes_per_launch_corrected'] / 1e6:8.2f} MB/launch")
PY
