#!/usr/bin/env python3
"""Prints the numbers DESIGN.md section 9 / README.md / profiles/README.md quote, from a measurement set directory (tools/final_measure.sh <tag> -> gpurun_out/<tag>/)."""
import glob, json, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06"
def L(name):
    try:
        return json.load(open(os.path.join(d, f"bench_{name}.json")))
    except Exception:
        return None
for f in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
    x = json.load(open(f)); r = x.get("roofline") or {}; q = x.get("reference_replay_ratio") or {}
    print(os.path.basename(f)[6:-5].ljust(30), "%.4g %s" % (x["value"], x["unit"]), "| %.2f us/step" % (1e3 * x["ms_per_step"]), "|", r.get("kernel"), "%.2f us [%.1f-%.1f]" % (r.get("avg_launch_us", 0), r.get("min_launch_us", 0), r.get("max_launch_us", 0)),
          "frac %.3f" % r.get("frac", 0), "traffic %s" % (r.get("traffic") and "%.1f MB" % (r["traffic"] / 1e6)), "| ratio32 %.4g (%.1f us/update)" % (q.get("value", 0), q.get("us_per_update", 0)), "| upd/s %.0f" % x.get("dqn_updates_per_s", 0), "tflops %.0f" % x.get("achieved_qnet_tflops_per_gpu", 0))
x = L("c3_loop")
if x:
    print(json.dumps(x["cpu_baseline"], indent=1)[:1500])
    r = x["roofline"]; print("pipe", r.get("pipe"), "vs f32", r.get("vs_f32_mfma_peak"))
for f in ("loop_c3", "loop_c2", "loop_c5", "env_c3"):
    p = os.path.join(d, f"pmc_traffic_{f}.json")
    if os.path.exists(p):
        z = json.load(open(p))
        print(f, z["csrc_sha256"][:8], {k: round(v["hbm_bytes_per_launch_corrected"] / 1e6, 1) for k, v in z["kernels"].items() if any(s in k for s in ("conv", "dense", "env", "reduce", "pack_w"))})
