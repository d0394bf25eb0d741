"""Micro-benchmark of the Q-network kernels on the GPU box (not part of the product).

    python tools/qnet_bench.py [config=c3] [batch=4096] [iters=50]

Times inference forward, training forward and backward of the fused path and of the per-layer path with
events on torch's current stream (the stream the library launches on), and prints the max difference of
the two paths and of each against the float64 oracle on a small slice.
"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dqn_oracle as O  # noqa: E402

dq = importlib.import_module("deepq-decoding_amd")
C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]
SHAPES = {"c1": ((4, 7, 7), 10), "c2": ((6, 11, 11), 26), "c3": ((7, 11, 11), 51), "c5": ((9, 15, 15), 99)}


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    shape, A = SHAPES[name]
    spec = O.QNetSpec(shape, C_LAYERS, FF_LAYERS, A)
    net = dq.QNetwork(shape, C_LAYERS, FF_LAYERS, A, max_batch=batch)
    params = net.init_params((11, 22))
    rng = np.random.RandomState(5)
    flat = params.cpu().numpy().copy()
    flat += (rng.randn(flat.size) * 0.02).astype(np.float32)
    params.copy_(torch.from_numpy(flat))
    obs_np = (rng.rand(batch, *shape) < 0.3).astype(np.uint8)
    obs = torch.from_numpy(obs_np).cuda()
    idx = torch.from_numpy(rng.permutation(batch).astype(np.int32)).cuda()
    dq_np = (rng.randn(batch, A) / batch).astype(np.float32)
    dqt = torch.from_numpy(dq_np).cuda()
    seed, t, base = (3, 4), 5, 0
    macs = spec_macs(shape, A)
    gf = 2 * macs * batch / 1e9
    print(f"{name} batch {batch}: forward {gf:.3f} GFLOP, fused supported: {net.fused_supported}")

    res = {}
    for fused in ([1, 0] if net.fused_supported else [0]):
        net.set_fused(fused)
        tag = "fused" if fused else "layer"
        q_inf = net.forward(params, obs).cpu().numpy()
        q_idx = net.forward(params, obs, index=idx).cpu().numpy()
        q_tr = net.forward(params, obs, training=True, seed=seed, t=t, sample_base=base).cpu().numpy()
        g = net.backward(params, dqt).cpu().numpy()
        res[tag] = (q_inf, q_tr, g)
        assert np.array_equal(q_idx, q_inf[idx.cpu().numpy()]), "gathered forward differs from the direct one"
        n = min(batch, 48)
        q_ref, _ = O.forward(spec, flat, obs_np[:n])
        keep = O.dropout_keep_mask(seed, t, base + np.arange(n), 512, 0.2)
        q_ref_tr, _ = O.forward(spec, flat, obs_np[:n], training=True, keep_masks=[keep])
        print(f"[{tag}] err vs oracle: inference {np.abs(q_inf[:n] - q_ref).max():.2e}  training {np.abs(q_tr[:n] - q_ref_tr).max():.2e}")
        t_inf = timeit(lambda: net.forward(params, obs), iters)
        t_gat = timeit(lambda: net.forward(params, obs, index=idx), iters)
        t_tr = timeit(lambda: net.forward(params, obs, training=True, seed=seed, t=t, sample_base=base), iters)

        def fb():
            net.forward(params, obs, training=True, seed=seed, t=t, sample_base=base)
            net.backward(params, dqt)
        t_fb = timeit(fb, iters)
        print(f"[{tag}] inference {t_inf:7.1f} us ({gf / t_inf * 1e3:6.1f} TF/s)  gathered {t_gat:7.1f} us  training fwd {t_tr:7.1f} us  "
              f"backward {t_fb - t_tr:7.1f} us ({2 * gf / max(t_fb - t_tr, 1e-9) * 1e3:6.1f} TF/s)")
    if len(res) == 2:
        for i, what in enumerate(["inference Q", "training Q", "gradient"]):
            a, b = res["fused"][i], res["layer"][i]
            print(f"fused vs layer, {what}: max abs diff {np.abs(a - b).max():.3e} (scale {np.abs(b).max():.3e})")


def spec_macs(shape, A):
    c, h, w = shape
    macs = 0
    for f, k, s in C_LAYERS:
        oh, ow = (h - k) // s + 1, (w - k) // s + 1
        macs += oh * ow * f * k * k * c
        c, h, w = f, oh, ow
    n = c * h * w
    for u, _ in FF_LAYERS:
        macs += n * u
        n = u
    return macs + n * A + A * (A + 1)


if __name__ == "__main__":
    main()
