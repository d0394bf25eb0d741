"""float64 numpy restatement of the DQN half of the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED at source level: the reference's
agent is the un-vendored, modified keras-rl fork github.com/R-Sweke/keras-rl (README.md:33; no commit
pinned) on Keras 2.2.2 / TensorFlow 1.x, none of which is importable here.  What is restated below is
the published keras-rl 0.4.x ``DQNAgent`` / Keras 2.2 arithmetic as used by the reference's call sites
(``TRAIN`` = /root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py):

  network      TRAIN:61-90 build_convolutional_nn: Conv2D(valid, channels_first)+ReLU ..., Flatten,
               Dense+ReLU+Dropout ..., Dense(nb_actions) linear; keras-rl dueling head
               (enable_dueling_network=True, TRAIN:127; dueling_type 'avg'): Dense(nb_actions+1) then
               Q = y[:,0:1] + y[:,1:] - mean(y[:,1:], axis=1).  Tensor shapes confirmed by the shipped
               weights (trained_models/*/*/final_dqn_weights.h5f: (3,3,C,64) (2,2,64,32) (2,2,32,32)
               (288,512) (512,A) (A,A+1); kernels HWIO, dense (in,out)).
  update       double DQN (keras-rl default enable_double_dqn=True; paper TEX:493,632):
               a* = argmax_a Q_online(s1);  y = r + gamma * (1 - terminal) * Q_target(s1)[a*];
               loss = mean_b 0.5 * (y_b - Q_online(s0_b)[a_b])^2   (delta_clip = inf -> huber == 0.5 x^2);
               metric mean_q = mean_b max_a Q_online(s0_b) of the TRAINING forward (dropout active).
  optimizer    Keras Adam (TRAIN:130): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t*m/(sqrt(v)+1e-7).
  policy       LinearAnnealedPolicy(EpsGreedyQPolicy) TRAIN:110-114: eps = max(min, max_ - (max_-min)*step/nb_steps).

Flat parameter layout (shared with the HIP library): for each layer kernel then bias, Keras order.
Activations are described in NHWC here only where noted; the math is layout-free.
"""
import numpy as np

from . import philox


class QNetSpec:
    def __init__(self, input_shape, c_layers, ff_layers, n_actions, dueling=True, dueling_mean="row"):
        # dueling_mean: which mean the dueling head's Lambda subtracts.  "row" (default, what the product implements): the mean over the ACTIONS of each
        # sample, Q(s, a) = V(s) + A(s, a) - mean_a' A(s, a') -- equation (9) of Wang et al. 2016 (arXiv:1511.06581), the paper the reference cites for its
        # dueling head (README.md:261, manuscript TEX:501,630), and keras-rl master's `K.mean(a[:, 1:], axis=1, keepdims=True)`.  "batch": the form upstream
        # keras-rl 0.4.2 is reported to have shipped with, `K.mean(a[:, 1:], keepdims=True)` -- no axis, i.e. ONE mean over batch AND actions, which makes a
        # state's Q-values depend on the other states of the batch.  The reference's fork (github.com/R-Sweke/keras-rl, README.md:33) is not in the tree, so
        # which of the two its training ran is UNPINNED; the two coincide for a batch of one (every acting / test forward, every shipped-agent check) and
        # differ in the 32-sample training forwards, targets and gradient (tests/test_oracle_dqn.py prints by how much).  DESIGN.md section 6.
        assert dueling_mean in ("row", "batch")
        self.dueling_mean = dueling_mean
        self.input_shape = tuple(int(x) for x in input_shape)          # (C, H, W)
        self.c_layers = [tuple(int(x) for x in l) for l in c_layers]   # [filters, kernel, stride]
        self.ff_layers = [(int(l[0]), float(l[1])) for l in ff_layers]  # [units, dropout rate]
        self.n_actions, self.dueling = int(n_actions), bool(dueling)
        self.layers = []            # (kind, dict)
        c, h, w = self.input_shape
        for (f, k, s) in self.c_layers:
            oh, ow = (h - k) // s + 1, (w - k) // s + 1
            self.layers.append(("conv", dict(cin=c, cout=f, k=k, s=s, ih=h, iw=w, oh=oh, ow=ow)))
            c, h, w = f, oh, ow
        n = c * h * w
        self.flat = n
        for (u, rate) in self.ff_layers:
            self.layers.append(("dense", dict(nin=n, nout=u, relu=True, dropout=rate)))
            n = u
        self.layers.append(("dense", dict(nin=n, nout=self.n_actions, relu=False, dropout=0.0)))
        if self.dueling:
            self.layers.append(("dense", dict(nin=self.n_actions, nout=self.n_actions + 1, relu=False, dropout=0.0)))

    def param_shapes(self):
        out = []
        for kind, L in self.layers:
            if kind == "conv":
                out.append(((L["k"], L["k"], L["cin"], L["cout"]), (L["cout"],)))
            else:
                out.append(((L["nin"], L["nout"]), (L["nout"],)))
        return out

    @property
    def n_params(self):
        return sum(int(np.prod(k)) + int(np.prod(b)) for k, b in self.param_shapes())

    def split(self, flat):
        out, o = [], 0
        for k, b in self.param_shapes():
            nk, nb = int(np.prod(k)), int(np.prod(b))
            out.append((flat[o:o + nk].reshape(k), flat[o + nk:o + nk + nb]))
            o += nk + nb
        assert o == len(flat)
        return out

    def forward_macs(self):
        m = 0
        for kind, L in self.layers:
            m += L["oh"] * L["ow"] * L["cout"] * L["k"] * L["k"] * L["cin"] if kind == "conv" else L["nin"] * L["nout"]
        return m


def glorot_init(spec, seed, stream=philox.STREAM_INIT):
    """Keras default initialisers (glorot_uniform kernels, zero biases) driven by the Philox INIT stream:
    element i of layer l gets u = (word + 0.5) / 2^32 with word = Philox(key=seed, ctr=(i>>2, 0, l, stream<<16))[i&3]."""
    flat = np.zeros(spec.n_params, dtype=np.float32)
    o = 0
    for l, (k, b) in enumerate(spec.param_shapes()):
        nk = int(np.prod(k))
        if len(k) == 4:
            fan_in, fan_out = k[0] * k[1] * k[2], k[0] * k[1] * k[3]
        else:
            fan_in, fan_out = k
        limit = np.sqrt(6.0 / (fan_in + fan_out))
        idx = np.arange(nk, dtype=np.uint64)
        words = philox.philox4x32_np((idx >> np.uint64(2)).astype(np.uint32), 0, l, stream << 16, seed)
        w = np.stack(words, axis=1)[np.arange(nk), (idx & np.uint64(3)).astype(np.int64)]
        u = (w.astype(np.float64) + 0.5) / 4294967296.0
        flat[o:o + nk] = ((2.0 * u - 1.0) * limit).astype(np.float32)
        o += nk + int(np.prod(b))
    return flat


def dropout_keep_mask(seed, t, sample_ids, n_units, rate):
    """keep[b, j] for update counter t.  One Philox call covers eight consecutive units, 16 bits per decision: unit j draws half-word
    j & 7 (bits 16 (h & 1) .. + 15 of word h >> 1) of Philox(key=seed, ctr=(t_lo, t_hi, sample_id, (j>>3) | DROPOUT<<16)); dropped iff
    the draw < ceil(rate * 2^16)."""
    sample_ids = np.asarray(sample_ids, dtype=np.uint32)
    j = np.arange(n_units, dtype=np.uint32)
    words = philox.philox4x32_np(int(t) & philox.MASK, (int(t) >> 32) & philox.MASK, sample_ids[:, None],
                                 (j[None, :] >> 3) | (philox.STREAM_DROPOUT << 16), seed)
    w = np.stack(words, axis=-1)
    h = (j & 7).astype(np.int64)
    sel = np.take_along_axis(w, np.broadcast_to((h >> 1)[None, :, None], w.shape[:2] + (1,)), axis=-1)[..., 0]
    draw = (sel.astype(np.uint64) >> (16 * (h & 1)).astype(np.uint64)[None, :]) & np.uint64(0xffff)
    return draw >= np.uint64(philox.threshold16(rate))


def _im2col(x, k, s):
    """x (B,C,H,W) -> (B, OH, OW, k*k*C) with the last axis ordered (ky, kx, c) = Keras HWIO flattening."""
    B, C, H, W = x.shape
    oh, ow = (H - k) // s + 1, (W - k) // s + 1
    cols = np.zeros((B, oh, ow, k, k, C), dtype=x.dtype)
    for ky in range(k):
        for kx in range(k):
            cols[:, :, :, ky, kx, :] = x[:, :, ky:ky + s * oh:s, kx:kx + s * ow:s].transpose(0, 2, 3, 1)
    return cols.reshape(B, oh, ow, k * k * C)


def _dueling_mean(spec, adv):
    """The mean the dueling head subtracts from the advantages (QNetSpec.dueling_mean): per sample, or one number for the whole batch."""
    if getattr(spec, "dueling_mean", "row") == "batch":
        return adv.mean(keepdims=True)
    return adv.mean(axis=1, keepdims=True)


def forward(spec, flat_params, obs, training=False, keep_masks=None):
    """Returns (Q (B,A) float64, cache).  keep_masks: list of boolean (B, units) arrays, one per dropout layer
    with rate > 0 (training only)."""
    P = spec.split(np.asarray(flat_params, dtype=np.float64))
    x = np.asarray(obs, dtype=np.float64)
    cache = dict(layers=[])
    mi = 0
    flat_done = False
    for (kind, L), (Wk, bk) in zip(spec.layers, P):
        if kind == "conv":
            cols = _im2col(x, L["k"], L["s"])                                  # (B,OH,OW,K)
            z = cols @ Wk.reshape(-1, L["cout"]) + bk                          # (B,OH,OW,Cout)
            y = np.maximum(z, 0.0)
            cache["layers"].append(dict(kind="conv", cols=cols, y=y, z=z, x_shape=x.shape))
            x = y.transpose(0, 3, 1, 2)                                        # back to (B,C,H,W)
        else:
            if not flat_done:
                x = x.reshape(x.shape[0], -1)                                  # Keras Flatten, channels_first order
                flat_done = True
            z = x @ Wk + bk
            y = np.maximum(z, 0.0) if L["relu"] else z
            keep = None
            if training and L["dropout"] > 0.0:
                keep = keep_masks[mi]
                mi += 1
                y_out = np.where(keep, y / (1.0 - L["dropout"]), 0.0)           # K.dropout: x / keep_prob * mask
            else:
                y_out = y
            cache["layers"].append(dict(kind="dense", x=x, y=y, z=z, keep=keep, rate=L["dropout"], relu=L["relu"]))
            x = y_out
    if spec.dueling:
        q = x[:, 0:1] + x[:, 1:] - _dueling_mean(spec, x[:, 1:])
    else:
        q = x
    cache["head_in"] = x
    return q, cache


def fragile_samples(cache, thr=2e-6, rel=None):
    """Samples with a ReLU pre-activation within `thr` of 0: an fp32 implementation may put it on the other side of the ReLU, which
    changes that sample's gradient by a finite amount (not a round-off).  Tests on large batches give such samples dq = 0.  thr is the
    pre-activation error bound of the HIP paths on unit-scale weights (2e-6 at |z| <~ 4: ten times their measured error against float64); with rel the threshold is rel * max |z|
    per layer instead -- the error of an fp32 contraction is relative to the layer's magnitude (trained weights: |z| up to ~70; the
    measured error of the HIP paths there is ~4e-7 of the largest value, tests/test_shipped_weights.py)."""
    bad = None
    for C in cache["layers"]:
        if C["kind"] == "conv" or C["relu"]:
            z = np.abs(C["z"])
            t = rel * float(z.max()) if rel else thr
            b = (z.reshape(z.shape[0], -1) < t).any(axis=1)
            bad = b if bad is None else (bad | b)
    return bad


def backward(spec, flat_params, cache, dq):
    """Gradient of sum(dq * Q) w.r.t. the flat parameters (float64)."""
    P = spec.split(np.asarray(flat_params, dtype=np.float64))
    dq = np.asarray(dq, dtype=np.float64)
    if spec.dueling:
        A = spec.n_actions
        g = np.zeros((dq.shape[0], A + 1))
        g[:, 0] = dq.sum(axis=1)
        if getattr(spec, "dueling_mean", "row") == "batch":
            g[:, 1:] = dq - dq.sum() / (A * dq.shape[0])                       # one mean over batch and actions: every sample receives the batch's total
        else:
            g[:, 1:] = dq - dq.sum(axis=1, keepdims=True) / A
    else:
        g = dq
    grads = [None] * len(P)
    for li in range(len(P) - 1, -1, -1):
        (kind, L), (Wk, bk), C = spec.layers[li], P[li], cache["layers"][li]
        if kind == "dense":
            if C["keep"] is not None:
                g = np.where(C["keep"], g / (1.0 - C["rate"]), 0.0)
            if C["relu"]:
                g = g * (C["y"] > 0.0)
            grads[li] = (C["x"].T @ g, g.sum(axis=0))
            g = g @ Wk.T
        else:
            B = C["y"].shape[0]
            if g.ndim == 2:                                                    # coming from Flatten: (B, C*OH*OW)
                g = g.reshape(B, L["cout"], L["oh"], L["ow"]).transpose(0, 2, 3, 1)
            g = g * (C["y"] > 0.0)                                             # (B,OH,OW,Cout)
            K = L["k"] * L["k"] * L["cin"]
            grads[li] = ((C["cols"].reshape(-1, K).T @ g.reshape(-1, L["cout"])).reshape(Wk.shape), g.sum(axis=(0, 1, 2)))
            if li > 0:
                dcols = g @ Wk.reshape(K, L["cout"]).T                         # (B,OH,OW,K)
                dcols = dcols.reshape(B, L["oh"], L["ow"], L["k"], L["k"], L["cin"])
                dx = np.zeros((B, L["ih"], L["iw"], L["cin"]))
                s = L["s"]
                for ky in range(L["k"]):
                    for kx in range(L["k"]):
                        dx[:, ky:ky + s * L["oh"]:s, kx:kx + s * L["ow"]:s, :] += dcols[:, :, :, ky, kx, :]
                g = dx                                                         # NHWC gradient w.r.t. previous y
    return np.concatenate([np.concatenate([gk.reshape(-1), gb.reshape(-1)]) for gk, gb in grads])


def td_targets(q_online_s1, q_target_s1, reward, terminal, gamma):
    """Double-DQN target (keras-rl DQNAgent.backward, enable_double_dqn)."""
    a_star = np.argmax(q_online_s1, axis=1)
    q = q_target_s1[np.arange(len(a_star)), a_star]
    return np.asarray(reward, dtype=np.float64) + gamma * q * (1.0 - np.asarray(terminal, dtype=np.float64))


def loss_and_grad(q_s0, action, y):
    """loss = mean_b 0.5 (y_b - Q[b,a_b])^2;  mean_q = mean_b max_a Q[b,a];  dQ = dloss/dQ."""
    B = q_s0.shape[0]
    idx = np.arange(B)
    diff = q_s0[idx, action] - y
    dq = np.zeros_like(q_s0)
    dq[idx, action] = diff / B
    return 0.5 * np.mean(diff ** 2), np.mean(q_s0.max(axis=1)), dq


def adam_step(p, g, m, v, t, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
    """Keras 2.2 Adam.get_updates; t = 1 for the first update.  Returns new (p, m, v)."""
    lr_t = lr * np.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t)
    m = beta_1 * m + (1.0 - beta_1) * g
    v = beta_2 * v + (1.0 - beta_2) * g * g
    return p - lr_t * m / (np.sqrt(v) + epsilon), m, v


def annealed_eps(step, value_max, value_min, nb_steps):
    """keras-rl LinearAnnealedPolicy.get_current_value (training)."""
    a = -float(value_max - value_min) / float(nb_steps)
    return max(value_min, a * float(step) + float(value_max))


def select_action(q, legal_mask, eps, masked_greedy, words):
    """Restates dq_policy_select for one lattice (include/deepq_hip.h): words = 4 uint32 of the POLICY stream."""
    legal = [a for a in range(len(q)) if (legal_mask >> a) & 1]
    if q is None or int(words[1]) < philox.threshold(eps):
        return legal[philox.bounded(words[0], len(legal))]
    if masked_greedy:
        return max(legal, key=lambda a: (q[a], -a))
    return int(np.argmax(q))


def split_f16x2(x):
    """The fused chains' operand representation (csrc/qnet.h "f16x2"): x ~= h + l 2^-11 with h = f16(x) (round to nearest even) and
    l = f16((x - h) 2^11), both returned as float64 arrays holding exactly representable f16 values."""
    x = np.asarray(x, dtype=np.float32)
    h = x.astype(np.float16).astype(np.float32)
    l = ((x - h) * np.float32(2048.0)).astype(np.float16)
    return h.astype(np.float64), l.astype(np.float64)


def forward_f16x2_emulated(spec, flat_params, obs):
    """Inference forward with every contraction's two operands rounded to the f16x2 representation and the THREE products the
    kernels issue (hH + 2^-11 (hL + lH); the binary observation of the first convolution is exact) accumulated EXACTLY (float64);
    activations are rounded to f32 between layers.  This is the operand scheme's own error, i.e. the best case of the fused HIP
    chains (which add their f32 accumulation error on top); tests use it to state the tolerance the scheme can meet at the
    magnitudes of the reference's trained agents (|Q| 10-70), where an ABSOLUTE 1e-5 is below what any fp32 path delivers."""
    P = spec.split(np.asarray(flat_params, dtype=np.float32))
    x = np.asarray(obs, dtype=np.float64)
    flat_done = False

    def mm(a, w):
        ah, al = split_f16x2(a)
        wh, wl = split_f16x2(w)
        return ah @ wh + (ah @ wl + al @ wh) / 2048.0

    for (kind, L), (Wk, bk) in zip(spec.layers, P):
        b64 = bk.astype(np.float64)
        if kind == "conv":
            cols = _im2col(x, L["k"], L["s"])
            z = mm(cols.reshape(-1, cols.shape[-1]), Wk.reshape(-1, L["cout"])).reshape(cols.shape[:3] + (L["cout"],)) + b64
            x = np.maximum(z, 0.0).astype(np.float32).astype(np.float64).transpose(0, 3, 1, 2)
        else:
            if not flat_done:
                x = x.reshape(x.shape[0], -1)
                flat_done = True
            z = mm(x, Wk) + b64
            x = (np.maximum(z, 0.0) if L["relu"] else z).astype(np.float32).astype(np.float64)
    if spec.dueling:
        return x[:, 0:1] + x[:, 1:] - _dueling_mean(spec, x[:, 1:])
    return x
