"""Shared helpers of the shipped-weight tests: the reference's documented production-decoding example and real observations.

README_SLICES is DATA: the five faulty syndrome slices printed in /root/reference/README.md:730-780 (= notebook 3, cells 16-22), as
the coordinates of their ones.  The README decodes that volume with trained_models/d5_x/0.007/final_dqn_weights.h5f
(README.md:562-564, 624; committed as tests/golden/keras_weights_d5_x_0.007.npz) and prints `corrections == [21]`
(README.md:797-829): the one deterministic known answer the reference holds for the Q-network forward (layer layout, channels_first
Flatten, dueling head, greedy selection)."""
import numpy as np

README_D = 5
README_SLICES = [
    [],                                  # syndrome slice 1
    [(1, 2), (2, 4)],                    # slice 2 (measurement errors)
    [(4, 1), (5, 2)],                    # slices 3-5: the X flip on qubit 21 = (4, 1)
    [(4, 1), (5, 2)],
    [(0, 1), (4, 1), (5, 2)],            # slice 5 (one measurement error)
]
README_CORRECTIONS = [21]
C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]
CONFIGS = {
    "d5_x": dict(d=5, error_model="X", use_Y=False, volume_depth=5),
    "d5_dp": dict(d=5, error_model="DP", use_Y=False, volume_depth=5),
}


def readme_faulty_syndromes():
    out = np.zeros((README_D, README_D + 1, README_D + 1), dtype=int)
    for j, ones in enumerate(README_SLICES):
        for a, b in ones:
            out[j, a, b] = 1
    return out


def readme_input_state(padding_syndrome):
    """README.md:786-792: a zeroed (d+1, 2d+1, 2d+1) volume with the padded slices in planes 0..d-1."""
    d = README_D
    state = np.zeros((d + 1, 2 * d + 1, 2 * d + 1), int)
    for j, s in enumerate(readme_faulty_syndromes()):
        state[j, :, :] = padding_syndrome(s)
    return state


def readme_decode_loop(forward, padding_actions, identity_index, input_state):
    """README.md:797-818, statement for statement (including its call padding_actions(corrections) with the LIST of action indices,
    which Environments.py:301-314 reads as a 0/1 vector: [21] marks qubit 0)."""
    corrections = []
    still_decoding = True
    while still_decoding:
        action = forward(input_state)
        if action not in corrections and action != identity_index:
            corrections.append(action)
            input_state[README_D, :, :] = padding_actions(corrections)
        else:
            still_decoding = False
    return corrections


def shipped_weights(family, p):
    from conftest import load_golden
    fx = load_golden(f"keras_weights_{family}_{p}")
    w = [fx[f"w{i}"] for i in range(12)]
    return w, np.concatenate([x.reshape(-1) for x in w]).astype(np.float32)


def real_observations(family, p, n, steps=6, seed=(0x5EED, 0xD0DEC0DE)):
    """n real observations of the family's environment at error rate p: the C oracle's lattices (n / steps of them) after 1 .. steps
    moves of the uniform-over-legal policy -- syndrome planes AND populated action planes, as the agent sees them."""
    from oracle import c_oracle
    per = -(-n // steps)
    env = c_oracle.COracleEnv(n_envs=per, p_phys=p, p_meas=p, seed=seed, **CONFIGS[family])
    env.reset()
    out = []
    for t in range(steps):
        out.append(env.obs.copy())
        env.step(env.policy_uniform_legal(t), auto_reset=True)
    return np.concatenate(out)[:n]
