"""Host-side lattice helpers with the names and call signatures of the reference's Function_Library.py
(/root/reference/cluster_scripts/d5_dp/Function_Library.py:13-326, "FL").

These are the small numpy utilities the notebooks' "production decoding" demo calls directly (building one
syndrome volume by hand, notebook 3 cells 16-26).  The training / evaluation hot path does not go through
them -- it runs in the HIP kernels (csrc/env.hip), which implement the same rules in bit-plane form.
Randomness here comes from numpy's global generator exactly like the reference's helpers.
"""
import numpy as np

from .env import generateSurfaceCodeLattice, _measurement_order  # noqa: F401


def multiplyPaulis(a, b):
    """FL:54-62: product of Pauli codes I,X,Y,Z = 0,1,2,3 up to phase == XOR of the codes."""
    return int(a) ^ int(b)


def generate_X_error(d, p_phys):
    """FL:105-122: independent X flips; draws d*d uniforms in row-major order."""
    return (np.random.rand(d * d).reshape(d, d) < p_phys).astype(int)


def generate_DP_error(d, p_phys):
    """FL:86-103: depolarising noise; a Pauli type in {1,2,3} is drawn only for the qubits that err."""
    error = np.zeros((d, d), int)
    for i in range(d):
        for j in range(d):
            if np.random.rand() < p_phys:
                error[i, j] = np.random.randint(1, 4)
    return error


def generate_IIDXZ_error(d, p_phys):
    """FL:124-150: independent X and Z flips (two uniforms per qubit, X first)."""
    u = np.random.rand(d, d, 2)
    x, z = u[..., 0] < p_phys, u[..., 1] < p_phys
    return (x * 1) ^ (z * 3)


def generate_error(d, p_phys, error_model):
    """FL:67-84."""
    if error_model == "X":
        return generate_X_error(d, p_phys)
    if error_model == "DP":
        return generate_DP_error(d, p_phys)
    if error_model == "IIDXZ":
        return generate_IIDXZ_error(d, p_phys)
    raise UnboundLocalError("local variable 'error' referenced before assignment")   # what FL:84 does for other strings


def generate_surface_code_syndrome_NoFT_efficient(error, qubits):
    """FL:152-174: type-3 plaquettes report the parity of the X components of their qubits, type-1 the Z components."""
    error = np.asarray(error).astype(int)
    d = error.shape[0]
    xbit = ((error == 1) | (error == 2)).astype(int)
    zbit = ((error == 2) | (error == 3)).astype(int)
    syndrome = np.zeros((d + 1, d + 1), int)
    for k in range(4):
        a, b, t = qubits[:, :, k, 0], qubits[:, :, k, 1], qubits[:, :, k, 2]
        np.add.at(syndrome, (a, b), np.where(t == 3, xbit, np.where(t == 1, zbit, 0)))
    return syndrome % 2


def generate_faulty_syndrome(true_syndrome, p_measurement_error):
    """FL:176-223: every live stabilizer is flipped with probability p (d*d-1 uniforms, bulk first, then the four
    boundaries); absent plaquettes stay 0."""
    true_syndrome = np.asarray(true_syndrome).astype(int)
    d = true_syndrome.shape[0] - 1
    faulty = np.zeros_like(true_syndrome)
    order = _measurement_order(d)
    flips = np.random.rand(len(order)) < p_measurement_error
    for (a, b), f in zip(order, flips):
        faulty[a, b] = true_syndrome[a, b] ^ int(f)
    return faulty


def obtain_new_error_configuration(old_configuration, new_gates):
    """FL:226-241: element-wise Pauli product; returns float64 like the reference (np.zeros default dtype)."""
    return (np.asarray(new_gates).astype(int) ^ np.asarray(old_configuration).astype(int)).astype(float)


def index_to_move(d, move_index, error_model, use_Y=True):
    """FL:243-294."""
    new_move = np.zeros((d, d))
    if error_model == "X":
        layers = 1
    elif error_model == "DP":
        layers = 3 if use_Y else 2
    else:
        print("Error model you have specified is not currently supported")
        return new_move
    if move_index < layers * d * d:
        layer, q = divmod(int(move_index), d * d)
        move_type = 1 if error_model == "X" else (layer + 1 if use_Y else (1 if layer == 0 else 3))
        new_move[q // d, q % d] = move_type
    return new_move


def generate_one_hot_labels_surface_code(error, err_model):
    """FL:296-326: homology class X + 2Z; X = parity of X components down column 0, Z = parity of Z components along row 0."""
    error = np.asarray(error).astype(int)
    X = int((((error[:, 0] == 1) | (error[:, 0] == 2)).sum()) % 2)
    Z = int((((error[0, :] == 3) | (error[0, :] == 2)).sum()) % 2)
    label = np.zeros(4 if err_model in ("IIDXZ", "DP") else 2, int)
    label[X + 2 * Z] = 1
    return label
