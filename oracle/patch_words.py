"""Patch words of an observation, restated from the reference's embedding in plain loops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  ``ENV`` = /root/reference/example_notebooks/Environments.py,
``FL`` = /root/reference/cluster_scripts/d5_dp/Function_Library.py.

The reference's observation is ``volume_depth`` planes from ``padding_syndrome`` (ENV:273-299) followed by
``n_action_layers`` planes from ``padding_actions`` (ENV:301-314), each (2d+1) x (2d+1).  Its first consumer is
``Conv2D(64, 3, strides=2)`` (FL:353): output pixel (oy, ox), 0 <= oy, ox < d, multiplies the 3 x 3 patch whose top-left cell
is (2 oy, 2 ox).  Of that patch

  * on a syndrome plane the four CORNERS (2 (oy + dy), 2 (ox + dx)), dy, dx in {0, 1}, are the even-even cells where
    ``padding_syndrome`` copies ``syndrome_in[x/2, y/2]`` (ENV:292-294): grid cells (oy + dy, ox + dx) of the faulty syndrome;
    the other five cells hold the decoration of ENV:284-298, which does not depend on the syndrome;
  * on an action plane the CENTRE (2 oy + 1, 2 ox + 1) is the odd-odd cell ``padding_actions`` sets for action index
    oy * d + ox (ENV:309-312); its other eight cells are never set.

So the observation is determined by, and determines, d * d words (include/deepq_hip.h ``dq_env_patch_output``):

    word[oy * d + ox] bit 4 j + 2 dy + dx     = obs[j, 2 (oy + dy), 2 (ox + dx)]          j < volume_depth
                      bit 4 volume_depth + l  = obs[volume_depth + l, 2 oy + 1, 2 ox + 1]  l < n_action_layers

``words_of`` / ``observation_of`` below are that definition and its inverse, cell by cell; ``constant_cells`` lists the cells the words
do not carry together with the value every observation has there (checked against every observation of every golden trace by
tests/test_oracle_env.py).
"""
import numpy as np


def static_plane(d):
    """padding_syndrome(zeros), ENV:284-298, cell by cell."""
    n = 2 * d + 1
    out = np.zeros((n, n), dtype=np.uint8)
    for x in range(n):
        for y in range(n):
            if (x == 0 or x == 2 * d) and y % 2 == 1:
                out[x, y] = 1
            if (y == 0 or y == 2 * d) and x % 2 == 1:
                out[x, y] = 1
            if x % 2 == 1 and y % 2 == 1 and (x + y) % 4 == 0:
                out[x, y] = 1
    return out


def words_of(obs, d, depth, layers):
    """obs: uint8 [depth + layers, 2d+1, 2d+1] -> list of d * d Python ints (the words)."""
    words = []
    for oy in range(d):
        for ox in range(d):
            w = 0
            for j in range(depth):
                for dy in (0, 1):
                    for dx in (0, 1):
                        w |= int(obs[j, 2 * (oy + dy), 2 * (ox + dx)]) << (4 * j + 2 * dy + dx)
            for l in range(layers):
                w |= int(obs[depth + l, 2 * oy + 1, 2 * ox + 1]) << (4 * depth + l)
            words.append(w)
    return words


def observation_of(words, d, depth, layers):
    """The inverse: the observation the reference would have built."""
    n = 2 * d + 1
    obs = np.zeros((depth + layers, n, n), dtype=np.uint8)
    st = static_plane(d)
    for j in range(depth):
        obs[j] = st
    for oy in range(d):
        for ox in range(d):
            w = int(words[oy * d + ox])
            for j in range(depth):
                for dy in (0, 1):
                    for dx in (0, 1):
                        obs[j, 2 * (oy + dy), 2 * (ox + dx)] = (w >> (4 * j + 2 * dy + dx)) & 1
            for l in range(layers):
                obs[depth + l, 2 * oy + 1, 2 * ox + 1] = (w >> (4 * depth + l)) & 1
    return obs


def words_array(obs_batch, d, depth, layers, stride=None):
    """A batch [..., C, 2d+1, 2d+1] -> int32 [..., stride] (stride >= d * d, the rest zero): the device layout."""
    lead = obs_batch.shape[:-3]
    flat = obs_batch.reshape((-1,) + obs_batch.shape[-3:])
    stride = d * d if stride is None else stride
    out = np.zeros((flat.shape[0], stride), dtype=np.uint32)
    for i in range(flat.shape[0]):
        out[i, :d * d] = words_of(flat[i], d, depth, layers)
    return out.view(np.int32).reshape(lead + (stride,))
