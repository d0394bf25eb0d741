"""CPU oracle for the DeepQ-Decoding hot path  --  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import, call, link or execute it, and there only as the checker / the timed CPU
baseline, never as the thing that produces product results.  The product path
(``deepq-decoding_amd``) raises if its HIP library is missing; it never falls
back to this code.

Pinning status
--------------
* Environment half (``lattice.py``, ``env_oracle.py``, ``env_oracle.c``,
  ``referee.py``, ``philox.py``): PINNED.  Checked against golden vectors made
  by importing the reference's own ``Environments.py`` /
  ``Function_Library.py`` in the build container under an injected RNG stream
  (``tools/gen_golden.py`` -> ``tests/golden/*.npz``) and against the one
  known-answer vector the reference documents (README.md:712-780).
* DQN half (``dqn_oracle.py``): PARITY UNPINNED at source level.  The
  reference's agent is an un-vendored, modified keras-rl fork on Keras/TF
  (README.md:33); none of it can be imported here.  The oracle restates the
  published keras-rl 0.4.x / Keras 2.2 update rule in float64 numpy and is
  cross-checked against torch-CPU autograd only.
"""
