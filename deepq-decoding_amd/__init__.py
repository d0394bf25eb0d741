"""deepq-decoding_amd: MI355X-native hot path of R-Sweke/DeepQ-Decoding.

The package directory name carries a hyphen (it mirrors the reference's name), so import it with

    import importlib; dq = importlib.import_module("deepq-decoding_amd")

or put ``deepq-decoding_amd/dropin`` on ``sys.path`` and use the reference's own module names
(``Environments``, ``Function_Library``, ``rl.agents.dqn`` ...).

Everything that computes goes through ``lib/libdeepq_hip.so`` (hand-written HIP for gfx950, C ABI in
``include/deepq_hip.h``).  There is no CPU fallback: without the library or without a GPU the
compute entry points raise ``DeepQError``.
"""
from ._lib import DeepQError, LIB_PATH, lib, require_gpu  # noqa: F401


def __getattr__(name):
    # torch-dependent modules are imported lazily so that `lib()` / ABI checks work without touching torch
    if name in ("VectorEnv", "Surface_Code_Environment_Multi_Decoding_Cycles", "generateSurfaceCodeLattice"):
        from . import env
        return getattr(env, name)
    if name in ("QNetwork", "td_target", "td_loss_grad", "adam_step", "replay_sample"):
        from . import qnet
        return getattr(qnet, name)
    if name in ("DQNAgent", "SequentialMemory", "EpsGreedyQPolicy", "GreedyQPolicy", "LinearAnnealedPolicy", "BoltzmannQPolicy",
                "FileLogger", "Adam", "build_convolutional_nn", "ConvQModel", "History"):
        from . import agent
        return getattr(agent, name)
    if name in ("dist", "hdf5_reader", "weights_io", "function_library", "runner"):
        import importlib
        return importlib.import_module("." + name, __name__)
    if name == "FeedForwardReferee":
        from .referee import FeedForwardReferee
        return FeedForwardReferee
    if name == "DQNCore":
        from .core import DQNCore
        return DQNCore
    raise AttributeError(name)
