from .dqn import DQNAgent  # noqa: F401
