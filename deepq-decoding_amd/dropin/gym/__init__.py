"""Stand-in for the two gym names the reference uses (Environments.py:4,78-84): spaces.Box and spaces.Discrete."""
from . import spaces


class Env:
    pass
