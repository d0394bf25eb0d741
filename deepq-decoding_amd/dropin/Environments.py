"""Drop-in for the reference's Environments.py: `from Environments import *` gives the GPU-backed class."""
from _bootstrap import package as _package

from Function_Library import *  # noqa: F401,F403  (the reference module re-exports the helpers the same way)

Surface_Code_Environment_Multi_Decoding_Cycles = _package("env").Surface_Code_Environment_Multi_Decoding_Cycles
VectorEnv = _package("env").VectorEnv
