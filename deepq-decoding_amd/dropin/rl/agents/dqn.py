from _bootstrap import package as _package

DQNAgent = _package("agent").DQNAgent
