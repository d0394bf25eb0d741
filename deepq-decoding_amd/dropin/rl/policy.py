from _bootstrap import package as _package

_a = _package("agent")
Policy, EpsGreedyQPolicy, GreedyQPolicy = _a.Policy, _a.EpsGreedyQPolicy, _a.GreedyQPolicy
LinearAnnealedPolicy, BoltzmannQPolicy = _a.LinearAnnealedPolicy, _a.BoltzmannQPolicy
