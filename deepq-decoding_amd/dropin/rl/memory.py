from _bootstrap import package as _package

SequentialMemory = _package("agent").SequentialMemory
