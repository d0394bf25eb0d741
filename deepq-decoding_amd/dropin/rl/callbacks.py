from _bootstrap import package as _package

_a = _package("agent")
Callback, FileLogger = _a.Callback, _a.FileLogger
