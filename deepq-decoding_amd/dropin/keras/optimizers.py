from _bootstrap import package as _package

Adam = _package("agent").Adam
