from . import _Unsupported


class LeakyReLU(_Unsupported):
    pass
