from . import _Unsupported


class BatchNormalization(_Unsupported):
    pass
