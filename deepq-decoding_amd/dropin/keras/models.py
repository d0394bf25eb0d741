"""`Sequential` that records layer descriptors and resolves to the GPU-backed model (ConvQModel) on first use; `load_model` exists
for the reference's static-decoder branch: a Dense-stack weight file becomes a FeedForwardReferee."""
from _bootstrap import package as _package

from .layers import Activation, Conv2D, Dense, Dropout, Flatten


class Sequential:
    def __init__(self, layers=None):
        self.layers = []
        self._model = None
        for l in layers or []:
            self.add(l)

    def add(self, layer):
        if self._model is not None:
            raise RuntimeError("the model was already built")
        self.layers.append(layer)

    # -- architecture recognition: Function_Library.py:338-377 ---------------------------------------------------------------
    def _describe(self):
        L = list(self.layers)
        if not L or not isinstance(L[0], Conv2D) or L[0].input_shape_arg is None:
            raise NotImplementedError("the first layer must be Conv2D(..., input_shape=(C, H, W), data_format='channels_first')")
        input_shape = tuple(L[0].input_shape_arg)
        i, cc, ff = 0, [], []

        def relu_after(i):
            if i < len(L) and isinstance(L[i], Activation) and L[i].activation == "relu":
                return i + 1
            raise NotImplementedError("every Conv2D / hidden Dense is followed by Activation('relu') in the reference's network")
        while i < len(L) and isinstance(L[i], Conv2D):
            cc.append([L[i].filters, L[i].kernel_size, L[i].strides])
            i = relu_after(i + 1)
        if i >= len(L) or not isinstance(L[i], Flatten):
            raise NotImplementedError("Flatten expected after the convolutions")
        i += 1
        while i + 1 < len(L) and isinstance(L[i], Dense) and not (isinstance(L[i + 1], Activation) and L[i + 1].activation == "linear"):
            units = L[i].units
            i = relu_after(i + 1)
            if i < len(L) and isinstance(L[i], Dropout):
                rate = L[i].rate
                i += 1
            else:
                rate = 0.0
            ff.append([units, rate])
        if i >= len(L) or not isinstance(L[i], Dense):
            raise NotImplementedError("the network ends in Dense(num_actions) + Activation('linear')")
        num_actions = L[i].units
        rest = L[i + 1:]
        if any(not (isinstance(r, Activation) and r.activation == "linear") for r in rest) or len(rest) > 1:
            raise NotImplementedError("unsupported layers after the output Dense")
        return cc, ff, input_shape, num_actions

    def _built(self):
        if self._model is None:
            cc, ff, input_shape, num_actions = self._describe()
            self._model = _package("agent").ConvQModel(cc, ff, input_shape, num_actions)
        return self._model

    def __getattr__(self, name):                 # everything else (get_weights, load_weights, summary, ...) is the built model's
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._built(), name)


def load_model(path, *args, **kwargs):
    """The reference loads its referee decoder with this (Single_Point_Training_Script.py:54-57).  A weight file holding a plain
    Dense stack resolves to FeedForwardReferee (same .predict protocol; the environment tabulates it once, env.py
    VectorEnv.set_referee_predict); anything else cannot be rebuilt without Keras."""
    try:
        return _package("referee").FeedForwardReferee.from_file(path)
    except Exception as e:
        raise NotImplementedError("Keras models cannot be rebuilt here (only a Dense-stack weight file resolves to a referee: %s); "
                                  "static_decoder=None selects the built-in look-up referee" % (e,))
