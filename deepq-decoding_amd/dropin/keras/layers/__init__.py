"""Layer descriptors (see keras/__init__.py)."""


class Layer:
    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.input_shape_arg = kwargs.get("input_shape")


class Conv2D(Layer):
    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", data_format=None, input_shape=None, **kwargs):
        super().__init__(input_shape=input_shape, **kwargs)
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        s = strides if isinstance(strides, int) else strides[0]
        if not isinstance(kernel_size, int) and len(set(kernel_size)) != 1 or not isinstance(strides, int) and len(set(strides)) != 1:
            raise NotImplementedError("only square kernels / equal strides (the reference's c_layers = [filters, kernel, stride])")
        if padding != "valid" or data_format != "channels_first":
            raise NotImplementedError("the reference builds Conv2D(padding='valid', data_format='channels_first') only")
        self.filters, self.kernel_size, self.strides = int(filters), int(k), int(s)


class Dense(Layer):
    def __init__(self, units, activation=None, **kwargs):
        super().__init__(**kwargs)
        self.units, self.activation = int(units), activation


class Activation(Layer):
    def __init__(self, activation, **kwargs):
        super().__init__(**kwargs)
        self.activation = activation


class Dropout(Layer):
    def __init__(self, rate, **kwargs):
        super().__init__(**kwargs)
        self.rate = float(rate)


class Flatten(Layer):
    pass


class _Unsupported(Layer):
    def __init__(self, *args, **kwargs):
        super().__init__(**kwargs)


class MaxPooling2D(_Unsupported):
    pass


class ZeroPadding2D(_Unsupported):
    pass


class GlobalAveragePooling2D(_Unsupported):
    pass
