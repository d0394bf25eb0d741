class ImageDataGenerator:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("not used by the decoding path")
