"""Stand-in for the parts of Keras 2.x the reference's driver scripts import (Single_Point_Training_Script.py:5-12): layer objects are
plain descriptors, `Sequential` collects them and is turned into the GPU-backed model description by the agent.  Only the
architecture family of Function_Library.build_convolutional_nn is accepted (Conv2D channels_first + relu ... Flatten, Dense + relu +
Dropout ..., Dense(num_actions) + linear); anything else raises NotImplementedError when the model is used."""
__version__ = "2.2.2-deepq-shim"
