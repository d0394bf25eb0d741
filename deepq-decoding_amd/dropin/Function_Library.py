"""Drop-in for the reference's Function_Library.py (names and signatures of Function_Library.py:13-377)."""
from _bootstrap import package as _package

_fl = _package("function_library")
generateSurfaceCodeLattice = _fl.generateSurfaceCodeLattice
multiplyPaulis = _fl.multiplyPaulis
generate_error = _fl.generate_error
generate_DP_error = _fl.generate_DP_error
generate_X_error = _fl.generate_X_error
generate_IIDXZ_error = _fl.generate_IIDXZ_error
generate_surface_code_syndrome_NoFT_efficient = _fl.generate_surface_code_syndrome_NoFT_efficient
generate_faulty_syndrome = _fl.generate_faulty_syndrome
obtain_new_error_configuration = _fl.obtain_new_error_configuration
index_to_move = _fl.index_to_move
generate_one_hot_labels_surface_code = _fl.generate_one_hot_labels_surface_code
build_convolutional_nn = _package("agent").build_convolutional_nn
