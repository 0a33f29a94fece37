"""gigalens_amd -- MI355X-native strong-lensing forward simulator behind the gigalens plugin API.

The hot path (ray-shoot -> light render -> chi^2 log-likelihood, forward and gradient) runs in
hand-written HIP kernels for gfx950, reached through the C ABI in ``include/gigalens_hip.h``.
There is no CPU or eager-PyTorch fallback: importing works anywhere, but any compute call raises
``gigalens_amd._native.NativeLibraryError`` unless ``lib/libgigalens_hip.so`` is built and a GPU is visible.
"""
__version__ = "0.1.0"
