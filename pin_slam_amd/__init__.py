"""pin_slam_amd -- MI355X (gfx950) native implementation of PIN-SLAM's neural-point SDF hot path.

Layout: ``csrc/`` hand-written HIP kernels + the C ABI (include/pin_abi.h), ``_lib`` the
ctypes binding, ``ops`` tensor-level wrappers (torch tensors only as device-memory owners),
``model/`` and ``utils/`` the drop-in mirrors of the reference's NeuralPoints / Decoder /
Mapper / Tracker call surface.
"""
__version__ = "0.1.0"
