"""facedeform_amd -- MI355X (gfx950) RBF deformation engine behind the `facedeform` SOP.

The product is facedeform_amd/lib/libfacedeform_hip.so (C ABI:
include/facedeform_hip.h; sources: facedeform_amd/csrc).  The Python modules
here are plumbing for tests and bench.py:
  capi   ctypes binding of the C ABI (no fallback: raises if the library or a
         gfx950 device is missing)
  sop    handle on the C++ cook mirror of SOP_FaceDeform::cookMySop
  synth  deterministic synthetic inputs
"""
__all__ = ["capi", "sop", "synth"]
__version__ = "0.1.0"
