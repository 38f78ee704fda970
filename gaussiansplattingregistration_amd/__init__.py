"""MI355X-native backend for the hot path of erikszasz/GaussianSplattingRegistration.

Two halves behind one C-ABI HIP library (``csrc/libgsr_hip.so``, declared in ``include/gsr_hip.h``):

* ``hem``  -- Hierarchical-EM Gaussian-mixture downsampler (reference ``src/cpp_ext``),
* ``icp``  -- the per-iteration ICP correspondence / transform step (reference: Open3D 0.16.0 behind
  ``src/utils/local_registration_util.py``).

Reference-shaped front ends: ``mixture_bind`` (same names as the pybind11 module), ``local_registration_util``
(``do_icp_registration`` and the enums), ``params``, ``gaussian_model``, ``point_cloud`` and headless
``controllers``.  There is no CPU fallback: without the HIP library or a GPU the entry points raise.
"""
__version__ = "0.1.0"
