"""raytracer_amd: MI355X-native rendering core behind the dw-evans/raytracer host API.

Layout (only what the hot path needs):

* ``raytracer_amd.csrc``     hand-written gfx950 HIP kernels + the C-ABI (``librt_hip.so``)
* ``raytracer_amd.ffi``      ctypes binding of ``include/rt_hip.h`` (fails loudly if the .so is absent)
* ``raytracer_amd.scripts``  numpy-only mirror of the reference's ``scripts`` package
  (Scene/Mesh/Camera/Material, Csys, chunker, Application/RayTracerDynamic)
* ``raytracer_amd.vecmath``  the few pyrr types the scene modules use (Vector3/Vector4)
"""

__version__ = "0.1.0"
