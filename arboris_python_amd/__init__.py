"""arboris-python_amd: MI355X-native batched rigid-body step behind arboris' plugin API.

Two entry levels:

* the arboris-compatible object API (``core.World`` / ``Body`` / joints /
  constraints / controllers / ``robots``), whose four step methods run on the
  GPU for a batch of one world;
* ``batch.BatchedWorlds``: thousands of independent instances of one flattened
  world advanced by the hand-written HIP kernels of ``csrc/`` through the C ABI
  declared in ``include/arbstep.h``.
"""
__all__ = ["core", "joints", "homogeneousmatrix", "twistvector", "adjointmatrix",
           "rigidmotion", "massmatrix", "shapes", "collisions", "constraints",
           "controllers", "robots", "flatten", "batch"]
__version__ = "0.1.0"
