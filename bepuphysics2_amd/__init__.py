"""bepuphysics2_amd — MI355X-native batched sequential-impulse solver + pose integrator (the bepuphysics2 hot path).

The product path is the HIP library ``csrc/libbepuhip.so`` behind the C ABI of ``include/bepuhip.h``; this package holds
the host-side mirror of the reference's interface for that path. Nothing here imports from ``oracle/``.
"""
from .scene import (BUNDLE_WIDTH, KINEMATIC_MASK, TYPE_IDS_BY_NAME, TYPE_TABLE, PoseIntegratorCallbacks, Scene, SceneBuilder,
                    SolveDescription, TypeBatchData, make_body)

__all__ = ["BUNDLE_WIDTH", "KINEMATIC_MASK", "TYPE_IDS_BY_NAME", "TYPE_TABLE", "PoseIntegratorCallbacks", "Scene", "SceneBuilder",
           "SolveDescription", "TypeBatchData", "make_body"]
