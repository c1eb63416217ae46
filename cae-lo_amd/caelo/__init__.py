"""caelo -- MI355X-native CAE-LO feature-and-matching engine (host side).

    caelo.api      the reference's function-level call surface (NumPy or torch-GPU arrays)
    caelo.engine   device engine: fused extract / match_pose hot path, one process per GPU
    caelo.dist     frame sharding + RCCL descriptor all-gather (torch.distributed)
    caelo.h5lite   minimal HDF5 reader for the Keras .h5 weights
    caelo.synth    seeded synthetic KITTI-shaped scans
    caelo._ffi     ctypes binding of libcaelo.so (include/caelo.h)

Sub-modules are imported lazily so that the light ones (h5lite, synth) work without torch.
"""
import importlib
import os
import sys

# The frame pipeline overlaps four HIP streams when the runtime has hardware queues for them: ROCclr deals streams onto
# GPU_MAX_HW_QUEUES queues (default 4, one of which the default stream holds) and reads the variable when HIP initialises, which
# happens at the first device call -- after this import in every entry point of the repository.  Respecting a value the user set.
_torch = sys.modules.get("torch")
if "GPU_MAX_HW_QUEUES" not in os.environ and not (_torch is not None and _torch.cuda.is_initialized()):
    os.environ["GPU_MAX_HW_QUEUES"] = "8"   # (left alone when HIP is already up: the pipeline then stays on three streams)

__all__ = ["api", "engine", "dist", "h5lite", "synth", "_ffi"]


def __getattr__(name):
    if name in __all__:
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
