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

__all__ = ["api", "engine", "dist", "h5lite", "synth", "_ffi"]


def __getattr__(name):
    if name in __all__:
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
