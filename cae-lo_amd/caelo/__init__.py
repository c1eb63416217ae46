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

def configure_runtime(hw_queues=8):
    """Opt in to the pipeline's fourth stream (voxel maps beside the key-point chain).

    ROCclr deals HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable once, when HIP
    initialises at the first device call.  An entry point that owns its process (bench.py, run_sequence.py, the test
    session) calls this BEFORE touching the GPU; a library user who imports caelo into a bigger application decides for
    himself -- importing caelo never changes the environment.  Returns True when the variable is in effect for this
    process (set here or by the user to >= hw_queues), False when HIP was already up (three streams then; the pipeline
    reports what it runs with through caelo_pipeline_stats)."""
    cur = os.environ.get("GPU_MAX_HW_QUEUES")
    if cur is not None:
        try:
            return int(cur.strip() or 0) >= hw_queues
        except ValueError:   # the runtime reads it with atoi: a value it would take for 0 leaves the default of four queues
            return False
    torch_mod = sys.modules.get("torch")
    if torch_mod is not None and torch_mod.cuda.is_initialized():
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = str(hw_queues)
    return True


__all__ = ["api", "engine", "dist", "h5lite", "synth", "_ffi", "hostblas", "hostexact", "configure_runtime"]


def __getattr__(name):
    if name in __all__ and name != "configure_runtime":
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
