"""medaka_amd -- MI355X-native consensus-inference engine for medaka's one hot path.

    from medaka_amd.models import GRUModel          # drop-in for medaka.architectures.GRUModel
    from medaka_amd.torch_ext import Batch           # mirror of medaka.torch_ext.Batch
    from medaka_amd.integration import install       # swap the engine into an installed medaka

Importing this package never touches the GPU and never falls back to a CPU implementation.
"""
__version__ = "0.1.0"
import os as _os

# HIP's hardware-queue pool (read when the runtime initialises, at the process's first HIP call): a model's two contexts own
# eleven streams, and on the default four queues they alias and serialise each other (csrc/api.hip, mdk_default_hw_queues --
# the library sets the same default when it is loaded; this line also covers a torch that touches the device before that).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
