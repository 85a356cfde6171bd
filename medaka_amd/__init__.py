"""medaka_amd -- MI355X-native consensus-inference engine for medaka's one hot path.

    from medaka_amd.models import GRUModel          # drop-in for medaka.architectures.GRUModel
    from medaka_amd.torch_ext import Batch           # mirror of medaka.torch_ext.Batch
    from medaka_amd.integration import install       # swap the engine into an installed medaka

Importing this package never touches the GPU and never falls back to a CPU implementation.
"""
__version__ = "0.1.0"
