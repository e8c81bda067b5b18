"""markovmodels.jl_amd -- MI355X-native engine behind MarkovModels.jl's inference API.

Host-side mirror of the reference's hot-path interface (src/MarkovModels.jl:14-45:
FSM, nstates, rawunion, CompiledFSM, batch, compile, expand, alpha-recursion,
beta-recursion, pdfposteriors, totalsum, totalcumsum, totalweightsum) over the C ABI in include/markovmodels_amd.h.
The directory name contains a dot, so load it with
``__graft_entry__.load_package()`` (importlib) rather than a plain import.
"""
from ._lib import LIB_PATH, SYMBOLS, DimensionMismatch, MarkovModelsAMDError  # noqa: F401
from .fsm import FSM, GeneralStateMap, StateMap, nstates, rawunion, statemap  # noqa: F401
from .inference import (  # noqa: F401
    BatchedFSM,
    CompiledFSM,
    alpharecursion,
    batch,
    bestpath,
    betarecursion,
    compile,
    compile_many,
    compiled_cache_clear,
    compiled_cache_stats,
    expand,
    maxstateposteriors,
    pdfposteriors,
    totalcumsum,
    totalsum,
    totalweightsum,
    αrecursion,
    βrecursion,
)
from . import dist, linalg  # noqa: F401
from .linalg import SparseCSR, SparseVector, eldiv_, elmul_, mul_  # noqa: F401
from .lfmmi import lfmmi_loss  # noqa: F401
