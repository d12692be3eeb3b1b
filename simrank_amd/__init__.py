"""MI355X-native SimRank / SimRank++ engine — drop-in for ysong1231/SimRank's ``fit`` classes.

    from simrank_amd import SimRank          # module with the reference's class names
    S = SimRank.SimRank().fit(edges)         # pandas edge list in, similarity DataFrame out

The iteration runs as hand-written HIP kernels for gfx950 behind the C ABI declared in
``include/simrank_hip.h``; there is no CPU fallback.
"""
from . import SimRank  # noqa: F401
from .estimators import (  # noqa: F401
    AprioriSimRank, BipartiteAprioriSimRank, BipartiteSimRank, BipartiteSimRankPP,
    BipartitleAprioriSimRank, BipartitleSimRank, BipartitleSimRankPP, SimRankPP)

__version__ = "0.1.0"
