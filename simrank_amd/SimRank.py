"""Import-path twin of the reference's ``SimRank/SimRank.py`` module: the same public names,
defined in ``estimators.py``."""
from .estimators import (  # noqa: F401
    AprioriSimRank, BipartiteAprioriSimRank, BipartiteSimRank, BipartiteSimRankPP,
    BipartitleAprioriSimRank, BipartitleSimRank, BipartitleSimRankPP, SimRank, SimRankPP)
from .progress import BAR_LENGTH, update_progress  # noqa: F401
