"""Console progress text of the reference (SimRank/Helper.py:1-19 and the messages at
SimRank.py:128, :132-134) — same characters, so logs of a drop-in run diff clean."""
import sys

BAR_LENGTH = 30


def _bar(fraction: float) -> str:
    done = int(round(BAR_LENGTH * fraction))
    return "#" * done + "-" * (BAR_LENGTH - done)


def update_progress(progress):
    """Write one ``\\rPercent: [###---] x% `` line fragment.  Same argument checks as the
    reference helper: ints are accepted, other non-floats and negatives raise ValueError."""
    if isinstance(progress, int):
        progress = float(progress)
    if not isinstance(progress, float):
        raise ValueError("Progress must be float")
    if progress < 0:
        raise ValueError("Progress below 0")
    tail = ""
    if progress >= 1:
        progress, tail = 1, "Done...\r\n"
    sys.stdout.write(f"\rPercent: [{_bar(progress)}] {round(progress * 100, 1)}% {tail}")
    sys.stdout.flush()


def announce_converged(k: int):
    sys.stdout.write(f"\rPercent: [{'#' * BAR_LENGTH}] 100% Complete! \n\rConverged at iteration {k}")
    sys.stdout.flush()
