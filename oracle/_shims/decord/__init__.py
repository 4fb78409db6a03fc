"""Import-only stand-in for `decord` (video reader imported by the reference's
train.py).  Used only by oracle/gen_golden.py."""
class VideoReader:  # pragma: no cover
    def __init__(self, *a, **k):
        raise RuntimeError("decord stub: video decoding is out of scope")
def cpu(*a, **k):  # pragma: no cover
    return None


def gpu(*a, **k):  # import-only stub (the reference's train.py imports the name)
    raise RuntimeError("decord stub: video decoding is not available in the oracle environment")
