"""poly_amd -- MI355X (gfx950) implementation of bebop/poly's search hot path.

Host-side mirror of the reference's Go packages over the C ABI of
``libpolyhip.so`` (include/polyhip.h):

    poly_amd.mash      <- search/mash      (Mash, New, Sketch, Similarity, Distance)
    poly_amd.align     <- search/align     (Scoring, NewScoring, SmithWaterman)
    poly_amd.primers   <- primers          (SantaLucia, MarmurDoty, MeltingTemp)
    poly_amd.seqhash   <- seqhash          (RotateSequence, Hash)

All compute happens in hand-written HIP kernels; PyTorch is used only to own
device memory and streams for the device-resident (``*_dev``) entry points.
There is no CPU fallback.
"""
from ._lib import GoPanic, PolyhipError, lib  # noqa: F401

__all__ = ["GoPanic", "PolyhipError", "lib"]
