"""mms_answer_selection_amd -- MI355X (gfx950) implementation of the MMS
metric-learning inner loop of lxmeng/mms_answer_selection: the SimCross /
SimMatrix similarity layers and the PairRankLoss ranking hinge, forward and
backward, as hand-written HIP kernels behind a C ABI (include/mms.h) and a
mirror of the Caffe Layer/Blob interface.

Nothing in this package imports oracle/ and nothing computes on the CPU: if the
HIP library is not built, calls raise.
"""
from . import capi  # noqa: F401
from .capi import MMSError  # noqa: F401

__all__ = ["capi", "MMSError"]
