"""MI355X-native engine for the Splice per-pair optimisation step.

Runtime tuning applied at import (before the HIP runtime initialises, overridable from the environment):

``GPU_MAX_HW_QUEUES=2``  A step is replayed as a two-branch hipGraph (main + side stream).  With the ROCm default of 4
    hardware queues per process the runtime spreads the branches and the caller's stream over more queues than the
    step has independent work for; measured on MI355X (ROCm 7.2) the same step takes 5.03 ms with 2-3 queues against
    5.18 ms with 4, and concurrent graphs of several engines in one process slow each other down badly with 8+.
"""
import os as _os

_os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
