"""tdeed_amd -- MI355X-native (gfx950) T-DEED hot path.

Directory name is ``t-deed_amd`` (build contract); import name is ``tdeed_amd``
(see the shim ``tdeed_amd.py`` at the repo root).

Importing the package does not touch the GPU.  Every compute entry point goes
through the C-ABI library ``csrc/libtdeed_hip.so`` (see ``include/tdeed_hip.h``);
there is no CPU or eager-PyTorch fallback: a missing library raises
``tdeed_amd._lib.HipLibraryMissing`` at the first op.
"""
import os as _os

# The forward keeps two batches in flight on two forked streams each and the input pipeline copies on two more; with the HIP
# runtime's default of 4 hardware queues the copy streams share a queue with compute streams and the host-fed rate drops from
# 2140 to 1730 clips/s at cfg2 (device-resident throughput is unchanged; 6, 12 and 16 queues measured worse than 8).  Read by
# the runtime when it initialises, so it only takes effect if the package is imported before the first HIP call; an explicit
# setting by the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
