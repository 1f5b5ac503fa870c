"""tdeed_amd -- MI355X-native (gfx950) T-DEED hot path.

Directory name is ``t-deed_amd`` (build contract); import name is ``tdeed_amd``
(see the shim ``tdeed_amd.py`` at the repo root).

Importing the package does not touch the GPU.  Every compute entry point goes
through the C-ABI library ``csrc/libtdeed_hip.so`` (see ``include/tdeed_hip.h``);
there is no CPU or eager-PyTorch fallback: a missing library raises
``tdeed_amd._lib.HipLibraryMissing`` at the first op.
"""
__version__ = "0.1.0"
