"""cryovit_amd -- MI355X-native (gfx950) implementation of CryoVIT's feature-extraction + segmentation hot path.

Host side: Python mirroring the reference's module layout (``training/``, ``run/``, ``datasets/``, ``models/``);
arithmetic: hand-written HIP kernels behind the C ABI declared in ``include/cryovit_hip.h``.  There is no CPU
fallback: importing the engine without ``libcryovit_hip.so`` raises.
"""

__version__ = "0.1.0"
