"""``python -m cryovit_amd.training.sam_features [key=value ...]`` -- feature extraction with the SAM2 image encoder.

Mirror of ``/root/reference/src/cryovit/training/sam_features.py``: config name ``sam_features`` (``use_sam: True``), the
same runner as the DINO entry point, missing keys -> logged + exit 1, runtime exceptions logged and swallowed."""

from __future__ import annotations

import logging
import sys
import warnings

from cryovit_amd.config import compose
from cryovit_amd.training.dino_features import _run

warnings.simplefilter("ignore")
logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")


def main(argv: list[str] | None = None) -> None:
    _run(compose("sam_features", sys.argv[1:] if argv is None else argv))


if __name__ == "__main__":
    main()
