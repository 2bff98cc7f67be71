"""``python -m cryovit_amd.training.eval_model [key=value ...]`` -- evaluate a trained CryoVIT head on the test records.

Mirror of ``/root/reference/src/cryovit/training/eval_model.py:16-44``: config name ``eval_model``, missing mandatory keys
or unknown samples -> logged + exit 1 (``validate_experiment_config``), runtime exceptions logged with traceback and
swallowed (exit 0; a failed rank of a multi-GPU launch exits 1).  Typical call:

    python -m cryovit_amd.training.eval_model model=cryovit datamodule=single datamodule.sample=Q109 label_key=mito \\
        paths.model_dir=... paths.data_dir=... paths.exp_dir=...
"""

from __future__ import annotations

import logging
import sys
import traceback
import warnings

from cryovit_amd.config import compose, validate_experiment_config
from cryovit_amd.run.sharding import world_info

warnings.simplefilter("ignore")
logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")


def main(argv: list[str] | None = None) -> None:
    cfg = compose("eval_model", sys.argv[1:] if argv is None else argv)
    validate_experiment_config(cfg)
    from cryovit_amd.run import eval_model

    try:
        eval_model.run_trainer(cfg)
    except Exception as err:  # noqa: BLE001  (reference behaviour: log and continue)
        logging.error("%s: %s", type(err).__name__, err)
        logging.error(traceback.format_exc())
        if world_info()[2] > 1:
            raise SystemExit(1) from err


if __name__ == "__main__":
    main()
