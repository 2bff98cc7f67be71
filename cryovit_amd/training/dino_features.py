"""``python -m cryovit_amd.training.dino_features [key=value ...]`` -- the feature-extraction entry point.

Mirror of ``/root/reference/src/cryovit/training/dino_features.py:16-41``: config name ``dino_features``, missing
mandatory keys -> logged + exit 1 (``validate_dino_config``), any runtime exception -> logged with traceback and the
process returns normally (l.33-37).  Uses Hydra when it is importable, else the built-in composer (same override
syntax).  Under ``python -m torch.distributed.run --nproc-per-node N`` the tomograms are sharded over N GPUs."""

from __future__ import annotations

import logging
import sys
import traceback
import warnings

from cryovit_amd.config import CONFIG_DIR, compose, validate_dino_config
from cryovit_amd.run.sharding import world_info

warnings.simplefilter("ignore")
logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")


def _run(cfg) -> None:
    from cryovit_amd.run import dino_features

    validate_dino_config(cfg)
    try:
        dino_features.run_trainer(cfg)
    except Exception as err:  # noqa: BLE001  (reference behaviour, l.33-37: log and continue; KeyboardInterrupt / SystemExit
        logging.error("%s: %s", type(err).__name__, err)  # are NOT swallowed here)
        logging.error(traceback.format_exc())
        if world_info()[2] > 1:  # a failed rank of a multi-GPU launch must be visible to the launcher
            raise SystemExit(1) from err


def main(argv: list[str] | None = None) -> None:
    argv = sys.argv[1:] if argv is None else argv
    try:
        import hydra  # type: ignore
        from omegaconf import OmegaConf  # type: ignore
    except ImportError:
        _run(compose("dino_features", argv))
        return

    from cryovit_amd.config import _wrap

    @hydra.main(config_path=str(CONFIG_DIR), config_name="dino_features", version_base="1.2")
    def _hydra_main(cfg):  # pragma: no cover - hydra is absent in the build image
        _run(_wrap(OmegaConf.to_container(cfg, resolve=False, throw_on_missing=False)))

    _hydra_main()


if __name__ == "__main__":
    main()
