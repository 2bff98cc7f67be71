"""``cryovit`` command line for the two script-level flows on the hot path (mirror of
``/root/reference/src/cryovit/cli/{cli,dino_cli,infer_cli}.py``): same command names, arguments, options and defaults.

    python -m cryovit_amd.cli features <tomograms> <result-folder> [--batch-size 64] [--visualize]
    python -m cryovit_amd.cli infer <tomograms> --model x.model [--result-folder DIR] [--threshold 0.5]

``train`` / ``evaluate`` (Lightning training loops) are outside the hot path and not provided.  Extra options, marked
"build extension", replace the network fetch of the encoder weights.
"""

from __future__ import annotations

import logging
from pathlib import Path
from typing import Annotated, Optional

import typer
from typer import Argument, Option

cli = typer.Typer(add_completion=False, no_args_is_help=True, pretty_exceptions_show_locals=False)


@cli.callback()
def callback():
    """CryoViT's command line interface (MI355X build): feature extraction and inference."""


def _encoder_overrides(encoder: Optional[str], checkpoint: Optional[str], synthetic_seed: Optional[int]) -> dict:
    out = {}
    if encoder:
        out["name"] = encoder
    if checkpoint:
        out["checkpoint"] = checkpoint
    if synthetic_seed is not None:
        out["synthetic_seed"] = synthetic_seed
    return out


@cli.command(name="features", no_args_is_help=True)
def features(
    tomograms: Annotated[str, Argument(help="Path to the folder or .txt file containing the tomograms to process.")],
    result_folder: Annotated[str, Argument(help="Path to the folder where the DINO features will be saved.")],
    batch_size: Annotated[int, Option(min=1, help="Batch size for DINO feature extraction.")] = 64,
    visualize: Annotated[bool, Option("--visualize", "-v", help="Save PCA visualization of DINO features? (not built: skipped)")] = False,
    encoder: Annotated[Optional[str], Option(help="build extension: encoder name (default dinov2_vitg14_reg)")] = None,
    checkpoint: Annotated[Optional[str], Option(help="build extension: local DINOv2 state_dict file")] = None,
    synthetic_seed: Annotated[Optional[int], Option(help="build extension: seeded random encoder weights")] = None,
):
    """Compute high-level features using DINOv2 for a set of tomograms."""
    from cryovit_amd.run.dino_features import run_dino
    from cryovit_amd.utils import load_files_from_path

    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")
    tomograms_path, result_path = Path(tomograms), Path(result_folder)
    assert tomograms_path.exists(), "Tomograms path does not exist."
    result_path.mkdir(parents=True, exist_ok=True)
    run_dino(load_files_from_path(tomograms_path), result_path, batch_size=batch_size, visualize=visualize,
             encoder=_encoder_overrides(encoder, checkpoint, synthetic_seed))


@cli.command(name="infer", no_args_is_help=True)
def infer(
    tomograms: Annotated[str, Argument(help="Path to the folder or .txt file containing the tomograms to process.")],
    model: Annotated[str, Option(help="Path to the .model file containing the pre-trained model.")],
    result_folder: Annotated[Optional[str], Option(help="Path to the folder where the inference results will be saved.")] = None,
    threshold: Annotated[float, Option(min=0.0, max=1.0, help="Threshold for binary segmentation.")] = 0.5,
    encoder: Annotated[Optional[str], Option(help="build extension: encode files without dino_features on the fly with this encoder")] = None,
    checkpoint: Annotated[Optional[str], Option(help="build extension: local DINOv2 state_dict file")] = None,
    synthetic_seed: Annotated[Optional[int], Option(help="build extension: seeded random encoder weights")] = None,
):
    """Segment tomograms using a pre-trained model."""
    from cryovit_amd.run.infer_model import run_inference
    from cryovit_amd.utils import load_files_from_path

    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")
    tomograms_path, model_path = Path(tomograms), Path(model)
    result_path = Path(result_folder) if result_folder else Path.cwd() / "predictions"
    assert tomograms_path.exists(), "Tomograms path does not exist."
    assert model_path.exists() and model_path.suffix == ".model", "Model path does not exist or is not a .model file."
    result_path.mkdir(parents=True, exist_ok=True)
    enc = None
    ov = _encoder_overrides(encoder, checkpoint, synthetic_seed)
    if ov:
        from cryovit_amd.config import compose
        from cryovit_amd.models.encoder import load_encoder

        cfg = compose("dino_features", [])
        enc = load_encoder(ov.get("name", "dinov2_vitg14_reg"), model_dir=cfg.model_dir, checkpoint=ov.get("checkpoint"),
                           synthetic_seed=ov.get("synthetic_seed"))
    run_inference(load_files_from_path(tomograms_path), model_path, result_path, threshold=threshold, encoder=enc)


if __name__ == "__main__":
    cli()
