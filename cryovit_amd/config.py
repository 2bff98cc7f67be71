"""Configuration for the hot-path entry points: same keys and validation behaviour as the reference's Hydra configs
(``/root/reference/src/cryovit/config.py:106-156,205-231`` and ``configs/dino_features.yaml``, ``paths/default.yaml``,
``datamodule/dino.yaml``).  Hydra / OmegaConf are used when importable; otherwise ``compose`` below implements the
subset of their behaviour those files need (defaults lists, ``key=value`` overrides, ``${a.b}`` interpolation,
``???`` = missing, ``_target_`` / ``_partial_`` instantiation)."""

from __future__ import annotations

import importlib
import logging
import re
import sys
from functools import partial
from pathlib import Path
from typing import Any

import yaml

from cryovit_amd.types import Sample

CONFIG_DIR = Path(__file__).resolve().parent / "configs"
MISSING = "???"
samples: list[str] = [s.name for s in Sample]
tomogram_exts: list[str] = [".hdf", ".mrc"]
DINO_PATCH_SIZE = 14


class Cfg(dict):
    """dict with attribute access (the part of DictConfig the runners use)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _load(rel: str, choices: dict[str, str] | None = None, at: str = "") -> dict:
    """Load configs/<rel>.yaml honouring its ``defaults`` list: ``- grp: opt`` pulls <dir of this file>/<grp>/<opt>.yaml
    in under key ``grp``; ``_self_`` marks where the file's own keys merge (last if absent).  ``choices`` maps a group path
    (``model``, ``datamodule/dataset``) to the option a ``group=option`` override selected; an ``optional`` group whose
    file does not exist is skipped; a group still at ``???`` stays missing (reported by ``missing_keys``)."""
    choices = choices or {}
    raw = yaml.safe_load((CONFIG_DIR / f"{rel}.yaml").read_text()) or {}
    defaults = raw.pop("defaults", [])
    out: dict = {}
    self_done = False
    for d in defaults:
        if d == "_self_":
            _merge(out, raw)
            self_done = True
        elif isinstance(d, dict):
            for grp, opt in d.items():
                if grp.startswith("override "):
                    continue  # hydra's own logging overrides
                optional = grp.startswith("optional ")
                grp = grp.removeprefix("optional ")
                gpath = f"{at}{grp}"
                opt = choices.get(gpath, opt)
                if isinstance(opt, str) and opt.startswith("${") and opt.endswith("}"):
                    opt = choices.get(opt[2:-1], opt)
                if isinstance(opt, list):  # "- callbacks: [a, b]": several options merged under the group
                    sub = {}
                    for o in opt:
                        _merge(sub, _load(str(Path(rel).parent / grp / str(o)), choices, f"{gpath}/"))
                    _merge(out, {grp: sub})
                    continue
                if opt == MISSING:
                    _merge(out, {grp: MISSING})
                    continue
                target = str(Path(rel).parent / grp / str(opt))
                if not (CONFIG_DIR / f"{target}.yaml").exists():
                    if optional:
                        continue
                    raise FileNotFoundError(f"config group {gpath!r} has no option {opt!r}")
                _merge(out, {grp: _load(target, choices, f"{gpath}/")})
        # bare strings other than _self_ name structured-config schemas in the reference: nothing to load here
    if not self_done:
        _merge(out, raw)
    return out


_INTERP = re.compile(r"\$\{([^}]+)\}")


def _resolve(root: dict, node):
    if isinstance(node, dict):
        for k in list(node):
            node[k] = _resolve(root, node[k])
        return node
    if isinstance(node, list):
        return [_resolve(root, v) for v in node]
    if isinstance(node, str):
        for _ in range(8):
            m = _INTERP.search(node)
            if not m:
                break
            cur: Any = root
            ref = m.group(1)
            if ref.startswith("hydra:runtime.choices."):  # the option selected for a config group (model=cryovit -> "cryovit")
                ref = "__choices__." + ref.removeprefix("hydra:runtime.choices.")
            try:
                for part in ref.split("."):
                    cur = cur[part]
            except (KeyError, TypeError):
                break  # OmegaConf resolves lazily: a dangling reference only matters if somebody reads the key
            cur = _resolve(root, cur)
            node = cur if m.span() == (0, len(node)) else node[: m.start()] + str(cur) + node[m.end() :]
            if not isinstance(node, str):
                break
    return node


def _is_group_choice(key: str, val: str) -> bool:
    """``model=cryovit`` / ``datamodule/dataset=file`` select a config-group option when such a file exists."""
    return bool(re.fullmatch(r"[\w\-]+", val)) and (CONFIG_DIR / key / f"{val}.yaml").exists()


def compose(config_name: str, overrides: list[str] | None = None) -> Cfg:
    choices, sets = {}, []
    for ov in overrides or []:
        if "=" not in ov:
            raise ValueError(f"override {ov!r} is not key=value")
        key, val = ov.split("=", 1)
        key = key.lstrip("+")
        if _is_group_choice(key, val):
            choices[key] = val
        else:
            sets.append((key, val))
    cfg = _load(config_name, choices)
    cfg["__choices__"] = dict(choices)
    for key, val in sets:
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val) if val != "" else ""
    cfg = _resolve(cfg, cfg)
    del cfg["__choices__"]
    return _wrap(cfg)


def missing_keys(cfg, prefix="") -> list[str]:
    out = []
    if isinstance(cfg, dict):
        for k, v in cfg.items():
            out += missing_keys(v, f"{prefix}{k}.")
    elif isinstance(cfg, str) and MISSING in cfg:
        out.append(prefix.rstrip("."))
    return out


def validate_dino_config(cfg) -> None:
    """Missing mandatory parameters are logged and the process exits with status 1 (config.py:205-231)."""
    miss = missing_keys(cfg)
    if miss:
        msg = ["The following parameters were missing from dino_features.yaml"]
        msg += [f"{i}. {k}" for i, k in enumerate(miss, 1)]
        logging.error("\n".join(msg))
        sys.exit(1)


def validate_experiment_config(cfg) -> None:
    """config.py:234-286: missing parameters, then sample names outside the ``Sample`` enum -> logged, exit 1; a single
    sample string becomes a one-element list."""
    miss = missing_keys(cfg)
    if miss:
        logging.error("\n".join(["The following parameters were missing from config:"] + [f"{i}. {k}" for i, k in enumerate(miss, 1)]))
        sys.exit(1)
    dm = cfg.datamodule
    if isinstance(dm.sample, str):
        dm.sample = [dm.sample]
    if isinstance(dm.get("test_sample"), str):
        dm.test_sample = [dm.test_sample]
    invalid = [s for s in dm.sample if s not in samples]
    if isinstance(dm.get("test_sample"), list):
        invalid += [s for s in dm.test_sample if s not in samples]
    if invalid:
        logging.error("\n".join(["The following datamodule parameters are not valid samples:"] + [f"{i}. {s}" for i, s in enumerate(invalid, 1)]))
        sys.exit(1)


def instantiate(node, **kwargs):
    """``hydra.utils.instantiate`` for ``_target_`` / ``_partial_`` nodes (the reference's plug-in mechanism)."""
    node = dict(node)
    target = node.pop("_target_")
    is_partial = bool(node.pop("_partial_", False))
    mod, _, attr = target.rpartition(".")
    fn = getattr(importlib.import_module(mod), attr)
    def build(v):  # Hydra instantiates nested ``_target_`` nodes at any depth (e.g. model.metrics.dice_metric)
        if isinstance(v, dict):
            return instantiate(v) if "_target_" in v else {k: build(x) for k, x in v.items()}
        return [build(x) for x in v] if isinstance(v, list) else v

    args = {k: build(v) for k, v in node.items()}
    args.update(kwargs)
    return partial(fn, **args) if is_partial else fn(**args)
