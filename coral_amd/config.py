"""Minimal Hydra/OmegaConf-compatible loader for CoRal's config surface.

`hydra` and `omegaconf` are not installable here; the reference only uses a small subset
(R/src/scripts/finetune_asr_model.py:33-36, R/config/asr_finetuning.yaml:1-11,47-49):
  * a `defaults:` list with config groups (`- model: whisper-large`, `- datasets: [a, b]`,
    `- _self_`; `override hydra/...` and unknown groups without a directory are ignored),
  * `key=value` / `a.b=c` / `group=name` / `group=[a,b]` command-line overrides,
  * `${a.b}` interpolation and the `${now:%Y-%m-%d}` resolver,
  * attribute + item access on the result (`config.model.name`, `config["seed"]`).
"""

from __future__ import annotations

import datetime as _dt
import re
from pathlib import Path

import yaml

CONFIG_DIR = Path(__file__).resolve().parents[1] / "config"


class DictConfig(dict):
    """dict with attribute access, recursively (OmegaConf's DictConfig look-alike)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def get(self, k, default=None):
        return self[k] if k in self else default


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, DictConfig):
        return DictConfig({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


def _parse_value(text: str):
    """YAML scalar/list semantics for an override value (`8`, `1e-4`, `true`, `[a,b]`, `null`)."""
    try:
        v = yaml.safe_load(text)
    except yaml.YAMLError:
        return text
    if isinstance(v, str) and re.fullmatch(r"[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+", v):
        return float(v)  # PyYAML reads `1e-4` as a string; Hydra reads a float
    return v


def _fix_floats(node):
    if isinstance(node, dict):
        return {k: _fix_floats(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_fix_floats(v) for v in node]
    if isinstance(node, str) and re.fullmatch(r"[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+", node):
        return float(node)
    return node


def _load_yaml(path: Path):
    with path.open() as f:
        return _fix_floats(yaml.safe_load(f) or {})


def _merge(dst: dict, src: dict):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def _set_path(cfg: dict, dotted: str, value):
    keys = dotted.split(".")
    node = cfg
    for k in keys[:-1]:
        node = node.setdefault(k, {})
    node[keys[-1]] = value


def _resolve(cfg: dict):
    """Resolve ${a.b} and ${now:fmt} until a fixed point."""
    pat = re.compile(r"\$\{([^${}]+)\}")

    def lookup(path):
        if path.startswith("now:"):
            return _dt.datetime.now().strftime(path[4:])
        node = cfg
        for k in path.split("."):
            node = node[k]
        return node

    def walk(node):
        changed = False
        items = node.items() if isinstance(node, dict) else enumerate(node)
        for k, v in list(items):
            if isinstance(v, (dict, list)):
                changed |= walk(v)
            elif isinstance(v, str) and "${" in v:
                m = pat.fullmatch(v)
                new = lookup(m.group(1)) if m else pat.sub(lambda mm: str(lookup(mm.group(1))), v)
                if new != v:
                    node[k] = new
                    changed = True
        return changed

    for _ in range(10):
        if not walk(cfg):
            break
    return cfg


def load_config(config_name: str = "asr_finetuning", overrides: list[str] | None = None,
                config_dir: Path | str | None = None) -> DictConfig:
    """Compose `<config_dir>/<config_name>.yaml` with its defaults list and CLI-style overrides."""
    cdir = Path(config_dir) if config_dir is not None else CONFIG_DIR
    root = _load_yaml(cdir / f"{config_name}.yaml")
    defaults = root.pop("defaults", [])
    groups: dict[str, object] = {}
    for d in defaults:
        if isinstance(d, dict):
            for g, choice in d.items():
                if g.startswith("override "):
                    continue
                groups[g] = choice
    plain: list[tuple[str, object]] = []
    for ov in overrides or []:
        if "=" not in ov:
            raise ValueError(f"override {ov!r} is not of the form key=value")
        key, val = ov.split("=", 1)
        key = key.lstrip("+")
        if key in groups or (cdir / key).is_dir():
            groups[key] = _parse_value(val)
        else:
            plain.append((key, _parse_value(val)))
    cfg: dict = {}
    for g, choice in groups.items():
        gdir = cdir / g
        if not gdir.is_dir() or choice is None:
            continue
        if isinstance(choice, list):  # e.g. datasets=[a,b]: merged mapping of the group files
            merged: dict = {}
            for c in choice:
                _merge(merged, _load_yaml(gdir / f"{c}.yaml"))
            cfg[g] = merged
        else:
            cfg[g] = _load_yaml(gdir / f"{choice}.yaml")
    _merge(cfg, root)  # `_self_` comes last in the reference's defaults lists
    for key, val in plain:
        _set_path(cfg, key, val)
    return _wrap(_resolve(cfg))
