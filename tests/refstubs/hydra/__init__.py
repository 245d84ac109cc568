"""Stand-in for hydra (absent from this image) — just enough of `@hydra.main(config_path=..., config_name=...)` to run the
reference's smpl_sim/run.py UNCHANGED: the primary config's `defaults` list (`_self_` + one file per config group), command-line
overrides `group=name` / `dotted.key=yaml_value`, `--config-path DIR` (`-cp`), `hydra.run.dir` as the output directory exposed through
hydra.core.hydra_config.HydraConfig.get().runtime.output_dir.  Test-side only (tests/test_reference_agent.py)."""
import functools
import os
import sys
import types

import yaml

from . import core  # noqa: F401
from .core import hydra_config


def _load(path):
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _set(d, dotted, val):
    ks = dotted.split(".")
    for k in ks[:-1]:
        d = d.setdefault(k, {})
    d[ks[-1]] = val


def compose(config_path, config_name, overrides):
    from omegaconf import OmegaConf
    primary = _load(os.path.join(config_path, config_name + ".yaml"))
    defaults = primary.pop("defaults", [])
    groups = {}
    for e in defaults:
        if isinstance(e, dict):
            groups.update(e)
    plain = []
    for o in overrides:
        k, v = o.split("=", 1)
        if k in groups and os.path.isdir(os.path.join(config_path, k)):
            groups[k] = v
        else:
            plain.append((k, yaml.safe_load(v)))
    cfg = {}
    for e in defaults:                                        # in the order of the defaults list, like hydra
        if e == "_self_":
            cfg.update(primary)
        else:
            for g in e:
                cfg[g] = _load(os.path.join(config_path, g, groups[g] + ".yaml"))
    if "_self_" not in defaults:
        cfg.update(primary)
    for k, v in plain:
        _set(cfg, k, v)
    hy = cfg.pop("hydra", {})
    full = OmegaConf.create(dict(cfg, hydra=hy))
    out = full.pop("hydra")
    return full, out


def main(version_base=None, config_path=None, config_name=None):
    def deco(fn):
        @functools.wraps(fn)
        def wrapper():
            argv, cp, overrides = sys.argv[1:], config_path, []
            i = 0
            while i < len(argv):
                if argv[i] in ("--config-path", "-cp"):
                    cp = argv[i + 1]; i += 2
                else:
                    overrides.append(argv[i]); i += 1
            cfg, hy = compose(cp, config_name, overrides)
            out_dir = os.path.abspath(hy.get("run", {}).get("dir", "outputs"))
            os.makedirs(out_dir, exist_ok=True)
            hydra_config.HydraConfig._cfg = types.SimpleNamespace(runtime=types.SimpleNamespace(output_dir=out_dir))
            return fn(cfg)
        return wrapper
    return deco
