"""Stand-in for omegaconf (absent from this image): OmegaConf.create(yaml_text | dict) -> attribute-style config with `${key}`
interpolation of top-level scalars resolved at creation (all the reference's cfg tree uses: `outputs/${exp_name}`).  Test-side only."""
import re

import yaml

from smplsim_amd.config import AttrDict

DictConfig = AttrDict
ListConfig = list


def _resolve(node, root):
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        return re.sub(r"\$\{([\w.]+)\}", lambda m: str(_lookup(root, m.group(1))), node)
    return node


def _lookup(root, dotted):
    cur = root
    for k in dotted.split("."):
        cur = cur[k]
    return cur


class OmegaConf:
    @staticmethod
    def create(obj=None):
        d = yaml.safe_load(obj) if isinstance(obj, str) else dict(obj or {})
        return AttrDict(_resolve(d, d))

    @staticmethod
    def to_container(cfg, resolve=True, throw_on_missing=False):
        return {k: OmegaConf.to_container(v) if isinstance(v, dict) else v for k, v in cfg.items()}

    @staticmethod
    def to_yaml(cfg):
        return yaml.safe_dump(OmegaConf.to_container(cfg))
