"""Hydra-compatible config surface without Hydra (not installed here): composes conf/config.yaml with the
``config`` group file, applies ``config=<group> config.<key>=<value>`` overrides and resolves the
interpolations the reference uses (``${config.*}``, ``${hydra:job.name}``, ``${hydra:runtime.output_dir}``,
``${now:<fmt>}``; conf/config.yaml:5-7,35-36, conf/config/unet.yaml:4)."""
import datetime
import os
import re

import yaml


class Config(dict):
    """dict with attribute access (OmegaConf-like for the keys train.py touches)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _parse_value(text):
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


def compose(conf_dir, overrides=(), job_name="train", now=None):
    """Return the resolved ``config`` node (what ``main(config)`` receives as ``config["config"]``)."""
    now = now or datetime.datetime.now()
    root = yaml.safe_load(open(os.path.join(conf_dir, "config.yaml"))) or {}
    group = None
    for d in root.get("defaults", []):
        if isinstance(d, dict) and "config" in d:
            group = d["config"]
    sets = []
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not key=value")
        k, v = ov.split("=", 1)
        if k == "config":
            group = v
        elif k.startswith("config."):
            sets.append((k[len("config."):], _parse_value(v)))
        elif k.startswith("hydra."):
            continue
        else:
            raise ValueError(f"unknown override '{k}' (expected config=<group> or config.<key>=<value>)")
    cfg = dict(root.get("config") or {})
    if group:
        path = os.path.join(conf_dir, "config", f"{group}.yaml")
        if not os.path.exists(path):
            raise FileNotFoundError(f"config group file {path} not found")
        cfg.update(yaml.safe_load(open(path)) or {})
    for k, v in sets:
        cfg[k] = v
    run_dir_t = (((root.get("hydra") or {}).get("run") or {}).get("dir")) or "./outputs"

    def resolve(val, depth=0):
        if not isinstance(val, str) or depth > 8:
            return val

        def sub(m):
            expr = m.group(1)
            if expr.startswith("now:"):
                return now.strftime(expr[4:])
            if expr == "hydra:job.name":
                return job_name
            if expr == "hydra:runtime.output_dir":
                return run_dir[0]
            if expr.startswith("config."):
                return str(resolve(cfg[expr[7:]], depth + 1))
            raise KeyError(f"unsupported interpolation ${{{expr}}}")
        return re.sub(r"\$\{([^}]+)\}", sub, val)

    run_dir = [None]
    run_dir[0] = os.path.abspath(resolve(run_dir_t.replace("${hydra:runtime.output_dir}", "")))
    out = Config({k: resolve(v) for k, v in cfg.items()})
    out.setdefault("hydra_path", run_dir[0])
    out.setdefault("job_name", job_name)
    return out


def parse_patch_size(config):
    """train.py:312-320: 'a, b, c' -> (a, b, c); 'a' -> int."""
    ps = config.patch_size
    if isinstance(ps, str):
        parts = ps.split(",")
        assert len(parts) <= 3, f"patch size can only be one str or three str but got {len(parts)}"
        config.patch_size = tuple(int(p) for p in parts) if len(parts) == 3 else int(ps)
    elif isinstance(ps, int):
        config.patch_size = ps
    return config.patch_size
