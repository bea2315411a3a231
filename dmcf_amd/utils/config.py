"""YAML configuration -- mirror of what the hot path needs from the reference's ``o3d/utils/config.py``.

``Config.load_from_file`` (config.py:232-263) reads the three top-level sections ``dataset`` / ``model`` /
``pipeline`` of ``configs/*.yml``; ``Config.merge_cfg_file`` (config.py:102-138) applies command line
overrides of the form ``--section.key value`` (run_pipeline.py:46-52), coercing to the type of the value
being replaced (config.py:188-216).  Model kwargs go straight into the model constructor
(run_pipeline.py:114), so YAML keys == constructor parameter names.
"""
import copy

import yaml


class ConfigDict(dict):
    """dict with attribute access; missing keys raise (the reference uses addict with __missing__ -> KeyError)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    return obj


def _coerce(old, new):
    """config.py:188-216: strings from the command line take the type of the value they replace."""
    if not isinstance(new, str):
        return new
    if isinstance(old, bool):
        return new.lower() in ("1", "true", "yes")
    if isinstance(old, int):
        return int(new)
    if isinstance(old, float):
        return float(new)
    if isinstance(old, (list, dict)) or old is None:
        try:
            return yaml.safe_load(new)
        except yaml.YAMLError:
            return new
    return new


class Config:
    def __init__(self, cfg_dict=None):
        if cfg_dict is None:
            cfg_dict = {}
        if not isinstance(cfg_dict, dict):
            raise TypeError(f"cfg_dict should be a dict, but got {type(cfg_dict)}")
        object.__setattr__(self, "_cfg_dict", _wrap(cfg_dict))

    @property
    def cfg_dict(self):
        return self._cfg_dict

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, name, default=None):
        return self._cfg_dict.get(name, default)

    def dump(self, **kwargs):
        return yaml.dump(_to_plain(self._cfg_dict), **kwargs)

    @staticmethod
    def load_from_file(filename):
        if not filename.endswith((".yml", ".yaml")):
            raise IOError("only yaml config files are supported on this path")
        with open(filename) as f:
            return Config(yaml.safe_load(f))

    @staticmethod
    def merge_cfg_file(cfg, args=None, extra_dict=None):
        """-> (cfg_dict_dataset, cfg_dict_pipeline, cfg_dict_model) like config.py:102-138."""
        args = args or {}
        if not isinstance(args, dict):
            args = vars(args)
        for key in ("device", "split", "main_log_dir", "output_dir"):
            if args.get(key) is not None:
                cfg.pipeline[key] = args[key]
                if key == "device":
                    cfg.model[key] = args[key]
        if args.get("dataset_path") is not None:
            cfg.dataset["dataset_path"] = args["dataset_path"]
        if args.get("ckpt_path") is not None:
            cfg.model["ckpt_path"] = args["ckpt_path"]
        for dotted, value in (extra_dict or {}).items():
            node = cfg._cfg_dict
            parts = dotted.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, ConfigDict())
            node[parts[-1]] = _coerce(node.get(parts[-1]), value)
        return (copy.deepcopy(cfg.dataset), copy.deepcopy(cfg.pipeline), copy.deepcopy(cfg.model))


def _to_plain(node):
    if isinstance(node, dict):
        return {k: _to_plain(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_to_plain(v) for v in node]
    return node
