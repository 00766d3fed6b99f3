"""mmcv-1.3.9-style python config files: exec the file, collect its top-level names, load `_base_` files first
and deep-merge (`_delete_=True` replaces a dict), `merge_from_dict` for `--cfg-options a.b=c`."""
import copy
import os
import types

BASE_KEY = "_base_"
DELETE_KEY = "_delete_"


class ConfigDict(dict):
    """dict with attribute access (what mmcv gets from addict)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_wrap(v) for v in obj)
    return obj


def _merge_a_into_b(a, b):
    b = dict(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and not v.get(DELETE_KEY, False):
            if not isinstance(b[k], dict):
                raise TypeError(f"{k}={v} in child config cannot inherit from base because {k} is a dict in the child "
                                f"but {type(b[k])} in the base; set {DELETE_KEY}=True to replace it")
            b[k] = _merge_a_into_b(v, b[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            b[k] = v
    return b


def _file2dict(filename):
    filename = os.path.abspath(os.path.expanduser(filename))
    if not os.path.isfile(filename):
        raise FileNotFoundError(filename)
    if not filename.endswith(".py"):
        raise IOError("only python config files are supported")
    ns = {"__file__": filename}
    with open(filename) as f:
        code = compile(f.read(), filename, "exec")
    exec(code, ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith("__") and not isinstance(v, (types.ModuleType, types.FunctionType))}
    if BASE_KEY in cfg:
        base = cfg.pop(BASE_KEY)
        base = base if isinstance(base, list) else [base]
        merged = {}
        for b in base:
            bcfg = _file2dict(os.path.join(os.path.dirname(filename), b))
            dup = merged.keys() & bcfg.keys()
            if dup:
                raise KeyError(f"duplicate keys in base configs: {sorted(dup)}")
            merged.update(bcfg)
        cfg = _merge_a_into_b(cfg, merged)
    return cfg


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, "_cfg_dict", _wrap(cfg_dict or {}))
        object.__setattr__(self, "_filename", filename)

    @staticmethod
    def fromfile(filename):
        return Config(_file2dict(filename), filename=filename)

    @property
    def filename(self):
        return self._filename

    def merge_from_dict(self, options):
        nested = {}
        for full_key, v in options.items():
            d = nested
            keys = full_key.split(".")
            for sub in keys[:-1]:
                d = d.setdefault(sub, {})
            d[keys[-1]] = v
        object.__setattr__(self, "_cfg_dict", _wrap(_merge_a_into_b(nested, self._cfg_dict)))

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def keys(self):
        return self._cfg_dict.keys()

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))
