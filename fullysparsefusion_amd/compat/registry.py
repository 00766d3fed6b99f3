"""Registry / build_from_cfg with mmcv 1.3.9 semantics: `cfg['type']` names a registered class, the remaining
keys are its constructor kwargs (+ `default_args`)."""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return f"Registry(name={self._name}, items={sorted(self._module_dict)})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def _register(self, cls, name=None, force=False):
        if not inspect.isclass(cls) and not callable(cls):
            raise TypeError(f"module must be a class or callable, got {type(cls)}")
        names = [name] if isinstance(name, str) else (name or [cls.__name__])
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f"{n} is already registered in {self._name}")
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls

        return deco

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f"cfg must be a dict, got {type(cfg)}")
    if "type" not in cfg and not (default_args and "type" in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", got {cfg}')
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    elif inspect.isclass(obj_type) or callable(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f"type must be a str or class, got {type(obj_type)}")
    try:
        return obj_cls(**args)
    except Exception as e:  # same behaviour as mmcv: name the class in the error
        raise type(e)(f"{getattr(obj_cls, '__name__', obj_cls)}: {e}") from e
