"""ConfigMixin / register_to_config: capture ctor kwargs into `self.config` (attribute + dict access)."""
import functools
import inspect
import json
import os


class FrozenDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        raise AttributeError("config is frozen")


class ConfigMixin:
    config_name = "config.json"

    def register_to_config(self, **kwargs):
        kwargs.pop("kwargs", None)
        # 0.13.1 also mirrors every config entry as a plain attribute (the pipeline reads `unet.in_channels`,
        # /root/reference/model/pipeline.py:370).
        for key, value in kwargs.items():
            try:
                setattr(self, key, value)
            except AttributeError:
                pass
        existing = dict(getattr(self, "_internal_dict", {}))
        existing.update(kwargs)
        object.__setattr__(self, "_internal_dict", FrozenDict(existing))

    @property
    def config(self):
        return self._internal_dict

    @classmethod
    def load_config(cls, path, subfolder=None, **_):
        if os.path.isdir(path):
            path = os.path.join(path, subfolder or "", cls.config_name)
        with open(path) as f:
            return json.load(f)

    @classmethod
    def from_config(cls, config, subfolder=None, **kwargs):
        if isinstance(config, (str, os.PathLike)):
            config = cls.load_config(config, subfolder=subfolder)
        params = inspect.signature(cls.__init__).parameters
        init = {k: v for k, v in dict(config).items() if k in params and not k.startswith("_")}
        init.update({k: v for k, v in kwargs.items() if k in params})
        return cls(**init)

    def save_config(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        cfg = dict(self.config)
        cfg["_class_name"] = type(self).__name__
        with open(os.path.join(save_directory, self.config_name), "w") as f:
            json.dump(cfg, f, indent=2, default=list)


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        names = [n for n in sig.parameters if n != "self"]
        cfg = {n: p.default for n, p in sig.parameters.items() if n != "self" and p.default is not inspect._empty}
        cfg.update(dict(zip(names, args)))
        cfg.update(kwargs)
        self.register_to_config(**{k: v for k, v in cfg.items() if not k.startswith("_")})
        init(self, *args, **kwargs)

    return inner
