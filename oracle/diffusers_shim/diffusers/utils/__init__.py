"""BaseOutput, logging and availability probes."""
import logging as _pylogging
from collections import OrderedDict
from dataclasses import fields


class BaseOutput(OrderedDict):
    """Dataclass-style output with attribute, key and integer/tuple access."""

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                OrderedDict.__setitem__(self, f.name, v)

    def __getitem__(self, k):
        if isinstance(k, str):
            return OrderedDict.__getitem__(self, k)
        return self.to_tuple()[k]

    def to_tuple(self):
        return tuple(OrderedDict.__getitem__(self, k) for k in self.keys())


class _Logging:
    @staticmethod
    def get_logger(name=None):
        return _pylogging.getLogger(name)


logging = _Logging()


def is_accelerate_available():
    try:
        import accelerate  # noqa: F401
        return True
    except Exception:
        return False


def is_xformers_available():
    return False
