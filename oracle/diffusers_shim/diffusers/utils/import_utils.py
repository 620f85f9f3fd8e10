from . import is_xformers_available, is_accelerate_available  # noqa: F401
