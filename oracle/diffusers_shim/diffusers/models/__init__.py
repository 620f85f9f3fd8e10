from .modeling_utils import ModelMixin  # noqa: F401
