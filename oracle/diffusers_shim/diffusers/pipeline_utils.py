"""Minimal DiffusionPipeline: module registry, `device`, progress bar, numpy_to_pil."""
import contextlib

import torch


class _Bar:
    def update(self, n=1):
        pass


class DiffusionPipeline:
    def register_modules(self, **kwargs):
        for name, module in kwargs.items():
            setattr(self, name, module)
        self._module_names = list(kwargs)

    @property
    def device(self):
        for name in getattr(self, "_module_names", []):
            m = getattr(self, name)
            if isinstance(m, torch.nn.Module):
                return next(m.parameters()).device
        return torch.device("cpu")

    @contextlib.contextmanager
    def progress_bar(self, iterable=None, total=None):
        yield _Bar()

    @staticmethod
    def numpy_to_pil(images):
        return images
