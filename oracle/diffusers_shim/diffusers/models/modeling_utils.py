"""ModelMixin: dtype/device helpers + from_pretrained/save_pretrained in the diffusers folder layout."""
import os

import torch


class ModelMixin(torch.nn.Module):
    _supports_gradient_checkpointing = True

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def enable_gradient_checkpointing(self):
        self.apply(lambda m: self._set_gradient_checkpointing(m, value=True))

    def disable_gradient_checkpointing(self):
        self.apply(lambda m: self._set_gradient_checkpointing(m, value=False))

    def _set_gradient_checkpointing(self, module, value=False):
        pass

    def set_use_memory_efficient_attention_xformers(self, valid, attention_op=None):
        pass

    def save_pretrained(self, save_directory, **_):
        self.save_config(save_directory)
        torch.save(self.state_dict(), os.path.join(save_directory, "diffusion_pytorch_model.bin"))

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kwargs):
        folder = os.path.join(path, subfolder or "")
        model = cls.from_config(folder)
        weights = os.path.join(folder, "diffusion_pytorch_model.bin")
        if os.path.exists(weights):
            model.load_state_dict(torch.load(weights, map_location="cpu"))
        model.eval()
        return model
