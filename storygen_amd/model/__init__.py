"""Host-side mirror of the reference's `model` package for the hot path: the drop-in `UNet2DConditionModel`
(/root/reference/model/unet_2d_condition.py) and `StableDiffusionPipeline` (/root/reference/model/pipeline.py), both
running on the HIP engine, plus the frozen networks either side of the loop (`AutoencoderKL`, `CLIPTextModel`: the third-party
classes inference.py:45-46 constructs) on the same kernels.  INTEGRATION.md shows how the reference's scripts bind to them."""
from .unet_2d_condition import UNet2DConditionModel, UNet2DConditionOutput  # noqa: F401
from .pipeline import StableDiffusionPipeline, StableDiffusionPipelineOutput  # noqa: F401
from .attention_processor import HipCrossAttnProcessor  # noqa: F401
from .encoders import AutoencoderKL, CLIPTextModel  # noqa: F401
