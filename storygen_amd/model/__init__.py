"""Host-side mirror of the reference's `model` package for the hot path: the drop-in `UNet2DConditionModel`
(/root/reference/model/unet_2d_condition.py) and `StableDiffusionPipeline` (/root/reference/model/pipeline.py), both
running on the HIP engine.  INTEGRATION.md shows how the reference's scripts bind to them."""
from .unet_2d_condition import UNet2DConditionModel, UNet2DConditionOutput  # noqa: F401
from .pipeline import StableDiffusionPipeline, StableDiffusionPipelineOutput  # noqa: F401
from .attention_processor import HipCrossAttnProcessor  # noqa: F401
