"""`AttentionBlock` is imported by /root/reference/model/unet_2d_blocks.py:4 but only instantiated by the dead
`UNetMidBlock2D` (never built for the SD-1.5 config), so a constructor-compatible stub suffices."""
from torch import nn


class AttentionBlock(nn.Module):
    def __init__(self, channels, num_head_channels=None, norm_num_groups=32, rescale_output_factor=1.0, eps=1e-5):
        super().__init__()
        raise NotImplementedError("AttentionBlock is not on the StoryGen SD-1.5 path")
