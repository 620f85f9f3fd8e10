"""Clean-room stand-in for the parts of `diffusers==0.13.1` that StoryGen's model files import.

TEST INFRASTRUCTURE ONLY (oracle).  The reference pins diffusers 0.13.1
(/root/reference/environment.yaml:108) but the package is neither vendored nor installable here
(no network), so the leaf semantics the reference relies on are restated from the published
behaviour of that release, using nothing but torch primitives.  Placing this directory first on
sys.path lets /root/reference/model/*.py be imported *verbatim* (see oracle/ref_runner.py).

Nothing in the shipped product (storygen_amd/) may import this package.
"""
from .models.vae import AutoencoderKL  # noqa: F401
from .schedulers import DDIMScheduler, DDPMScheduler, PNDMScheduler  # noqa: F401

__version__ = "0.13.1+shim"
