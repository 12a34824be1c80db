"""dawn-pytorch_amd: MI355X (gfx950) native implementation of DAWN's video-flow-diffusion denoising path
(DDIM sampler + spatio-temporal UNet) and of the LFG flow decode that turns its output into frames, behind the
reference's own Python API.

Import name: ``dawn_pytorch_amd`` (the source directory is ``dawn-pytorch_amd/``).
Compute runs only in the hand-written HIP kernels of ``libdawn_hip.so`` (C ABI: include/dawn_hip.h);
there is no CPU / eager fallback."""
from .unet import Unet3D, DynamicNfUnet3D                                    # noqa: F401
from .diffusion import GaussianDiffusion, DynamicNfGaussianDiffusion          # noqa: F401
from .flow_decoder import FlowDecoder                                         # noqa: F401

__all__ = ["Unet3D", "DynamicNfUnet3D", "GaussianDiffusion", "DynamicNfGaussianDiffusion", "FlowDecoder"]
