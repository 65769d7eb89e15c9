"""MI355X-native path tracer for the RendererPT compute path.

Python here is binding and plumbing only: scene conversion, the renderer classes and all
kernels are native (C++ / HIP).  See include/lpt.h for the device-layer C-ABI.
"""
from .api import (BSDF_BASIC, BSDF_HEITZ, BUILDER_BINNED_SAH, BUILDER_LBVH, BUILDER_PLOC, BUILDER_SPLIT_BVH, GPU_BUILDERS,
                  PIPELINE_AUTO, PIPELINE_HYBRID, PIPELINE_MEGAKERNEL, PIPELINE_PERSISTENT, PIPELINE_WAVEFRONT, Device, LptError, Renderer, Scene,
                  frame_ubos, glibc_rand, write_png)

__all__ = ["Device", "Scene", "Renderer", "LptError", "frame_ubos", "glibc_rand", "write_png",
           "BSDF_HEITZ", "BSDF_BASIC", "PIPELINE_WAVEFRONT", "PIPELINE_MEGAKERNEL", "PIPELINE_PERSISTENT", "PIPELINE_AUTO", "PIPELINE_HYBRID",
           "BUILDER_SPLIT_BVH", "BUILDER_BINNED_SAH", "BUILDER_LBVH", "BUILDER_PLOC", "GPU_BUILDERS"]
