"""cuda-pathtracer_amd — MI355X-native replacement for the per-pixel path-tracing megakernel of
DavidPeicho/cuda-pathtracer (raytrace() / setupFunctionTables(), cuda_opengl/include/shaders/raytrace.h).

The product is libptamd.so (hand-written HIP for gfx950 behind the C-ABI of include/ptamd.h);
this package is the thin host mirror of the reference interface.  The directory name
carries a hyphen; import it as `cuda_pathtracer_amd` (alias module at the repo root).
"""
from . import native
from .native import (KERNEL_AUTO, KERNEL_BRUTE_FORCE, KERNEL_BVH, KERNEL_BVH_PERSISTENT, KERNEL_BVH_BLOCKWISE, KERNEL_BVH_SPLIT, KERNEL_BVH_RESTART, KERNEL_BVH_RESTART_FMA,
                     PtamdError)
from .scene import (HostScene, cubemap_for_scene, cubemap_from_color, cubemap_from_cross,
                    FACE_DTYPE, MATERIAL_DTYPE, LIGHT_DTYPE, TEXTURE_DTYPE, CAMERA_DTYPE)
from .render import (Context, FrameRenderer, host_bvh_trace, host_bvh4_trace, host_bvh4q_trace, host_bvh8_trace, interleaved_rows, wang_hash, REFERENCE_BOUNCES,
                     POST_NONE, POST_GRAYSCALE, POST_SEPIA, POST_INVERT)
from .tiles import row_bands, band_of_rank, BandGather, interleaved_bands, rank_times_ms, time_gather_ms
from .synthetic import tessellate
from .image import save_ppm, load_ppm, save_png
from .images import pil_image_loader, ldr_to_float, load_image, load_image8, native_image_loader, resize_float

__all__ = [n for n in dir() if not n.startswith("_")]
