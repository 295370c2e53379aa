"""ctypes binding of libptamd.so — the C-ABI declared in include/ptamd.h.

The library is the product: there is no Python/CPU fallback.  Importing this module
fails loudly when the shared object has not been built (``python -c "import
__graft_entry__ as g; g.build()"`` or ``make lib``).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptamd.so")


class PtamdError(RuntimeError):
    """Raised when a C-ABI entry point returns a non-zero ptamd_status."""

    def __init__(self, status: int, message: str):
        super().__init__(f"ptamd status {status}: {message}")
        self.status = status


PTAMD_OK, PTAMD_ERR_ARG, PTAMD_ERR_HIP, PTAMD_ERR_IO, PTAMD_ERR_LIMIT = 0, 1, 2, 3, 4
KERNEL_AUTO, KERNEL_BRUTE_FORCE, KERNEL_BVH, KERNEL_BVH_PERSISTENT, KERNEL_BVH_BLOCKWISE, KERNEL_BVH_SPLIT, KERNEL_BVH_RESTART = 0, 1, 2, 3, 4, 5, 6
KERNEL_BVH_RESTART_FMA = 7   # opt-in, NOT bit-exact: the restart kernel with floating-point contraction allowed (include/ptamd.h)


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Float2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Face(C.Structure):  # scene_data.h:46-53
    _fields_ = [("vertices", Float3 * 3), ("normals", Float3 * 3), ("texcoords", Float2 * 3),
                ("tangent", Float3), ("material_id", C.c_uint32)]


class Material(C.Structure):  # scene_data.h:95-100
    _fields_ = [("diffuse_spec_map", C.c_int32), ("normal_map", C.c_int32), ("ior", C.c_float), ("_pad", C.c_int32)]


class Light(C.Structure):  # scene_data.h:109-115
    _fields_ = [("color", Float3), ("vec", Float3), ("emission", C.c_float), ("radius", C.c_float)]


class Camera(C.Structure):  # scene_data.h:123-133
    _fields_ = [("position", Float3), ("dir", Float3), ("u", Float3), ("v", Float3),
                ("fov_x", C.c_float), ("speed", C.c_float), ("aperture", C.c_float), ("focus_dist", C.c_float)]


class TextureDesc(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("nb_chan", C.c_int32), ("_pad", C.c_uint32), ("offset", C.c_uint64)]


class SceneDesc(C.Structure):
    _fields_ = [("faces", C.POINTER(Face)), ("n_faces", C.c_uint32),
                ("mesh_sizes", C.POINTER(C.c_uint32)), ("n_meshes", C.c_uint32),
                ("materials", C.POINTER(Material)), ("n_materials", C.c_uint32),
                ("lights", C.POINTER(Light)), ("n_lights", C.c_uint32),
                ("textures", C.POINTER(TextureDesc)), ("n_textures", C.c_uint32),
                ("texels", C.POINTER(C.c_float)), ("n_texel_floats", C.c_uint64)]


class Launch(C.Structure):
    _fields_ = [("surface_rgba8", C.c_void_p), ("temporal_framebuffer", C.c_void_p), ("stream", C.c_void_p),
                ("camera", Camera), ("scene_id", C.c_uint32), ("cubemap_id", C.c_uint32),
                ("width", C.c_uint32), ("height", C.c_uint32), ("row_begin", C.c_uint32), ("row_end", C.c_uint32),
                ("frame_nb", C.c_uint32), ("bounces", C.c_uint32), ("moved", C.c_int32), ("post_id", C.c_uint32),
                ("kernel", C.c_uint32), ("band_local_buffers", C.c_uint32), ("frame_count", C.c_uint32),
                ("machine_share", C.c_uint32),
                ("interleave_ranks", C.c_uint32), ("interleave_rank", C.c_uint32), ("interleave_rows", C.c_uint32),
                ("reset_accumulation", C.c_uint32), ("no_pipelining", C.c_uint32)]


class TraceStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("mesh_hits", C.c_uint64), ("nmap_hits", C.c_uint64), ("samples", C.c_uint64),
                ("wave_node_iters", C.c_uint64), ("wave_tri_iters", C.c_uint64),
                ("fetch_events", C.c_uint64), ("fetch_rays", C.c_uint64),
                ("idle_unstarted", C.c_uint64), ("idle_finished", C.c_uint64), ("idle_parked", C.c_uint64)]


class SceneInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_faces", "n_lights", "n_nodes", "n_leaves", "max_leaf_size", "depth",
                                          "node_bytes", "tri_bytes", "lds_bytes_bvh", "lds_bytes_brute",
                                          "n_nodes4", "depth4")]


assert C.sizeof(Face) == 112 and C.sizeof(Material) == 16 and C.sizeof(Light) == 32 and C.sizeof(Camera) == 64

IMAGE_LOAD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                            C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_float)))
IMAGE_FREE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float))
LOAD_FIX_BACKSLASHES, LOAD_NO_IMAGES = 1, 2   # ptamd_host_scene_load flags

# name -> (restype, argtypes); every symbol include/ptamd.h declares
SIGNATURES = {
    "ptamd_get_last_error": (C.c_char_p, []),
    "ptamd_version": (C.c_char_p, []),
    "ptamd_build_id": (C.c_char_p, []),
    "ptamd_host_scene_load": (C.c_int, [C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "ptamd_host_scene_load_ex": (C.c_int, [C.c_char_p, C.c_uint32, IMAGE_LOAD_FN, IMAGE_FREE_FN, C.c_void_p,
                                            C.POINTER(C.c_void_p)]),
    "ptamd_image_loadf": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.POINTER(C.POINTER(C.c_float))]),
    "ptamd_image_load8": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.POINTER(C.POINTER(C.c_uint8))]),
    "ptamd_image_free": (None, [C.c_void_p]),
    "ptamd_image_save_png": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_int32]),
    "ptamd_image_resize_float": (C.c_int, [C.POINTER(C.c_float), C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_int32, C.c_int32,
                                           C.c_int32]),
    "ptamd_host_scene_unloaded_count": (C.c_uint32, [C.c_void_p]),
    "ptamd_host_scene_unloaded_name": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "ptamd_host_scene_free": (None, [C.c_void_p]),
    "ptamd_host_scene_desc": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "ptamd_host_scene_camera": (C.c_int, [C.c_void_p, C.POINTER(Camera)]),
    "ptamd_host_scene_cubemap": (C.c_char_p, [C.c_void_p]),
    "ptamd_cubemap_from_color": (C.c_int, [C.c_uint32, C.POINTER(C.c_float)]),
    "ptamd_cubemap_from_cross": (C.c_int, [C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "ptamd_create": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p)]),
    "ptamd_destroy": (None, [C.c_void_p]),
    "ptamd_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc), C.POINTER(C.c_uint32)]),
    "ptamd_upload_cubemap": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32)]),
    "ptamd_setup_function_tables": (C.c_int, [C.c_void_p]),
    "ptamd_raytrace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Camera), C.c_uint32,
                                 C.c_uint32, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32]),
    "ptamd_raytrace_ex": (C.c_int, [C.c_void_p, C.POINTER(Launch)]),
    "ptamd_reset_frame_counter": (C.c_int, [C.c_void_p]),
    "ptamd_wang_hash": (C.c_uint32, [C.c_uint32]),
    "ptamd_interleaved_rows": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "ptamd_release_captured": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ptamd_raytrace_stats": (C.c_int, [C.c_void_p, C.POINTER(Launch), C.POINTER(TraceStats)]),
    "ptamd_scene_info_get": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(SceneInfo)]),
    "ptamd_set_timeline": (C.c_int, [C.c_void_p, C.c_uint32]),
    "ptamd_read_timeline": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32)]),
    "ptamd_device_error_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "ptamd_phase_cycles": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "ptamd_gamma_table_selftest": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ptamd_trace_rays": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.c_uint32,
                                   C.POINTER(C.c_int32)]),
    "ptamd_host_bvh_trace": (C.c_int, [C.POINTER(Face), C.c_uint32, C.POINTER(C.c_float), C.c_uint32,
                                       C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "ptamd_host_bvh4_trace": (C.c_int, [C.POINTER(Face), C.c_uint32, C.POINTER(C.c_float), C.c_uint32,
                                       C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "ptamd_host_bvh4q_trace": (C.c_int, [C.POINTER(Face), C.c_uint32, C.POINTER(C.c_float), C.c_uint32,
                                       C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "ptamd_host_bvh8_trace": (C.c_int, [C.POINTER(Face), C.c_uint32, C.POINTER(C.c_float), C.c_uint32,
                                       C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "ptamd_trace_rays_queue": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                        C.POINTER(C.c_uint32)]),
    "ptamd_device_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "ptamd_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ptamd_device_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "ptamd_device_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ptamd_stream_synchronize": (C.c_int, [C.c_void_p, C.c_void_p]),
}

_lib = None


def load() -> C.CDLL:
    """Loads libptamd.so (once) and types every entry point.  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64, and libptamd.so names
    # the same sonames.  Loaded after torch, the library binds to the copies torch brought (one runtime, shared device state);
    # loaded BEFORE torch it pulls in /opt/rocm's copies, torch then brings its own, and whichever initialises second finds
    # "no ROCm-capable device" (seen on the GPU box: build() + smoke() in one interpreter).  So torch, when it is installed,
    # is imported first.  Hosts without torch (C++, or Python with PTAMD_NO_TORCH_PRELOAD=1) are unaffected.
    if "torch" not in sys.modules and os.environ.get("PTAMD_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP library has not been built. "
            "Run `python -c \"import __graft_entry__ as g; g.build()\"` (or `make lib`). "
            "There is no CPU fallback for the render path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != PTAMD_OK:
        raise PtamdError(status, load().ptamd_get_last_error().decode("utf-8", "replace"))
