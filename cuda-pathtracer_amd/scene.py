"""Host-side scene container: what the reference's scene::Scene::upload parses and
flattens (cuda_opengl/src/scene/scene.cpp:86-358, material_loader.cpp:164-401), as numpy
views over the arrays the C-ABI exchanges (include/ptamd.h: ptamd_scene_desc).

Parsing is done by the native loader (host/scene_loader.cpp); this module only wraps it.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import native as N

FACE_DTYPE = np.dtype([("vertices", "<f4", (3, 3)), ("normals", "<f4", (3, 3)), ("texcoords", "<f4", (3, 2)),
                       ("tangent", "<f4", (3,)), ("material_id", "<u4")])
MATERIAL_DTYPE = np.dtype([("diffuse_spec_map", "<i4"), ("normal_map", "<i4"), ("ior", "<f4"), ("_pad", "<i4")])
LIGHT_DTYPE = np.dtype([("color", "<f4", (3,)), ("vec", "<f4", (3,)), ("emission", "<f4"), ("radius", "<f4")])
TEXTURE_DTYPE = np.dtype([("w", "<i4"), ("h", "<i4"), ("nb_chan", "<i4"), ("_pad", "<u4"), ("offset", "<u8")])
CAMERA_DTYPE = np.dtype([("position", "<f4", (3,)), ("dir", "<f4", (3,)), ("u", "<f4", (3,)), ("v", "<f4", (3,)),
                         ("fov_x", "<f4"), ("speed", "<f4"), ("aperture", "<f4"), ("focus_dist", "<f4")])
assert FACE_DTYPE.itemsize == 112 and MATERIAL_DTYPE.itemsize == 16 and LIGHT_DTYPE.itemsize == 32
assert TEXTURE_DTYPE.itemsize == 24 and CAMERA_DTYPE.itemsize == 64

DEFAULT_CUBEMAP_COLOR = 0x131B23  # gpu_processor.cpp:75


def _copy_array(ptr, count, dtype):
    if count == 0:
        return np.zeros(0, dtype=dtype)
    buf = C.string_at(C.cast(ptr, C.c_void_p).value, count * dtype.itemsize)
    return np.frombuffer(buf, dtype=dtype).copy()


class HostScene:
    """Flattened scene + initial camera + cubemap name (scene::Scene's host state)."""

    def __init__(self, faces, mesh_sizes, materials, lights, textures, texels, camera=None, cubemap="",
                 unloaded_textures=()):
        self.unloaded_textures = list(unloaded_textures)
        self.faces = np.ascontiguousarray(faces, dtype=FACE_DTYPE)
        self.mesh_sizes = np.ascontiguousarray(mesh_sizes, dtype=np.uint32)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        self.lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        self.textures = np.ascontiguousarray(textures, dtype=TEXTURE_DTYPE)
        self.texels = np.ascontiguousarray(texels, dtype=np.float32)
        self.camera = np.zeros((), dtype=CAMERA_DTYPE) if camera is None else np.array(camera, dtype=CAMERA_DTYPE)
        self.cubemap = cubemap

    @classmethod
    def load(cls, scene_path: str, normalise_backslashes: bool = False, image_loader=None,
             decode_images: bool = True) -> "HostScene":
        """Parses a .scene file and the OBJ/MTL it names (ptamd_host_scene_load[_ex]).

        Texture files are decoded by libptamd's built-in decoder (bit-identical to the reference's
        stb_image); `decode_images=False` skips them (every texture degrades to its 1x1 constant).
        image_loader: optional replacement `f(path) -> float32[h, w, c] | None` with stbi_loadf
        semantics (see images.pil_image_loader)."""
        lib = N.load()
        h = C.c_void_p()
        flags = (N.LOAD_FIX_BACKSLASHES if normalise_backslashes else 0) | (0 if decode_images else N.LOAD_NO_IMAGES)
        if image_loader is None:
            N.check(lib.ptamd_host_scene_load(scene_path.encode(), flags, C.byref(h)))
        else:
            keep = {}

            def _load(user, path, pw, ph, pc, pdata):
                try:
                    img = image_loader(path.decode())
                except Exception:
                    img = None
                if img is None:
                    return 1
                img = np.ascontiguousarray(img, dtype=np.float32)
                if img.ndim == 2:
                    img = img[:, :, None]
                buf = (C.c_float * img.size).from_buffer_copy(img.tobytes())
                keep[C.addressof(buf)] = buf
                pw[0], ph[0], pc[0] = img.shape[1], img.shape[0], img.shape[2]
                pdata[0] = C.cast(buf, C.POINTER(C.c_float))
                return 0

            def _free(user, data):
                keep.pop(C.cast(data, C.c_void_p).value, None)

            cb_load, cb_free = N.IMAGE_LOAD_FN(_load), N.IMAGE_FREE_FN(_free)
            N.check(lib.ptamd_host_scene_load_ex(scene_path.encode(), flags, cb_load, cb_free, None, C.byref(h)))
        try:
            d = N.SceneDesc()
            N.check(lib.ptamd_host_scene_desc(h, C.byref(d)))
            cam = N.Camera()
            N.check(lib.ptamd_host_scene_camera(h, C.byref(cam)))
            camera = np.frombuffer(bytes(cam), dtype=CAMERA_DTYPE)[0].copy()
            cubemap = lib.ptamd_host_scene_cubemap(h).decode()
            unloaded = [lib.ptamd_host_scene_unloaded_name(h, i).decode()
                        for i in range(lib.ptamd_host_scene_unloaded_count(h))]
            return cls(_copy_array(d.faces, d.n_faces, FACE_DTYPE),
                       _copy_array(d.mesh_sizes, d.n_meshes, np.dtype("<u4")),
                       _copy_array(d.materials, d.n_materials, MATERIAL_DTYPE),
                       _copy_array(d.lights, d.n_lights, LIGHT_DTYPE),
                       _copy_array(d.textures, d.n_textures, TEXTURE_DTYPE),
                       _copy_array(d.texels, d.n_texel_floats, np.dtype("<f4")),
                       camera, cubemap, unloaded)
        finally:
            lib.ptamd_host_scene_free(h)

    def desc(self) -> N.SceneDesc:
        """ptamd_scene_desc borrowing this object's arrays (keep `self` alive while it is used)."""
        d = N.SceneDesc()
        d.faces = self.faces.ctypes.data_as(C.POINTER(N.Face)); d.n_faces = len(self.faces)
        d.mesh_sizes = self.mesh_sizes.ctypes.data_as(C.POINTER(C.c_uint32)); d.n_meshes = len(self.mesh_sizes)
        d.materials = self.materials.ctypes.data_as(C.POINTER(N.Material)); d.n_materials = len(self.materials)
        d.lights = self.lights.ctypes.data_as(C.POINTER(N.Light)); d.n_lights = len(self.lights)
        d.textures = self.textures.ctypes.data_as(C.POINTER(N.TextureDesc)); d.n_textures = len(self.textures)
        d.texels = self.texels.ctypes.data_as(C.POINTER(C.c_float)); d.n_texel_floats = len(self.texels)
        return d

    def camera_struct(self) -> N.Camera:
        return N.Camera.from_buffer_copy(self.camera.tobytes())

    def scene_bytes(self) -> int:
        """Bytes of geometry/material/light tables as the reference lays them out."""
        return (self.faces.nbytes + self.materials.nbytes + self.lights.nbytes + self.texels.nbytes)


def cubemap_from_color(rgb: int = DEFAULT_CUBEMAP_COLOR) -> np.ndarray:
    """1x1x6 constant cubemap (gpu_processor.cpp:37-57); returns float32[6,1,1,4]."""
    out = np.zeros(24, dtype=np.float32)
    N.check(N.load().ptamd_cubemap_from_color(rgb, out.ctypes.data_as(C.POINTER(C.c_float))))
    return out.reshape(6, 1, 1, 4)


def cubemap_from_cross(cross: np.ndarray) -> np.ndarray:
    """4x3 cube-cross image (H, W, C>=3 float32) -> float32[6, size, size, 4]
    (gpu_processor.cpp:101-127, texture_utils.cpp:5-52)."""
    cross = np.ascontiguousarray(cross, dtype=np.float32)
    h, w, c = cross.shape
    size = w // 4
    out = np.zeros(6 * max(size, 1) * max(size, 1) * 4, dtype=np.float32)
    osz = C.c_uint32()
    N.check(N.load().ptamd_cubemap_from_cross(cross.ctypes.data_as(C.POINTER(C.c_float)), w, h, c,
                                              out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(osz)))
    return out.reshape(6, osz.value, osz.value, 4)


def cubemap_for_scene(scene: HostScene, honour_hex: bool = False, asset_folder: str = None,
                      image_loader=None) -> np.ndarray:
    """The cubemap the reference binds for this scene (gpu_processor.cpp:68-161).

    With `asset_folder` the cube-cross image `folder/name` is decoded (built-in decoder unless an
    `image_loader` is given) and cut into six faces when it is a valid cross (width/4 == height/3, power of two); any failure —
    as for indoor.scene's missing garden.jpg — ends in the 1x1 fallback of colour 0x131b23
    (:128-132).  uploadCubemaps passes `folder + "/" + name` to uploadCubemap (:177-181), so the
    `path.empty() || isHexa(path)` arm (:89-93) is never taken by the reference: a `0xRRGGBB`
    name also ends in the fallback.  `honour_hex=True` applies the constant syntax as its author
    evidently intended (not reference behaviour)."""
    name = scene.cubemap
    if honour_hex and name.startswith("0x") and len(name) > 2 and all(ch in "0123456789abcdefABCDEF" for ch in name[2:]):
        return cubemap_from_color(int(name, 16) & 0xFFFFFF)
    if asset_folder is not None and name:
        if image_loader is None:
            from .images import native_image_loader as image_loader
        try:
            img = image_loader(asset_folder + "/" + name)
            if img is not None:
                return cubemap_from_cross(img)
        except (N.PtamdError, OSError, ValueError):
            pass
    return cubemap_from_color(DEFAULT_CUBEMAP_COLOR)
