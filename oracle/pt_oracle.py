"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so, built from oracle/pt_oracle.c).

TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (see the header of pt_oracle.c).  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product package
(cuda-pathtracer_amd/) must never import this module.

The oracle consumes the reference's own data layout: AoS Face[] per mesh, Material/LightProp
tables, a texture table with host pointers, and the 6-face float4 cubemap.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")

FACE_DTYPE = np.dtype([("vertices", "<f4", (3, 3)), ("normals", "<f4", (3, 3)), ("texcoords", "<f4", (3, 2)),
                       ("tangent", "<f4", (3,)), ("material_id", "<u4")])


class F3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class F2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("position", F3), ("dir", F3), ("u", F3), ("v", F3),
                ("fov_x", C.c_float), ("speed", C.c_float), ("aperture", C.c_float), ("focus_dist", C.c_float)]


class Light(C.Structure):
    _fields_ = [("color", F3), ("vec", F3), ("emission", C.c_float), ("radius", C.c_float)]


class Texture(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("nb_chan", C.c_int32), ("_pad", C.c_int32), ("data", C.c_void_p)]


class Mesh(C.Structure):
    _fields_ = [("size", C.c_uint32), ("_pad", C.c_uint32), ("data", C.c_void_p)]


class Scene(C.Structure):
    _fields_ = [("meshes", C.c_void_p), ("n_meshes", C.c_uint32), ("_p0", C.c_uint32),
                ("materials", C.c_void_p), ("n_materials", C.c_uint32), ("_p1", C.c_uint32),
                ("lights", C.c_void_p), ("n_lights", C.c_uint32), ("_p2", C.c_uint32),
                ("textures", C.c_void_p), ("n_textures", C.c_uint32), ("_p3", C.c_uint32),
                ("cubemap", C.c_void_p), ("cubemap_size", C.c_uint32), ("_p4", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32), ("t", C.c_float), ("u", C.c_float), ("v", C.c_float)]


_lib = None


def build() -> None:
    """Compiles the oracle (gcc).  Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "oracle"], cwd=os.path.dirname(_HERE))


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    _lib = _bind(C.CDLL(LIB_PATH))
    return _lib


def build_variant(out_path: str, flags=()) -> C.CDLL:
    """A STUDY instantiation of the same source (pt_oracle.c: OR_STUDY_*, -ffp-contract=fast): what another faithful build of
    the integrator may compute.  Never the oracle; used by scripts/tolerance_study.py and tests/test_tolerance_study.py."""
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    cmd = ["gcc", "-O2", "-std=c11", "-mfma", "-fPIC", "-shared", "-o", out_path, os.path.join(_HERE, "pt_oracle.c"), "-lm", "-lpthread"]
    if not any(f.startswith("-ffp-contract") for f in flags):
        cmd.insert(1, "-ffp-contract=off")
    subprocess.check_call(cmd[:1] + list(flags) + cmd[1:])
    return _bind(C.CDLL(out_path))


def _bind(lib: C.CDLL) -> C.CDLL:
    lib.or_wang_hash.restype = C.c_uint32
    lib.or_wang_hash.argtypes = [C.c_uint32]
    lib.or_xorwow_init.argtypes = [C.c_uint32, C.POINTER(C.c_uint32)]
    lib.or_xorwow_next.restype = C.c_uint32
    lib.or_xorwow_next.argtypes = [C.POINTER(C.c_uint32)]
    lib.or_xorwow_uniform.restype = C.c_float
    lib.or_xorwow_uniform.argtypes = [C.POINTER(C.c_uint32)]
    lib.or_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.or_powf.restype = C.c_float
    lib.or_powf.argtypes = [C.c_float, C.c_float]
    lib.or_generate_ray.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Camera), C.POINTER(F3), C.POINTER(F3)]
    lib.or_intersect_triangle.restype = C.c_int
    lib.or_intersect_triangle.argtypes = [C.c_void_p, C.POINTER(F3), C.POINTER(F3), C.POINTER(F3), C.POINTER(F2),
                                          C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.or_intersect_sphere.restype = C.c_int
    lib.or_intersect_sphere.argtypes = [C.POINTER(F3), C.POINTER(F3), C.POINTER(Light), C.POINTER(C.c_float)]
    lib.or_intersect.argtypes = [C.POINTER(Scene), C.POINTER(F3), C.POINTER(F3), C.POINTER(Hit)]
    lib.or_intersect_batch.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_uint32, C.c_void_p]
    lib.or_exposure.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.or_tex_cubemap.argtypes = [C.POINTER(Scene), C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
    lib.or_pack_rgba.restype = C.c_uint32
    lib.or_pack_rgba.argtypes = [C.POINTER(C.c_float)]
    lib.or_post_process.argtypes = [C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.or_render.restype = C.c_int
    lib.or_render.argtypes = [C.POINTER(Scene), C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                              C.c_uint32, C.c_int32, C.c_int32, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p,
                              C.c_int32]
    lib.or_last_stats.argtypes = [C.POINTER(C.c_uint64)]
    return lib


class OracleScene:
    """Builds the reference-layout pointer graph over numpy arrays (kept alive here)."""

    def __init__(self, faces, mesh_sizes, materials, lights, textures, texels, cubemap):
        """faces: FACE_DTYPE[n]; mesh_sizes: uint32[m]; materials: (diffuse_spec_map, normal_map,
        ior, pad) records of 16 B; lights: 32 B records; textures: records with w,h,nb_chan,offset;
        texels: float32 blob; cubemap: float32[6, size, size, 4]."""
        self.faces = np.ascontiguousarray(faces)
        assert self.faces.dtype.itemsize == 112
        self.materials = np.ascontiguousarray(materials)
        assert self.materials.dtype.itemsize == 16
        self.lights = np.ascontiguousarray(lights)
        assert self.lights.size == 0 or self.lights.dtype.itemsize == 32
        self.texels = np.ascontiguousarray(texels, dtype=np.float32)
        self.cubemap = np.ascontiguousarray(cubemap, dtype=np.float32)
        mesh_sizes = np.asarray(mesh_sizes, dtype=np.uint32)
        assert int(mesh_sizes.sum()) == len(self.faces)
        self._meshes = (Mesh * max(len(mesh_sizes), 1))()
        off = 0
        for i, n in enumerate(mesh_sizes):
            self._meshes[i].size = int(n)
            self._meshes[i].data = self.faces.ctypes.data + off * 112
            off += int(n)
        self._textures = (Texture * max(len(textures), 1))()
        for i, t in enumerate(textures):
            self._textures[i].w, self._textures[i].h = int(t["w"]), int(t["h"])
            self._textures[i].nb_chan = int(t["nb_chan"])
            self._textures[i].data = self.texels.ctypes.data + int(t["offset"]) * 4
        s = Scene()
        s.meshes = C.addressof(self._meshes); s.n_meshes = len(mesh_sizes)
        s.materials = self.materials.ctypes.data; s.n_materials = len(self.materials)
        s.lights = self.lights.ctypes.data if self.lights.size else None; s.n_lights = len(self.lights)
        s.textures = C.addressof(self._textures); s.n_textures = len(textures)
        s.cubemap = self.cubemap.ctypes.data; s.cubemap_size = self.cubemap.shape[1]
        self.c = s

    @classmethod
    def from_host_scene(cls, hs, cubemap) -> "OracleScene":
        """hs: any object with faces/mesh_sizes/materials/lights/textures/texels numpy arrays
        in the C-ABI layouts (the product loader's output is used as DATA here)."""
        return cls(hs.faces, hs.mesh_sizes, hs.materials, hs.lights, hs.textures, hs.texels, cubemap)


def camera_from_record(rec) -> Camera:
    """rec: a 64-byte camera record (numpy void / bytes) in scene::Camera layout."""
    raw = rec.tobytes() if hasattr(rec, "tobytes") else bytes(rec)
    return Camera.from_buffer_copy(raw)


def wang_hash(a: int) -> int:
    return load().or_wang_hash(a & 0xFFFFFFFF)


def render(scene: OracleScene, cam: Camera, width: int, height: int, spp: int = 1, bounces: int = 3,
           moved: bool = False, post_id: int = 0, rows=None, nthreads: int = 0, first_frame: int = 1,
           accum: np.ndarray | None = None, lib: C.CDLL | None = None):
    """N-spp render = N static launches with frame seeds first_frame.. (raytrace.cu:296-300).
    Returns (accum float32[H,W,3] in the reference's row-flipped order, rgba uint8[H,W,4]).
    lib: a study instantiation from build_variant() instead of the oracle."""
    lib = lib or load()
    if nthreads <= 0:
        nthreads = os.cpu_count() or 1
    y0, y1 = rows if rows is not None else (0, height)
    tfb = np.zeros((height, width, 3), dtype=np.float32) if accum is None else accum
    rgba = np.zeros((height, width, 4), dtype=np.uint8)
    for k in range(first_frame, first_frame + spp):
        rc = lib.or_render(C.byref(scene.c), C.byref(cam), width, height, y0, y1, wang_hash(k), k,
                           1 if moved else 0, post_id, bounces, tfb.ctypes.data, rgba.ctypes.data, nthreads)
        if rc != 0:
            raise RuntimeError(f"or_render failed with {rc}")
    return tfb, rgba


def last_stats():
    out = (C.c_uint64 * 3)()
    load().or_last_stats(out)
    return {"calls": out[0], "mesh_hits": out[1], "nmap_hits": out[2]}


def intersect(scene: OracleScene, rays: np.ndarray) -> np.ndarray:
    """Brute-force nearest hit (intersection.cuh:161-246) for rays float32[n,6] = dir, origin.
    Returns int32[n,4] = kind, index, t bits, 0 (same record as ptamd_trace_rays)."""
    lib = load()
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    out = np.zeros((len(rays), 4), dtype=np.int32)
    lib.or_intersect_batch(C.byref(scene.c), rays.ctypes.data, len(rays), out.ctypes.data)
    return out
