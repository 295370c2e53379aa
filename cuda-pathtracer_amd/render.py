"""Host-side mirror of the reference's render boundary.

Reference interface (cuda_opengl/include/shaders/raytrace.h:9-17):

    cudaError_t raytrace(cudaArray_const_t array, const scene::Scenes& scenes,
                         unsigned scene_id, const std::vector<scene::Cubemap>& cubemaps,
                         int cubemap_id, const scene::Camera* cam, unsigned width,
                         unsigned height, cudaStream_t stream, float3* temporal_framebuffer,
                         bool moved, unsigned post_id);
    void setupFunctionTables();

`Context.raytrace` keeps the same argument meaning (the scene/cubemap tables live in the
context, as GPUProcessor owns them in the reference: gpu_processor.cpp:271-331), launches
asynchronously on the given stream and borrows the caller's device buffers.  torch is used
only to own device memory and streams; every pixel is produced by libptamd.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import native as N
from .scene import HostScene

POST_NONE, POST_GRAYSCALE, POST_SEPIA, POST_INVERT = 0, 1, 2, 3
REFERENCE_BOUNCES = 3  # static_samples = 1 -> max_bounces = 3 (raytrace.cu:243,66)


def _ptr(buf) -> int:
    """Device address of a torch tensor / raw int pointer."""
    if isinstance(buf, int):
        return buf
    if hasattr(buf, "data_ptr"):
        if not buf.is_cuda:
            raise ValueError("output buffers must live in device memory (no CPU path exists)")
        if not buf.is_contiguous():
            raise ValueError("output buffers must be contiguous")
        return buf.data_ptr()
    raise TypeError(f"unsupported buffer type {type(buf)!r}")


def _stream_handle(stream) -> int:
    if stream is None:
        return 0
    if isinstance(stream, int):
        return stream
    return int(stream.cuda_stream)  # torch.cuda.Stream


class Context:
    """One device context = the device-side state GPUProcessor holds for raytrace()."""

    def __init__(self, device: int = 0):
        self._lib = N.load()
        h = C.c_void_p()
        N.check(self._lib.ptamd_create(device, C.byref(h)))
        self._h = h
        self.device = device

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ptamd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- uploads (scene.cpp:370-391, gpu_processor.cpp:68-238)
    def upload_scene(self, scene: HostScene) -> int:
        d = scene.desc()
        sid = C.c_uint32()
        N.check(self._lib.ptamd_upload_scene(self._h, C.byref(d), C.byref(sid)))
        return sid.value

    def upload_cubemap(self, faces: np.ndarray) -> int:
        faces = np.ascontiguousarray(faces, dtype=np.float32)
        if faces.ndim != 4 or faces.shape[0] != 6 or faces.shape[1] != faces.shape[2] or faces.shape[3] != 4:
            raise ValueError("cubemap must be float32[6, size, size, 4]")
        cid = C.c_uint32()
        N.check(self._lib.ptamd_upload_cubemap(self._h, faces.ctypes.data_as(C.POINTER(C.c_float)),
                                               faces.shape[1], C.byref(cid)))
        return cid.value

    def setup_function_tables(self) -> None:
        """setupFunctionTables() (raytrace.cu:360-375)."""
        N.check(self._lib.ptamd_setup_function_tables(self._h))

    def scene_info(self, scene_id: int) -> dict:
        info = N.SceneInfo()
        N.check(self._lib.ptamd_scene_info_get(self._h, scene_id, C.byref(info)))
        return {n: getattr(info, n) for n, _ in N.SceneInfo._fields_}

    # ---- the hot path
    def raytrace(self, array, scene_id: int, cubemap_id: int, cam: N.Camera, width: int, height: int,
                 stream, temporal_framebuffer, moved: bool, post_id: int) -> None:
        """One reference raytrace() call: 1 spp, context-held frame counter, 3 bounces."""
        N.check(self._lib.ptamd_raytrace(self._h, _ptr(array), scene_id, cubemap_id, C.byref(cam), width, height,
                                         _stream_handle(stream), _ptr(temporal_framebuffer),
                                         1 if moved else 0, post_id))

    def reset_frame_counter(self) -> None:
        N.check(self._lib.ptamd_reset_frame_counter(self._h))

    def make_launch(self, array, temporal_framebuffer, scene_id: int, cubemap_id: int, cam: N.Camera,
                    width: int, height: int, frame_nb: int, bounces: int = REFERENCE_BOUNCES,
                    moved: bool = False, post_id: int = POST_NONE, stream=None,
                    rows: Optional[tuple] = None, kernel: int = N.KERNEL_AUTO,
                    band_local_buffers: bool = False, frame_count: int = 1, machine_share: int = 0,
                    interleave: Optional[tuple] = None, reset_accumulation: bool = False, no_pipelining: bool = False) -> N.Launch:
        """interleave = (ranks, rank, band_rows): render the interleaved bands of `rank` (ptamd_launch.interleave_*)."""
        l = N.Launch()
        l.surface_rgba8 = _ptr(array)
        l.temporal_framebuffer = _ptr(temporal_framebuffer)
        l.stream = _stream_handle(stream)
        l.camera = cam
        l.scene_id, l.cubemap_id = scene_id, cubemap_id
        l.width, l.height = width, height
        l.row_begin, l.row_end = rows if rows is not None else (0, height)
        l.frame_nb, l.bounces = frame_nb, bounces
        l.moved, l.post_id, l.kernel = (1 if moved else 0), post_id, kernel
        l.band_local_buffers = 1 if band_local_buffers else 0
        l.frame_count = frame_count
        l.machine_share = machine_share
        if interleave is not None:
            l.interleave_ranks, l.interleave_rank, l.interleave_rows = interleave
        l.reset_accumulation = 1 if reset_accumulation else 0
        l.no_pipelining = 1 if no_pipelining else 0
        return l

    def raytrace_ex(self, launch: N.Launch) -> None:
        N.check(self._lib.ptamd_raytrace_ex(self._h, C.byref(launch)))

    def trace_rays_queue(self, scene_id: int, rays, out, config: int = 0, refill_min: int = 8, stream=None) -> int:
        """The walk-only kernel fed from a ray queue (ptamd_trace_rays_queue): rays float32[n, 6] and out int32[n, 4] are device
        tensors; asynchronous on `stream`.  Returns the waves resident per CU."""
        waves = C.c_uint32(0)
        N.check(self._lib.ptamd_trace_rays_queue(self._h, scene_id, _ptr(rays), int(rays.shape[0]), _ptr(out), config, refill_min,
                                                 _stream_handle(stream), C.byref(waves)))
        return int(waves.value)

    def release_captured(self, stream=None) -> None:
        """The graphs captured on `stream` are gone: its sample slab and ring slots are no longer pinned (ptamd_release_captured)."""
        N.check(self._lib.ptamd_release_captured(self._h, _stream_handle(stream)))

    def raytrace_stats(self, launch: N.Launch) -> dict:
        st = N.TraceStats()
        N.check(self._lib.ptamd_raytrace_stats(self._h, C.byref(launch), C.byref(st)))
        return {n: getattr(st, n) for n, _ in N.TraceStats._fields_}

    def phase_cycles(self) -> dict:
        """Shader-clock cycles by phase, summed over the waves of the last raytrace_stats launch of the restart kernel."""
        v = (C.c_uint64 * 12)()
        N.check(self._lib.ptamd_phase_cycles(self._h, v))
        d = dict(zip(("refill", "box_phases", "leaf_phases", "light_loop", "round_loop", "leaf_phases_entered", "node_fetches", "visits", "shading_record_fetch", "path_post_and_parking", "path_post_to_bsdf", "bsdf_sample"), [int(x) for x in v]))
        d["lights_and_shading"] = d["round_loop"] - d["refill"] - d["box_phases"] - d["leaf_phases"]   # (what the three stamps leave: light loop + path_post + round bookkeeping)
        return d

    def trace_rays(self, scene_id: int, rays: np.ndarray, kernel: int = N.KERNEL_BVH) -> np.ndarray:
        """rays float32[n,6] = dir.xyz, origin.xyz -> int32[n,4] = kind, index, t bits, 0."""
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        out = np.zeros((len(rays), 4), dtype=np.int32)
        N.check(self._lib.ptamd_trace_rays(self._h, scene_id, kernel, rays.ctypes.data_as(C.POINTER(C.c_float)),
                                           len(rays), out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def set_timeline(self, max_waves: int) -> None:
        """Per-wave time stamps of the default kernel's launches (ptamd_set_timeline); 0 = off."""
        N.check(self._lib.ptamd_set_timeline(self._h, max_waves))

    def read_timeline(self, n_waves: int):
        """uint64[n_waves, 4] = (entry, scene staged, no ticket left, exit) in device-clock ticks, and the clock in kHz;
        synchronises and clears the buffer.  Rows of waves the last launches did not have are 0."""
        out = np.zeros((n_waves, 4), dtype=np.uint64)
        khz = C.c_uint32()
        N.check(self._lib.ptamd_read_timeline(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), n_waves, C.byref(khz)))
        return out, khz.value

    def gamma_table_selftest(self):
        """(values checked, mismatches) of the tonemap's gamma table against the pow sequence it replaces; synchronises."""
        n, bad = C.c_uint64(), C.c_uint64()
        N.check(self._lib.ptamd_gamma_table_selftest(self._h, C.byref(n), C.byref(bad)))
        return n.value, bad.value

    def device_error_count(self) -> int:
        """Protocol time-outs of the split kernel since creation (0 unless there is a bug); synchronises."""
        v = C.c_uint64()
        N.check(self._lib.ptamd_device_error_count(self._h, C.byref(v)))
        return v.value

    def synchronize(self, stream=None) -> None:
        N.check(self._lib.ptamd_stream_synchronize(self._h, _stream_handle(stream)))


def host_bvh_trace(scene: HostScene, rays: np.ndarray):
    """Host mirror of the device BVH walk (test hook, no GPU): returns (int32[n,4], nodes, tris)."""
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    out = np.zeros((len(rays), 4), dtype=np.int32)
    counters = (C.c_uint64 * 2)(0, 0)
    N.check(N.load().ptamd_host_bvh_trace(scene.faces.ctypes.data_as(C.POINTER(N.Face)), len(scene.faces),
                                          rays.ctypes.data_as(C.POINTER(C.c_float)), len(rays),
                                          out.ctypes.data_as(C.POINTER(C.c_int32)), counters))
    return out, counters[0], counters[1]


def host_bvh4_trace(scene: HostScene, rays: np.ndarray):
    """Host mirror of the device's four-wide stack walk (test hook, no GPU): (int32[n,4], wide nodes visited,
    triangles tested, depth of the wide tree)."""
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    out = np.zeros((len(rays), 4), dtype=np.int32)
    counters = (C.c_uint64 * 5)(0, 0, 0, 0, 0)
    N.check(N.load().ptamd_host_bvh4_trace(scene.faces.ctypes.data_as(C.POINTER(N.Face)), len(scene.faces),
                                           rays.ctypes.data_as(C.POINTER(C.c_float)), len(rays),
                                           out.ctypes.data_as(C.POINTER(C.c_int32)), counters))
    host_bvh4_trace.top_visits = (counters[3], counters[4])   # visits to the first 85 / 341 nodes
    return out, counters[0], counters[1], counters[2]


def host_bvh4q_trace(scene: HostScene, rays: np.ndarray):
    """Host mirror of the four-wide walk over the 64-byte quantised nodes (test hook, no GPU): as host_bvh4_trace."""
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    out = np.zeros((len(rays), 4), dtype=np.int32)
    counters = (C.c_uint64 * 5)(0, 0, 0, 0, 0)
    N.check(N.load().ptamd_host_bvh4q_trace(scene.faces.ctypes.data_as(C.POINTER(N.Face)), len(scene.faces),
                                            rays.ctypes.data_as(C.POINTER(C.c_float)), len(rays),
                                            out.ctypes.data_as(C.POINTER(C.c_int32)), counters))
    host_bvh4q_trace.top_visits = (counters[3], counters[4])
    return out, counters[0], counters[1], counters[2]


def host_bvh8_trace(scene: HostScene, rays: np.ndarray):
    """Host mirror of the device's eight-wide walk over quantised nodes (test hook, no GPU): (int32[n,4], nodes visited,
    triangles tested, depth, node count); .top_visits = visits to the first 73 / 585 nodes."""
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    out = np.zeros((len(rays), 4), dtype=np.int32)
    counters = (C.c_uint64 * 6)(0, 0, 0, 0, 0, 0)
    N.check(N.load().ptamd_host_bvh8_trace(scene.faces.ctypes.data_as(C.POINTER(N.Face)), len(scene.faces),
                                           rays.ctypes.data_as(C.POINTER(C.c_float)), len(rays),
                                           out.ctypes.data_as(C.POINTER(C.c_int32)), counters))
    host_bvh8_trace.top_visits = (counters[3], counters[4])
    return out, counters[0], counters[1], counters[2], counters[5]


def interleaved_rows(height: int, ranks: int, rank: int, band_rows: int) -> int:
    """Rows the interleaved bands of `rank` hold (ptamd_interleaved_rows)."""
    return N.load().ptamd_interleaved_rows(height, ranks, rank, band_rows)


def wang_hash(a: int) -> int:
    return N.load().ptamd_wang_hash(a & 0xFFFFFFFF)


class FrameRenderer:
    """N-spp frame driver: N consecutive static launches with frame seeds 1..N accumulating
    into the temporal framebuffer — how the reference converges an image (raytrace.cu:255-258,
    296-300).  Buffers are torch tensors on the context's device."""

    def __init__(self, ctx: Context, scene_id: int, cubemap_id: int, cam: N.Camera, width: int, height: int,
                 rows: Optional[tuple] = None, band_local: bool = False, machine_share: int = 0,
                 interleave: Optional[tuple] = None, surface=None):
        """interleave = (ranks, rank, band_rows): this renderer owns the interleaved bands of `rank` (band j of the frame
        belongs to rank j % ranks); its buffers hold those bands one after the other.
        surface: render into this uint8[rows, width, 4] device tensor instead of allocating one (e.g. the send buffer
        of a BandGather: no staging copy per frame)."""
        import torch
        self.ctx, self.scene_id, self.cubemap_id, self.cam = ctx, scene_id, cubemap_id, cam
        self.width, self.height = width, height
        self.rows = rows if rows is not None else (0, height)
        self.band_local = band_local
        self.machine_share = machine_share   # > 1: this renderer shares the GPU with that many launches in flight
        self.interleave = interleave
        if interleave is not None:
            self.band_local = band_local = True
            self.rows = (0, height)
        n_rows = (self.rows[1] - self.rows[0]) if band_local else height
        if interleave is not None:
            n_rows = interleaved_rows(height, *interleave)
        dev = torch.device("cuda", ctx.device)
        if surface is not None:
            if tuple(surface.shape) != (n_rows, width, 4) or surface.dtype != torch.uint8 or not surface.is_contiguous():
                raise ValueError(f"surface must be a contiguous uint8[{n_rows}, {width}, 4] tensor")
            self.surface = surface
        else:
            self.surface = torch.zeros((n_rows, width, 4), dtype=torch.uint8, device=dev)
        self.accum = torch.zeros((n_rows, width, 3), dtype=torch.float32, device=dev)

    def reset(self) -> None:
        self.accum.zero_()
        self.surface.zero_()

    def render(self, spp: int, bounces: int = REFERENCE_BOUNCES, post_id: int = POST_NONE,
               kernel: int = N.KERNEL_AUTO, stream=None, first_frame: int = 1, batched: bool = False,
               reset: bool = False) -> None:
        """`batched=True` issues the spp frames as ONE launch (ptamd_launch.frame_count): same
        accumulator and final surface bit for bit, no kernel tails between frames.  `reset=True` starts a new
        accumulation: the first launch treats the accumulator as zero (ptamd_launch.reset_accumulation) — the same
        result as clearing it first."""
        if batched and spp > 1:
            l = self.ctx.make_launch(self.surface, self.accum, self.scene_id, self.cubemap_id, self.cam,
                                     self.width, self.height, frame_nb=first_frame, bounces=bounces, post_id=post_id,
                                     stream=stream, rows=self.rows, kernel=kernel,
                                     band_local_buffers=self.band_local, frame_count=spp, machine_share=self.machine_share,
                                     interleave=self.interleave, reset_accumulation=reset)
            self.ctx.raytrace_ex(l)
            return
        for k in range(first_frame, first_frame + spp):
            l = self.ctx.make_launch(self.surface, self.accum, self.scene_id, self.cubemap_id, self.cam,
                                     self.width, self.height, frame_nb=k, bounces=bounces, post_id=post_id,
                                     stream=stream, rows=self.rows, kernel=kernel,
                                     band_local_buffers=self.band_local, machine_share=self.machine_share,
                                     interleave=self.interleave, reset_accumulation=reset and k == first_frame)
            self.ctx.raytrace_ex(l)
