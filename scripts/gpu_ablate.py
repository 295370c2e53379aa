"""Where does the time go?  Launch-time ablations on the GPU (not part of the test-suite)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import cuda_pathtracer_amd as P
from helpers import make_scene

W, H = 1920, 1080
ctx = P.Context(0)
indoor = P.HostScene.load(os.path.join(ROOT, "assets", "indoor.scene"))
empty = make_scene(P, np.zeros((0, 3, 3), np.float32), camera=dict(position=tuple(indoor.camera["position"]), dir=tuple(indoor.camera["dir"]), fov_x=float(indoor.camera["fov_x"]), aperture=0.01, focus_dist=3.555))
cube = P.cubemap_from_color()
cid = ctx.upload_cubemap(cube)

def timeit(hs, sid, bounces, kernel, moved=False, n=20):
    fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), W, H)
    st = torch.cuda.current_stream()
    def go(k):
        l = ctx.make_launch(fr.surface, fr.accum, sid, cid, hs.camera_struct(), W, H, frame_nb=k, bounces=bounces, moved=moved, kernel=kernel, stream=st)
        ctx.raytrace_ex(l)
    for k in range(1, 4): go(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for k in range(1, n + 1): go(k)
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    l = ctx.make_launch(fr.surface, fr.accum, sid, cid, hs.camera_struct(), W, H, frame_nb=1, bounces=bounces, moved=moved, kernel=kernel)
    s = ctx.raytrace_stats(l)
    return ms, s

sid_e = ctx.upload_scene(empty); sid_i = ctx.upload_scene(indoor)
for name, kern in (("tile", P.KERNEL_BVH), ("blockwise", P.KERNEL_BVH_BLOCKWISE)):
    ms, s = timeit(empty, sid_e, 4, kern); print(f"{name:10s} empty scene B=4      : {ms:.4f} ms/launch", flush=True)
    ms, s = timeit(indoor, sid_i, 1, kern, moved=True); print(f"{name:10s} indoor moved(preview): {ms:.4f} ms/launch rays {s['rays']} nodes/ray {s['nodes_visited']/max(s['rays'],1):.2f}")
    for B in (1, 2, 3, 4, 8):
        ms, s = timeit(indoor, sid_i, B, kern)
        print(f"{name:10s} indoor B={B}: {ms:.4f} ms/launch  rays {s['rays']} (mesh hits {s['mesh_hits']}) nodes/ray {s['nodes_visited']/max(s['rays'],1):.2f} tris/ray {s['tris_tested']/max(s['rays'],1):.2f} node-lane-util {s['nodes_visited']/max(64*s['wave_node_iters'],1):.3f} tri-lane-util {s['tris_tested']/max(64*s['wave_tri_iters'],1):.3f} wave-iters node {s['wave_node_iters']} tri {s['wave_tri_iters']} idle-slot shares: unstarted {s['idle_unstarted']/max(64*s['wave_node_iters'],1):.3f} finished {s['idle_finished']/max(64*s['wave_node_iters'],1):.3f} parked {s['idle_parked']/max(64*s['wave_node_iters'],1):.3f}", flush=True)
