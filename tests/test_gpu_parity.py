"""GPU parity tests proper: the HIP megakernel, called through the C-ABI, against the CPU oracle
on the same inputs and against the committed golden fixtures.

Tolerance: ZERO.  Accumulators (float32) and RGBA8 surfaces must be bit-identical — both sides
execute the same IEEE binary32/binary64 operation sequence (DESIGN.md "Defined arithmetic").
At BASELINE.json's full size (1920x1080, 4 spp, 4 bounces), where the oracle would take
minutes, parity is carried by size-independent properties: BVH kernel == brute-force kernel
(the reference algorithm, itself oracle-checked at small sizes), row-band splits == full frame,
N spp == sum of N one-spp launches, and an oracle check of a crop of full-resolution rows.
"""
import os

import numpy as np
import pytest

from conftest import ASSETS, GOLDEN
from golden.make_golden import ALL, case_inputs
from helpers import make_scene, random_rays, random_soup, synthetic_cubemap

pytestmark = pytest.mark.gpu


def gpu_render(P, ctx, hs, cube, W, H, spp, bounces, kernel, moved=False, post_id=0, rows=None, band_local=False,
               first_frame=1, ids=None):
    import torch
    sid, cid = ids if ids is not None else (ctx.upload_scene(hs), ctx.upload_cubemap(cube))
    fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), W, H, rows=rows, band_local=band_local)
    for k in range(first_frame, first_frame + spp):
        l = ctx.make_launch(fr.surface, fr.accum, sid, cid, hs.camera_struct(), W, H, frame_nb=k, bounces=bounces,
                            moved=moved, post_id=post_id, rows=fr.rows, kernel=kernel, band_local_buffers=band_local)
        ctx.raytrace_ex(l)
    torch.cuda.synchronize()
    return fr.accum.cpu().numpy(), fr.surface.cpu().numpy()


def assert_same(acc, rgba, ref_acc, ref_rgba, what=""):
    bad = (acc.view(np.uint32) != ref_acc.view(np.uint32)).any(axis=2)
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} accumulator pixels differ (first {np.argwhere(bad)[:3].tolist()})"
    np.testing.assert_array_equal(rgba, ref_rgba, err_msg=what)


KERNELS = ["persistent", "restart", "split", "blockwise", "bvh", "brute"]


def batched_ok():
    """frame_count > 1 needs a persistent kernel behind PTAMD_KERNEL_AUTO: under the tuning knob
    PTAMD_DEFAULT_KERNEL=1/2/4 (scripts/gpu_knobtest.sh) the batched cases do not apply."""
    return os.environ.get("PTAMD_DEFAULT_KERNEL", "3") in ("3", "5", "6")


def needs_batched_default():
    if not batched_ok():
        pytest.skip("PTAMD_DEFAULT_KERNEL selects a kernel that cannot batch frames")


def kid(P, name):
    return {"bvh": P.KERNEL_BVH, "brute": P.KERNEL_BRUTE_FORCE, "persistent": P.KERNEL_BVH_PERSISTENT,
            "blockwise": P.KERNEL_BVH_BLOCKWISE, "split": P.KERNEL_BVH_SPLIT, "restart": P.KERNEL_BVH_RESTART}[name]


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", ALL)
def test_hip_equals_golden_and_oracle(P, O, gpu_ctx, name, kernel):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    hs, cube = case_inputs(name)
    W, H, spp, B = int(g["W"]), int(g["H"]), int(g["spp"]), int(g["bounces"])
    acc, rgba = gpu_render(P, gpu_ctx, hs, cube, W, H, spp, B, kid(P, kernel), moved=bool(g["moved"]), post_id=int(g["post_id"]))
    assert_same(acc, rgba, g["accum"], g["rgba"], f"{name}/{kernel} vs golden")
    ref_acc, ref_rgba = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), W, H,
                                 spp=spp, bounces=B, moved=bool(g["moved"]), post_id=int(g["post_id"]))
    assert_same(acc, rgba, ref_acc, ref_rgba, f"{name}/{kernel} vs oracle")


@pytest.mark.parametrize("kernel", KERNELS)
def test_config1_256x256_1spp_2bounces(P, O, gpu_ctx, indoor, kernel):
    """BASELINE.json configs[0]: indoor, 256x256, 1 spp, 2 bounces."""
    cube = P.cubemap_for_scene(indoor)
    acc, rgba = gpu_render(P, gpu_ctx, indoor, cube, 256, 256, 1, 2, kid(P, kernel))
    ref = O.render(O.OracleScene.from_host_scene(indoor, cube), O.camera_from_record(indoor.camera), 256, 256, spp=1, bounces=2)
    assert_same(acc, rgba, *ref, f"config1/{kernel}")


@pytest.mark.parametrize("W,H", [(1, 1), (15, 17), (16, 16), (33, 9), (130, 47)])
def test_ragged_frame_sizes(P, O, gpu_ctx, indoor, W, H):
    """Frames that are not multiples of the 16x16 tile, down to a single pixel (Q13 padded grid)."""
    cube = P.cubemap_for_scene(indoor)
    ref = O.render(O.OracleScene.from_host_scene(indoor, cube), O.camera_from_record(indoor.camera), W, H, spp=2, bounces=3)
    for kernel in KERNELS:
        acc, rgba = gpu_render(P, gpu_ctx, indoor, cube, W, H, 2, 3, kid(P, kernel))
        assert_same(acc, rgba, *ref, f"{W}x{H}/{kernel}")


def test_all_post_processes_and_deep_bounces(P, O, gpu_ctx):
    hs = P.HostScene.load(os.path.join(ASSETS, "color_sample.scene"))   # has an ior 1.5 material: refraction branch
    cube = synthetic_cubemap(np.random.default_rng(11), 4)               # bilinear env lookups
    osc, ocam = O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera)
    for post in (0, 1, 2, 3):
        ref = O.render(osc, ocam, 48, 32, spp=2, bounces=8, post_id=post)
        acc, rgba = gpu_render(P, gpu_ctx, hs, cube, 48, 32, 2, 8, P.KERNEL_AUTO, post_id=post)
        assert_same(acc, rgba, *ref, f"post {post}")


def test_degenerate_scenes(P, O, gpu_ctx):
    """Empty scene (env only), lights only, one triangle, faces with NaN tangents + normal maps."""
    rng = np.random.default_rng(21)
    cube = synthetic_cubemap(rng, 2)
    lights = [((0.0, 0.5, 1.0), (1.0, 0.9, 0.8), 4.0, 0.8)]
    cases = {
        "empty": make_scene(P, np.zeros((0, 3, 3), np.float32)),
        "lights_only": make_scene(P, np.zeros((0, 3, 3), np.float32), lights=lights),
        "one_tri": make_scene(P, np.float32([[[-2, -2, 0], [2, -2, 0], [0, 2, 0]]]), lights=lights),
    }
    # degenerate UVs -> inf/NaN tangents feeding the normal-map path
    tris = random_soup(rng, 24, extent=1.0, size=0.8)
    uvs = np.zeros((24, 3, 2), np.float32)
    cases["nan_tangent_nmap"] = make_scene(P, tris, uvs=uvs, materials=[(0, 1, 1.0)],
                                           textures=[np.float32([[[0.5, 0.6, 0.7, 0.3]]]), rng.uniform(0, 1, (4, 4, 3)).astype(np.float32)],
                                           lights=lights)
    # black albedo: throughput 0 -> p = 0 -> throughput *= 1/0 = NaN -> clamp maps NaN to 1.0: white pixels
    cases["black_albedo_nan"] = make_scene(P, random_soup(rng, 40, extent=1.2, size=0.9),
                                           textures=[np.float32([[[0.0, 0.0, 0.0, 0.25]]])], lights=lights)
    for name, hs in cases.items():
        ref = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), 40, 24, spp=2, bounces=4)
        if name == "black_albedo_nan":
            assert (ref[0] == 2.0).all(axis=2).sum() > 20      # pixels whose two samples both clamped NaN -> 1.0
        for kernel in KERNELS:
            acc, rgba = gpu_render(P, gpu_ctx, hs, cube, 40, 24, 2, 4, kid(P, kernel))
            # NaN accumulators cannot occur (clamp maps NaN to 1.0, raytrace.cu:248), so bit compare is total
            assert_same(acc, rgba, *ref, f"{name}/{kernel}")


def test_reference_raytrace_entry_point_counts_frames(P, O, gpu_ctx, indoor):
    """ptamd_raytrace == raytrace(): context-held frame counter (raytrace.cu:296-300), 3 bounces,
    `moved` resets the counter and the accumulator (preview mode)."""
    import torch
    cube = P.cubemap_for_scene(indoor)
    sid, cid = gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube)
    W, H = 72, 40
    dev = torch.device("cuda", 0)
    surf = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    tfb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    cam = indoor.camera_struct()
    gpu_ctx.reset_frame_counter()
    for _ in range(3):
        gpu_ctx.raytrace(surf, sid, cid, cam, W, H, None, tfb, False, 0)
    torch.cuda.synchronize()
    osc, ocam = O.OracleScene.from_host_scene(indoor, cube), O.camera_from_record(indoor.camera)
    ref = O.render(osc, ocam, W, H, spp=3, bounces=3)
    assert_same(tfb.cpu().numpy(), surf.cpu().numpy(), *ref, "3 static frames")
    # camera moved: counter back to 1, preview sample replaces the accumulator
    gpu_ctx.raytrace(surf, sid, cid, cam, W, H, None, tfb, True, 0)
    torch.cuda.synchronize()
    ref = O.render(osc, ocam, W, H, spp=1, bounces=3, moved=True)
    assert_same(tfb.cpu().numpy(), surf.cpu().numpy(), *ref, "moved frame")
    # then static again: frame 2 accumulates on top of the preview (reference behaviour)
    gpu_ctx.raytrace(surf, sid, cid, cam, W, H, None, tfb, False, 0)
    torch.cuda.synchronize()
    acc = ref[0].copy()
    ref2 = O.render(osc, ocam, W, H, spp=1, bounces=3, first_frame=2, accum=acc)
    assert_same(tfb.cpu().numpy(), surf.cpu().numpy(), *ref2, "static frame after move")


def test_row_band_split_is_bit_identical(P, gpu_ctx, indoor):
    """Multi-GPU contract (L3): any split into row bands, full-frame or band-local buffers,
    reproduces the single-launch frame bit for bit (global-coordinate seeds)."""
    cube = P.cubemap_for_scene(indoor)
    ids = (gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube))
    W, H, spp, B = 200, 121, 2, 4
    full_acc, full_rgba = gpu_render(P, gpu_ctx, indoor, cube, W, H, spp, B, P.KERNEL_BVH, ids=ids)
    for k in (P.KERNEL_BVH_PERSISTENT, P.KERNEL_BVH_RESTART, P.KERNEL_BVH_BLOCKWISE, P.KERNEL_BVH_SPLIT):
        pa, pr = gpu_render(P, gpu_ctx, indoor, cube, W, H, spp, B, k, ids=ids)
        assert_same(pa, pr, full_acc, full_rgba, f"kernel {k} vs tile kernel")
    for world in (2, 3, 8):
        accs, rgbas = [], []
        for rows in P.row_bands(H, world):
            a, r = gpu_render(P, gpu_ctx, indoor, cube, W, H, spp, B, P.KERNEL_AUTO, rows=rows, band_local=True, ids=ids)
            assert a.shape[0] == rows[1] - rows[0]
            accs.append(a)
            rgbas.append(r)
        np.testing.assert_array_equal(np.concatenate(rgbas, axis=0), full_rgba)
        # accumulator bands are stored row-flipped: band r covers tfb rows [H-e, H-b)
        np.testing.assert_array_equal(np.concatenate(accs[::-1], axis=0).view(np.uint32), full_acc.view(np.uint32))
    # full-frame buffers, band launches
    import torch
    fr = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H)
    for k in range(1, spp + 1):
        for rows in P.row_bands(H, 5):
            gpu_ctx.raytrace_ex(gpu_ctx.make_launch(fr.surface, fr.accum, *ids, indoor.camera_struct(), W, H, frame_nb=k,
                                                    bounces=B, rows=rows, kernel=P.KERNEL_BVH))
    torch.cuda.synchronize()
    assert_same(fr.accum.cpu().numpy(), fr.surface.cpu().numpy(), full_acc, full_rgba, "band launches into full buffers")


def test_interleaved_band_split_is_bit_identical(P, gpu_ctx, indoor):
    """SURVEY §8-e's interleaved assignment (band j of 8/16/24 rows -> rank j % R): every rank's launch renders its bands
    into band-local buffers; put back in frame order, surfaces AND accumulators equal the single-launch frame, for
    sequential and batched launches, ragged frame heights included."""
    if os.environ.get("PTAMD_DEFAULT_KERNEL", "6") != "6":
        pytest.skip("interleaved bands need the restart kernel behind PTAMD_KERNEL_AUTO")
    import torch
    cube = P.cubemap_for_scene(indoor)
    ids = (gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube))
    for (W, H, spp, B) in ((200, 121, 2, 4), (1920, 1080, 4, 4)):
        full_acc, full_rgba = gpu_render(P, gpu_ctx, indoor, cube, W, H, spp, B, P.KERNEL_BVH, ids=ids)
        for world, br in ((2, 16), (3, 8), (8, 16), (5, 24)):
            rgba = np.zeros_like(full_rgba)
            acc = np.zeros_like(full_acc)
            for rank in range(world):
                fr = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H, interleave=(world, rank, br))
                fr.render(spp=spp, bounces=B, batched=batched_ok() and rank % 2 == 0)
                torch.cuda.synchronize()
                s, a = fr.surface.cpu().numpy(), fr.accum.cpu().numpy()
                bands = P.interleaved_bands(H, world, rank, br)
                assert s.shape[0] == sum(e - b for b, e in bands)
                local = 0
                for b, e in bands:
                    rgba[b:e] = s[local:local + (e - b)]
                    # the accumulator is stored row-flipped (raytrace.cu:252): local row i lives at rows - 1 - i
                    acc[H - e:H - b] = a[s.shape[0] - (local + (e - b)):s.shape[0] - local]
                    local += e - b
            assert_same(acc, rgba, full_acc, full_rgba, f"interleaved {world} ranks x {br} rows, {W}x{H}")
    l = gpu_ctx.make_launch(fr.surface, fr.accum, *ids, indoor.camera_struct(), W, H, frame_nb=1, band_local_buffers=True,
                            interleave=(4, 4, 16))
    with pytest.raises(P.PtamdError):
        gpu_ctx.raytrace_ex(l)                                   # rank out of range
    l = gpu_ctx.make_launch(fr.surface, fr.accum, *ids, indoor.camera_struct(), W, H, frame_nb=1, band_local_buffers=True,
                            interleave=(4, 1, 12))
    with pytest.raises(P.PtamdError):
        gpu_ctx.raytrace_ex(l)                                   # band rows not a multiple of 8


def test_batched_frames_equal_consecutive_launches(P, O, gpu_ctx, indoor):
    """ptamd_launch.frame_count = N: one launch + resolve == N consecutive raytrace() calls, bit for bit
    (accumulator and final surface), also on top of a non-zero accumulator and for row bands."""
    needs_batched_default()
    import torch
    cube = P.cubemap_for_scene(indoor)
    ids = (gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube))
    for (W, H, spp, B, post, rows, band_local) in ((200, 121, 4, 4, 0, None, False), (97, 50, 7, 3, 2, (13, 41), True),
                                                  (1920, 1080, 4, 4, 0, None, False)):
        seq = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H, rows=rows, band_local=band_local)
        bat = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H, rows=rows, band_local=band_local)
        seq.render(spp=2, bounces=B, post_id=post)                       # pre-existing accumulation
        bat.render(spp=2, bounces=B, post_id=post)
        seq.render(spp=spp, bounces=B, post_id=post, first_frame=3)
        bat.render(spp=spp, bounces=B, post_id=post, first_frame=3, batched=True)
        torch.cuda.synchronize()
        assert_same(bat.accum.cpu().numpy(), bat.surface.cpu().numpy(), seq.accum.cpu().numpy(), seq.surface.cpu().numpy(),
                    f"batched {W}x{H} spp{spp}")
    # oracle check of the batched path itself
    ref = O.render(O.OracleScene.from_host_scene(indoor, cube), O.camera_from_record(indoor.camera), 80, 48, spp=5, bounces=3)
    fr = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), 80, 48)
    fr.render(spp=5, bounces=3, batched=True)
    torch.cuda.synchronize()
    assert_same(fr.accum.cpu().numpy(), fr.surface.cpu().numpy(), *ref, "batched vs oracle")
    # the split kernel batches frames the same way
    sp = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), 80, 48)
    sp.render(spp=5, bounces=3, batched=True, kernel=P.KERNEL_BVH_SPLIT)
    torch.cuda.synchronize()
    assert_same(sp.accum.cpu().numpy(), sp.surface.cpu().numpy(), *ref, "batched split kernel vs oracle")
    assert gpu_ctx.device_error_count() == 0          # no producer/consumer spin ever timed out
    l = gpu_ctx.make_launch(fr.surface, fr.accum, *ids, indoor.camera_struct(), 80, 48, frame_nb=1, frame_count=3, moved=True)
    with pytest.raises(P.PtamdError):
        gpu_ctx.raytrace_ex(l)
    l = gpu_ctx.make_launch(fr.surface, fr.accum, *ids, indoor.camera_struct(), 80, 48, frame_nb=1, frame_count=3, kernel=P.KERNEL_BVH)
    with pytest.raises(P.PtamdError):
        gpu_ctx.raytrace_ex(l)


def test_reset_accumulation_equals_a_cleared_accumulator(P, gpu_ctx, indoor):
    """ptamd_launch.reset_accumulation: the launch that starts an accumulation treats the temporal framebuffer as zero —
    bit for bit what clearing it first gives, whatever it held (NaNs included), for every kernel, sequential and batched,
    full frame, band and interleaved rows."""
    import torch
    cube = P.cubemap_for_scene(indoor)
    ids = (gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube))
    W, H, spp, B = 200, 121, 3, 4
    for kernel in KERNELS:
        for batched in ((False, True) if (kernel in ("persistent", "restart", "split")) else (False,)):
            for kw in (dict(), dict(rows=(13, 77), band_local=True)) + ((dict(interleave=(3, 1, 8)),) if kernel == "restart" else ()):
                clean = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H, **kw)
                clean.render(spp=spp, bounces=B, kernel=kid(P, kernel), batched=batched)
                dirty = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H, **kw)
                dirty.accum.fill_(float("nan"))
                dirty.accum[::2] = 123.5
                dirty.render(spp=spp, bounces=B, kernel=kid(P, kernel), batched=batched, reset=True)
                torch.cuda.synchronize()
                rows = clean.rows if "interleave" not in kw else (0, H)
                if kw.get("band_local") or "interleave" in kw:
                    a, b = dirty.accum.cpu().numpy(), clean.accum.cpu().numpy()
                    sa, sb = dirty.surface.cpu().numpy(), clean.surface.cpu().numpy()
                else:                                   # full-frame buffers, band launch: compare the rendered rows only
                    a, b = dirty.accum.cpu().numpy()[H - rows[1]:H - rows[0]], clean.accum.cpu().numpy()[H - rows[1]:H - rows[0]]
                    sa, sb = dirty.surface.cpu().numpy()[rows[0]:rows[1]], clean.surface.cpu().numpy()[rows[0]:rows[1]]
                assert_same(a, sa, b, sb, f"reset {kernel} batched={batched} {kw}")


def test_batched_launches_in_flight_on_one_context(P, gpu_ctx, indoor):
    """two batched launches of ONE context on different streams run concurrently (machine_share = 2); each
    stream has its own sample scratch, so neither frame sees the other's samples — including when a scratch regrows."""
    needs_batched_default()
    import torch
    cube = P.cubemap_for_scene(indoor)
    ids = (gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube))
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for (W, H, spp) in ((320, 200, 3), (640, 360, 5)):          # second round: bigger frames -> every scratch regrows
        want = {}
        for first in (1, 7, 13):
            fr = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H)
            fr.render(spp=spp, bounces=4, first_frame=first)
            torch.cuda.synchronize()
            want[first] = (fr.accum.cpu().numpy(), fr.surface.cpu().numpy())
        frs = [P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H, machine_share=2) for _ in streams]
        for rep in range(3):
            for fr in frs:
                fr.reset()
            torch.cuda.synchronize()
            for fr, st, first in zip(frs, streams, (1, 7, 13)):
                fr.render(spp=spp, bounces=4, first_frame=first, batched=True, stream=st)
            torch.cuda.synchronize()
            for fr, first in zip(frs, (1, 7, 13)):
                assert_same(fr.accum.cpu().numpy(), fr.surface.cpu().numpy(), *want[first], f"in-flight batched {W}x{H} frame {first} rep {rep}")


def test_batched_launch_is_graph_capturable(P, gpu_ctx, indoor):
    """A warmed-up batched launch (megakernel + resolve pass, ticket heads re-zeroed by the resolve pass) makes no
    synchronising or allocating HIP call, so a host may capture it into a hipGraph and replay it: every replay produces the
    eager launch's accumulator and surface bit for bit."""
    needs_batched_default()
    import torch
    W, H, spp = 640, 360, 4
    cube = P.cubemap_for_scene(indoor)
    ids = (gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube))
    eager = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H)
    eager.render(spp=spp, bounces=4, batched=True, reset=True)
    torch.cuda.synchronize()
    want = (eager.accum.cpu().numpy(), eager.surface.cpu().numpy())
    fr = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fr.render(spp=spp, bounces=4, batched=True, reset=True, stream=side)     # warm-up: this stream's sample scratch
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fr.render(spp=spp, bounces=4, batched=True, reset=True, stream=torch.cuda.current_stream())
    for rep in range(3):
        fr.accum.fill_(7.0)
        fr.surface.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert_same(fr.accum.cpu().numpy(), fr.surface.cpu().numpy(), *want, f"graph replay {rep}")


def test_trace_rays_device_equals_oracle(P, O, gpu_ctx):
    """Nearest-hit records (kind, index, t bits) of both device traversals == brute-force oracle,
    including light spheres."""
    rng = np.random.default_rng(17)
    for name in ("indoor", "island"):
        hs = P.HostScene.load(os.path.join(ASSETS, name + ".scene"))
        sid = gpu_ctx.upload_scene(hs)
        rays = random_rays(rng, 50000, extent=3.5)
        want = O.intersect(O.OracleScene.from_host_scene(hs, P.cubemap_from_color()), rays)
        for k in (P.KERNEL_BVH, P.KERNEL_BRUTE_FORCE, P.KERNEL_BVH_RESTART):     # _RESTART: the four-wide stack walk
            np.testing.assert_array_equal(gpu_ctx.trace_rays(sid, rays, k), want)
        assert (want[:, 0] == 1).sum() > 5000
    soup = make_scene(P, random_soup(rng, 3000, extent=3.0, size=0.25))       # does not fit in LDS: global-memory variant
    sid = gpu_ctx.upload_scene(soup)
    rays = random_rays(rng, 20000)
    want = O.intersect(O.OracleScene.from_host_scene(soup, P.cubemap_from_color()), rays)
    np.testing.assert_array_equal(gpu_ctx.trace_rays(sid, rays, P.KERNEL_BVH), want)
    np.testing.assert_array_equal(gpu_ctx.trace_rays(sid, rays, P.KERNEL_BVH_RESTART), want)
    assert gpu_ctx.scene_info(sid)["n_nodes4"] > 300 and gpu_ctx.scene_info(sid)["depth4"] >= 4


def test_wide_walk_with_a_spilling_stack(P, O, gpu_ctx, monkeypatch):
    """The four-wide walk keeps most of its per-lane stack in LDS and the rest in a global slab: with only two LDS entries
    per lane (PTAMD_STACK_LDS) nearly every push beyond the second goes through the slab — same pixels."""
    rng = np.random.default_rng(41)
    hs = make_scene(P, random_soup(rng, 2500, extent=2.5, size=0.3), lights=[((0.0, 3.0, 1.0), (1, 1, 1), 6.0, 0.7)])
    cube = synthetic_cubemap(rng, 4)
    ref = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), 96, 64, spp=2, bounces=4)
    monkeypatch.setenv("PTAMD_TUNING", "1")
    monkeypatch.setenv("PTAMD_STACK_LDS", "2")
    acc, rgba = gpu_render(P, gpu_ctx, hs, cube, 96, 64, 2, 4, P.KERNEL_BVH_RESTART)
    assert_same(acc, rgba, *ref, "wide walk, stack mostly in the global slab")
    monkeypatch.delenv("PTAMD_STACK_LDS")
    acc, rgba = gpu_render(P, gpu_ctx, hs, cube, 96, 64, 2, 4, P.KERNEL_BVH_RESTART)
    assert_same(acc, rgba, *ref, "wide walk, stack in LDS")


def test_eight_wide_quantised_walk_on_the_device(P, O, monkeypatch):
    """The eight-wide form of the tree (Bvh::nodes8: float origin, per-axis power-of-two scale, 8-bit planes, children in
    direction slots) walked by the device — a tuning knob since it measured 8 % slower than the four-wide walk:
    nearest-hit records == brute-force oracle, rendered frames == oracle, with the stack in LDS and mostly in the global
    slab, and a batched launch == consecutive launches."""
    import torch
    monkeypatch.setenv("PTAMD_TUNING", "1")
    monkeypatch.setenv("PTAMD_WIDE8", "1")
    rng = np.random.default_rng(43)
    with P.Context(0) as ctx:                       # (knobs are read when a context is created)
        soup = make_scene(P, random_soup(rng, 3000, extent=3.0, size=0.25))
        sid = ctx.upload_scene(soup)
        rays = random_rays(rng, 20000)
        rays[:40, 0] = 0.0
        rays[40:80, 1:3] = -0.0                     # zeros of either sign: the octant comes from the sign bit
        want = O.intersect(O.OracleScene.from_host_scene(soup, P.cubemap_from_color()), rays)
        np.testing.assert_array_equal(ctx.trace_rays(sid, rays, P.KERNEL_BVH_RESTART), want)
        assert (want[:, 0] == 1).sum() > 2000
        hs = make_scene(P, random_soup(rng, 2500, extent=2.5, size=0.3), lights=[((0.0, 3.0, 1.0), (1, 1, 1), 6.0, 0.7)])
        cube = synthetic_cubemap(rng, 4)
        ref = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), 96, 64, spp=2, bounces=4)
        for stack_lds in (None, "2"):
            if stack_lds:
                monkeypatch.setenv("PTAMD_STACK_LDS", stack_lds)
            acc, rgba = gpu_render(P, ctx, hs, cube, 96, 64, 2, 4, P.KERNEL_BVH_RESTART)
            assert_same(acc, rgba, *ref, f"eight-wide walk, PTAMD_STACK_LDS={stack_lds}")
        monkeypatch.delenv("PTAMD_STACK_LDS")
        ids = (ctx.upload_scene(hs), ctx.upload_cubemap(cube))
        fr = P.FrameRenderer(ctx, *ids, hs.camera_struct(), 96, 64)
        # (the batched launch names the restart kernel itself: PTAMD_DEFAULT_KERNEL, which only stands behind PTAMD_KERNEL_AUTO,
        # may be pinned to a kernel that cannot batch — scripts/gpu_knobtest.sh)
        fr.render(spp=2, bounces=4, kernel=P.KERNEL_BVH_RESTART, batched=True)
        torch.cuda.synchronize()
        assert_same(fr.accum.cpu().numpy(), fr.surface.cpu().numpy(), *ref, "eight-wide walk, batched")


@pytest.mark.parametrize("quantised", ["1", "0"])
def test_quantised_four_wide_walk_on_the_device(P, O, monkeypatch, quantised):
    """(quantised = "0": the same checks on the float nodes, the default.)
    The four-wide tree in 64-byte quantised nodes (Bvh::nodes4q: float origin, per-axis power-of-two scale, 8-bit planes, the
    order of four octants stored and read inverted for the opposite four; PTAMD_WIDE4Q=1): nearest-hit records == brute-force
    oracle (zeros of either sign in the directions), rendered frames == oracle with the stack in LDS and mostly in the global
    slab and with / without the LDS treelet, a batched launch == consecutive launches."""
    import torch
    monkeypatch.setenv("PTAMD_TUNING", "1")
    monkeypatch.setenv("PTAMD_WIDE4Q", quantised)
    rng = np.random.default_rng(47)
    with P.Context(0) as ctx:                       # (knobs are read when a context is created)
        soup = make_scene(P, random_soup(rng, 3000, extent=3.0, size=0.25))
        sid = ctx.upload_scene(soup)
        rays = random_rays(rng, 20000)
        rays[:40, 0] = 0.0
        rays[40:80, 1:3] = -0.0
        rays[80:120, 2] = -0.0                      # (octants 4..7 read the stored order inverted)
        want = O.intersect(O.OracleScene.from_host_scene(soup, P.cubemap_from_color()), rays)
        np.testing.assert_array_equal(ctx.trace_rays(sid, rays, P.KERNEL_BVH_RESTART), want)
        assert (want[:, 0] == 1).sum() > 2000
        hs = make_scene(P, random_soup(rng, 2500, extent=2.5, size=0.3), lights=[((0.0, 3.0, 1.0), (1, 1, 1), 6.0, 0.7)])
        cube = synthetic_cubemap(rng, 4)
        ref = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), 96, 64, spp=2, bounces=4)
        for knobs in ({}, {"PTAMD_STACK_LDS": "2"}, {"PTAMD_TREELET": "0"}, {"PTAMD_TREELET": "37", "PTAMD_STACK_LDS": "3"}):
            for k, v in knobs.items():
                monkeypatch.setenv(k, v)
            with P.Context(0) as c2:
                acc, rgba = gpu_render(P, c2, hs, cube, 96, 64, 2, 4, P.KERNEL_BVH_RESTART)
            assert_same(acc, rgba, *ref, f"quantised four-wide walk, {knobs}")
            for k in knobs:
                monkeypatch.delenv(k)
        ids = (ctx.upload_scene(hs), ctx.upload_cubemap(cube))
        fr = P.FrameRenderer(ctx, *ids, hs.camera_struct(), 96, 64)
        fr.render(spp=2, bounces=4, kernel=P.KERNEL_BVH_RESTART, batched=True)
        torch.cuda.synchronize()
        assert_same(fr.accum.cpu().numpy(), fr.surface.cpu().numpy(), *ref, "quantised four-wide walk, batched")


def test_scene_that_fills_the_lds_share_keeps_its_pools_in_global_memory(P, O, gpu_ctx):
    """An LDS-resident scene of 53-64 KB leaves no room for the restart kernel's path pools next to two scene copies: they
    go to the global slab instead (ptamd_api.cpp); same pixels either way."""
    rng = np.random.default_rng(77)
    soup = random_soup(rng, 400, extent=2.0, size=0.35)
    for n in range(280, 400, 10):                       # the first size whose nodes + triangles land in the window
        hs = make_scene(P, soup[:n], lights=[((0.0, 2.5, 1.0), (1, 1, 1), 5.0, 0.6)])
        info = gpu_ctx.scene_info(gpu_ctx.upload_scene(hs))
        if 53 * 1024 < info["lds_bytes_bvh"] <= 64 * 1024:
            break
    cube = synthetic_cubemap(rng, 2)
    assert 53 * 1024 < info["lds_bytes_bvh"] <= 64 * 1024 and info["n_nodes"] <= 896, info
    ref = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), 96, 64, spp=3, bounces=4)
    for k in (P.KERNEL_BVH_RESTART, P.KERNEL_BVH_PERSISTENT, P.KERNEL_BVH):
        acc, rgba = gpu_render(P, gpu_ctx, hs, cube, 96, 64, 3, 4, k)
        assert_same(acc, rgba, *ref, f"LDS-filling scene kernel {k}")


def test_large_scene_uses_global_memory_variant(P, O, gpu_ctx):
    """A scene whose traversal set exceeds the LDS budget renders through the L2-resident path."""
    rng = np.random.default_rng(33)
    hs = make_scene(P, random_soup(rng, 2500, extent=2.5, size=0.3),
                    lights=[((0.0, 3.0, 1.0), (1, 1, 1), 6.0, 0.7)])
    cube = synthetic_cubemap(rng, 4)
    sid = gpu_ctx.upload_scene(hs)
    info = gpu_ctx.scene_info(sid)
    assert info["lds_bytes_bvh"] > 64 * 1024
    ref = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), 48, 32, spp=2, bounces=3)
    for k in (P.KERNEL_BVH, P.KERNEL_BVH_PERSISTENT, P.KERNEL_BVH_RESTART, P.KERNEL_BVH_BLOCKWISE, P.KERNEL_BVH_SPLIT):
        acc, rgba = gpu_render(P, gpu_ctx, hs, cube, 48, 32, 2, 3, k)
        assert_same(acc, rgba, *ref, f"global-memory BVH kernel {k}")


def test_full_size_properties_1080p_4spp_4bounces(P, O, gpu_ctx, indoor):
    """BASELINE.json configs[1] at full size."""
    import torch
    cube = P.cubemap_for_scene(indoor)
    ids = (gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube))
    W, H, spp, B = 1920, 1080, 4, 4
    bvh_acc, bvh_rgba = gpu_render(P, gpu_ctx, indoor, cube, W, H, spp, B, P.KERNEL_AUTO, ids=ids)
    # (0) persistent waves with lane refill == one-thread-per-pixel walk
    for k in (P.KERNEL_BVH, P.KERNEL_BVH_PERSISTENT, P.KERNEL_BVH_RESTART, P.KERNEL_BVH_BLOCKWISE, P.KERNEL_BVH_SPLIT):
        t_acc, t_rgba = gpu_render(P, gpu_ctx, indoor, cube, W, H, spp, B, k, ids=ids)
        assert_same(bvh_acc, bvh_rgba, t_acc, t_rgba, f"1080p default kernel vs kernel {k}")
    # (1) the BVH walk returns the brute-force loop's result for every ray of every path
    bf_acc, bf_rgba = gpu_render(P, gpu_ctx, indoor, cube, W, H, spp, B, P.KERNEL_BRUTE_FORCE, ids=ids)
    assert_same(bvh_acc, bvh_rgba, bf_acc, bf_rgba, "1080p BVH vs brute force")
    # (2) N spp == sum of N one-spp launches
    parts = [gpu_render(P, gpu_ctx, indoor, cube, W, H, 1, B, P.KERNEL_AUTO, first_frame=k, ids=ids)[0] for k in (1, 2, 3, 4)]
    np.testing.assert_array_equal(bvh_acc, ((parts[0] + parts[1]) + parts[2]) + parts[3])
    # (3) idempotence: rendering again gives the same bits
    again = gpu_render(P, gpu_ctx, indoor, cube, W, H, spp, B, P.KERNEL_AUTO, ids=ids)
    assert_same(*again, bvh_acc, bvh_rgba, "re-render")
    # (4) 8-way row split == full frame
    rg = np.concatenate([gpu_render(P, gpu_ctx, indoor, cube, W, H, spp, B, P.KERNEL_AUTO, rows=r, band_local=True, ids=ids)[1]
                         for r in P.row_bands(H, 8)], axis=0)
    np.testing.assert_array_equal(rg, bvh_rgba)
    # (5) oracle on full-resolution rows (a crop the CPU finishes in seconds)
    rows = (536, 542)
    acc = np.zeros((H, W, 3), np.float32)
    ref_acc, ref_rgba = O.render(O.OracleScene.from_host_scene(indoor, cube), O.camera_from_record(indoor.camera), W, H,
                                 spp=spp, bounces=B, rows=rows, accum=acc)
    np.testing.assert_array_equal(bvh_rgba[rows[0]:rows[1]], ref_rgba[rows[0]:rows[1]])
    np.testing.assert_array_equal(bvh_acc[H - rows[1]:H - rows[0]].view(np.uint32), ref_acc[H - rows[1]:H - rows[0]].view(np.uint32))
    assert (bvh_rgba[..., 3] == 0).all() and bvh_acc.max() <= 4.0 and bvh_acc.min() >= 0.0
    torch.cuda.synchronize()


@pytest.fixture(scope="module")
def atrium(P, tmp_path_factory):
    """BASELINE.json configs[3] asset: the seed-fixed atrium (264 832 one-sided triangles) written as OBJ + MTL + .scene
    and loaded through the product's loader, like the reference's own scenes (scene.cpp:304-358)."""
    from cuda_pathtracer_amd.synthetic import write_atrium
    return P.HostScene.load(write_atrium(str(tmp_path_factory.mktemp("atrium"))))


def test_config4_sponza_class_deep_bvh(P, O, gpu_ctx, atrium):
    """BASELINE.json configs[3]: Sponza-class OBJ (~250k triangles), 1920x1080, 4 spp, 4 bounces — nodes + triangles =
    29 MB, walked from L2.  Full frame: default kernel == tile kernel == one batched launch on every pixel; a 64-row
    band: == the brute-force kernel (the reference algorithm: every face, storage order); two full-width rows: == the
    CPU oracle (brute force over 264 832 faces per ray, seconds); plus the small-frame oracle check of every variant."""
    big = atrium
    assert len(big.faces) == 264832 and len(big.materials) == 6 and len(big.lights) == 6
    assert (big.materials["ior"] == np.float32(1.5)).sum() == 1            # the glass balusters: refraction branch
    cube = P.cubemap_for_scene(big)
    ids = (gpu_ctx.upload_scene(big), gpu_ctx.upload_cubemap(cube))
    info = gpu_ctx.scene_info(ids[0])
    assert info["lds_bytes_bvh"] > 20 * 1024 * 1024 and info["depth"] >= 18
    osc, ocam = O.OracleScene.from_host_scene(big, cube), O.camera_from_record(big.camera)
    ref = O.render(osc, ocam, 64, 36, spp=1, bounces=3)
    for k in (P.KERNEL_AUTO, P.KERNEL_BVH, P.KERNEL_BVH_PERSISTENT, P.KERNEL_BVH_RESTART, P.KERNEL_BVH_BLOCKWISE, P.KERNEL_BVH_SPLIT):
        acc, rgba = gpu_render(P, gpu_ctx, big, cube, 64, 36, 1, 3, k, ids=ids)
        assert_same(acc, rgba, *ref, f"sponza-class kernel {k}")
    W, H, spp, B = 1920, 1080, 4, 4
    a0, r0 = gpu_render(P, gpu_ctx, big, cube, W, H, spp, B, P.KERNEL_AUTO, ids=ids)
    a1, r1 = gpu_render(P, gpu_ctx, big, cube, W, H, spp, B, P.KERNEL_BVH, ids=ids)
    assert_same(a0, r0, a1, r1, "sponza-class 1080p 4 spp default vs tile")
    # the batched launch (what bench.py times) gives the same frame
    import torch
    if batched_ok():
        fr = P.FrameRenderer(gpu_ctx, *ids, big.camera_struct(), W, H)
        fr.render(spp=spp, bounces=B, batched=True)
        torch.cuda.synchronize()
        assert_same(fr.accum.cpu().numpy(), fr.surface.cpu().numpy(), a0, r0, "sponza-class 1080p 4 spp batched vs sequential")
    # brute-force kernel on a band (full-frame buffers, band launch)
    band = (500, 564)
    ab, rb = gpu_render(P, gpu_ctx, big, cube, W, H, spp, B, P.KERNEL_BRUTE_FORCE, rows=band, ids=ids)
    np.testing.assert_array_equal(rb[band[0]:band[1]], r0[band[0]:band[1]])
    np.testing.assert_array_equal(ab[H - band[1]:H - band[0]].view(np.uint32), a0[H - band[1]:H - band[0]].view(np.uint32))
    # oracle on two full-width rows inside that band
    rows = (530, 532)
    ref_acc, ref_rgba = O.render(osc, ocam, W, H, spp=spp, bounces=B, rows=rows, accum=np.zeros((H, W, 3), np.float32))
    np.testing.assert_array_equal(r0[rows[0]:rows[1]], ref_rgba[rows[0]:rows[1]])
    np.testing.assert_array_equal(a0[H - rows[1]:H - rows[0]].view(np.uint32), ref_acc[H - rows[1]:H - rows[0]].view(np.uint32))
    assert (r0[..., :3] > 0).mean() > 0.4                                  # a lit interior, not a black frame


def test_config4_second_case_tessellated_indoor(P, O, gpu_ctx, indoor):
    """Second deep-BVH case (round 1's stand-in, kept): indoor.obj tessellated 24x24 per face = 256 896 coplanar
    triangles.  Small frame vs the oracle; full frame default kernel == tile kernel; close to the 446-face image."""
    big = P.tessellate(indoor, 24)
    assert len(big.faces) == 256896
    cube = P.cubemap_for_scene(big)
    ids = (gpu_ctx.upload_scene(big), gpu_ctx.upload_cubemap(cube))
    ref = O.render(O.OracleScene.from_host_scene(big, cube), O.camera_from_record(big.camera), 64, 36, spp=1, bounces=3)
    for k in (P.KERNEL_AUTO, P.KERNEL_BVH):
        acc, rgba = gpu_render(P, gpu_ctx, big, cube, 64, 36, 1, 3, k, ids=ids)
        assert_same(acc, rgba, *ref, f"tessellated indoor kernel {k}")
    W, H, spp, B = 1920, 1080, 4, 4
    a0, r0 = gpu_render(P, gpu_ctx, big, cube, W, H, spp, B, P.KERNEL_AUTO, ids=ids)
    a1, r1 = gpu_render(P, gpu_ctx, big, cube, W, H, spp, B, P.KERNEL_BVH, ids=ids)
    assert_same(a0, r0, a1, r1, "tessellated indoor 1080p 4 spp default vs tile")
    b0, q0 = gpu_render(P, gpu_ctx, indoor, cube, W, H, spp, B, P.KERNEL_AUTO)
    assert np.abs(a0.mean() - b0.mean()) < 0.04


def test_config5_4k_16spp_8bounces_dof(P, O, gpu_ctx):
    """BASELINE.json configs[4] at full size: indoor, 3840x2160, 16 spp, 8 bounces, aperture 0.113 (the crate_land
    value, crate_land.scene:4).  The default path (one batched launch; at >= 5 bounces the persistent kernel refills
    lanes mid-path) == 16 consecutive launches of the tile BVH kernel == 16 launches of the brute-force kernel on every
    pixel, and == the CPU oracle on four full-width rows."""
    needs_batched_default()
    import torch
    hs = P.HostScene.load(os.path.join(ASSETS, "indoor.scene"))
    hs.camera["aperture"] = np.float32(0.113)
    cube = P.cubemap_for_scene(hs)
    ids = (gpu_ctx.upload_scene(hs), gpu_ctx.upload_cubemap(cube))
    W, H, spp, B = 3840, 2160, 16, 8
    fr = P.FrameRenderer(gpu_ctx, *ids, hs.camera_struct(), W, H)
    fr.render(spp=spp, bounces=B, batched=True)
    torch.cuda.synchronize()
    acc, rgba = fr.accum.cpu().numpy(), fr.surface.cpu().numpy()
    del fr
    for k, name in ((P.KERNEL_BVH, "tile BVH"), (P.KERNEL_BRUTE_FORCE, "brute force")):
        a, r = gpu_render(P, gpu_ctx, hs, cube, W, H, spp, B, k, ids=ids)
        assert_same(acc, rgba, a, r, f"4K 16 spp 8 bounces: default batched vs {name} sequential")
        del a, r
    rows = (1078, 1082)
    ref_acc, ref_rgba = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), W, H,
                                 spp=spp, bounces=B, rows=rows, accum=np.zeros((H, W, 3), np.float32))
    np.testing.assert_array_equal(rgba[rows[0]:rows[1]], ref_rgba[rows[0]:rows[1]])
    np.testing.assert_array_equal(acc[H - rows[1]:H - rows[0]].view(np.uint32), ref_acc[H - rows[1]:H - rows[0]].view(np.uint32))
    assert (rgba[..., 3] == 0).all() and acc.max() <= 16.0 and acc.min() >= 0.0
    # depth of field is on: neighbouring pixels at the focus-free distance differ from the pinhole-ish default render
    hs2 = P.HostScene.load(os.path.join(ASSETS, "indoor.scene"))
    assert float(hs2.camera["aperture"]) < 0.05


def test_far_camera_keeps_bvh_exact(P, O, gpu_ctx):
    """the slab test's rounding grows with the ray origin.  Thin unit-scale geometry seen from 2 000 units
    (inside the margin's reach: BVH walk) and from 60 000 units (beyond it: the launcher switches to the exhaustive
    face loop) must both equal the brute-force kernel and the oracle."""
    rng = np.random.default_rng(5)
    n = 200
    c = rng.uniform(-1.0, 1.0, size=(n, 1, 3))
    tris = (c + rng.normal(scale=0.08, size=(n, 3, 3)) * np.float32([1.0, 1.0, 0.002])).astype(np.float32)   # thin in z
    cube = synthetic_cubemap(rng, 2)
    # (margin_floor = 1e-3 + extent * 2^-20 here: the launcher walks the tree up to (|camera| + extent) * 2^-21 <= margin_floor, i.e.
    # about 2 097 units — 2 090 is just inside, 2 300 just outside, where every face is tested: ADVICE r3 on the two-step slab form)
    for dist, fov in ((2.0e3, 0.002), (2.09e3, 0.002), (2.3e3, 0.002), (6.0e4, 0.00007)):
        hs = make_scene(P, tris, lights=[((0.0, 0.5, 2.0), (1.0, 0.9, 0.8), 4.0, 0.3)],
                        camera=dict(position=(0.0, 0.0, dist), dir=(0.0, 0.0, -1.0), fov_x=fov, aperture=0.0, focus_dist=dist))
        ref = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), 64, 48, spp=2, bounces=3)
        assert (ref[0] > 0).any()
        for kernel in KERNELS:
            acc, rgba = gpu_render(P, gpu_ctx, hs, cube, 64, 48, 2, 3, kid(P, kernel))
            assert_same(acc, rgba, *ref, f"camera at {dist:g}/{kernel}")
        # batched launches take the same route
        import torch
        if batched_ok():
            sid, cid = gpu_ctx.upload_scene(hs), gpu_ctx.upload_cubemap(cube)
            fr = P.FrameRenderer(gpu_ctx, sid, cid, hs.camera_struct(), 64, 48)
            fr.render(spp=2, bounces=3, batched=True)
            torch.cuda.synchronize()
            assert_same(fr.accum.cpu().numpy(), fr.surface.cpu().numpy(), *ref, f"camera at {dist:g}/batched")


def test_far_camera_with_interleaved_bands_and_batches(P, O, gpu_ctx):
    """a camera beyond the box margins' reach used to send interleaved-band launches to the one-thread-per-pixel
    kernel, which renders the WHOLE frame into band-local buffers.  Such launches now stay in the restart kernel (its
    every-triangle instantiation): interleaved bands, batched or not, put back in frame order equal the full-frame render
    and the oracle."""
    if os.environ.get("PTAMD_DEFAULT_KERNEL", "6") != "6":
        pytest.skip("interleaved bands need the restart kernel behind PTAMD_KERNEL_AUTO")
    import torch
    rng = np.random.default_rng(5)
    n = 200
    c = rng.uniform(-1.0, 1.0, size=(n, 1, 3))
    tris = (c + rng.normal(scale=0.08, size=(n, 3, 3)) * np.float32([1.0, 1.0, 0.002])).astype(np.float32)
    cube = synthetic_cubemap(rng, 2)
    dist, fov = 1.0e5, 0.00004
    hs = make_scene(P, tris, lights=[((0.0, 0.5, 2.0), (1.0, 0.9, 0.8), 4.0, 0.3)],
                    camera=dict(position=(0.0, 0.0, dist), dir=(0.0, 0.0, -1.0), fov_x=fov, aperture=0.0, focus_dist=dist))
    W, H, spp, B = 72, 50, 2, 3
    ref_acc, ref_rgba = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), W, H, spp=spp, bounces=B)
    assert (ref_acc > 0).any()
    ids = (gpu_ctx.upload_scene(hs), gpu_ctx.upload_cubemap(cube))
    for kernel in (P.KERNEL_AUTO, P.KERNEL_BVH_RESTART):
        acc, rgba = gpu_render(P, gpu_ctx, hs, cube, W, H, spp, B, kernel, ids=ids)
        assert_same(acc, rgba, ref_acc, ref_rgba, f"far camera, kernel {kernel}")
    world, rank_rows = 4, 8
    for batched in (False, True):
        rgba = np.zeros_like(ref_rgba)
        acc = np.zeros_like(ref_acc)
        for rank in range(world):
            fr = P.FrameRenderer(gpu_ctx, *ids, hs.camera_struct(), W, H, interleave=(world, rank, rank_rows))
            fr.render(spp=spp, bounces=B, batched=batched and batched_ok())
            torch.cuda.synchronize()
            s, a = fr.surface.cpu().numpy(), fr.accum.cpu().numpy()
            local = 0
            for b, e in P.interleaved_bands(H, world, rank, rank_rows):
                rgba[b:e] = s[local:local + (e - b)]
                acc[H - e:H - b] = a[s.shape[0] - (local + (e - b)):s.shape[0] - local]
                local += e - b
        assert_same(acc, rgba, ref_acc, ref_rgba, f"far camera, interleaved bands, batched={batched}")


def test_back_to_back_launches_on_one_stream(P, gpu_ctx, indoor):
    """A host that issues launches without waiting for them (the reference's render loop, gpu_processor.cpp:365-386) has
    consecutive launches of a stream pipelined by the library: half-GPU launches whose megakernels run on internal
    streams, resolve passes in order on the caller's.  Same accumulator and surface, bit for bit, as when the host waits
    for every launch — single-frame and batched launches, two renderers taking turns on one stream, and the default
    stream as well as a created one."""
    import torch
    cube = P.cubemap_for_scene(indoor)
    ids = (gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube))
    W, H, B = 1920, 1080, 4

    def run(stream, wait, batched):
        frs = [P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H) for _ in range(2)]
        torch.cuda.synchronize()   # (the buffers were zeroed on the default stream)
        frame = 1
        for step in range(6):
            fr = frs[step % 2]
            n = 3 if batched else 1
            fr.render(spp=n, bounces=B, kernel=P.KERNEL_AUTO, stream=stream, first_frame=frame, batched=batched and batched_ok())
            frame += n
            if wait:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return [(f.accum.cpu().numpy(), f.surface.cpu().numpy()) for f in frs]

    for batched in (False, True):
        want = run(None, True, batched)
        for stream in (None, torch.cuda.Stream()):
            got = run(stream, False, batched)
            for (a, s), (wa, ws) in zip(got, want):
                assert_same(a, s, wa, ws, f"back to back, batched={batched}, stream={'default' if stream is None else 'created'}")


def test_crate_land_with_real_textures_and_cubemap(P, O, gpu_ctx):
    """The reference scene that exercises sampleTexture on 1024^2 RGBA textures, normal mapping and
    a bilinear 1024^2 cubemap, decoded by the built-in decoder (== the reference's stb_image, test_ref_thirdparty)."""
    hs = P.HostScene.load(os.path.join(ASSETS, "crate_land.scene"))
    cube = P.cubemap_for_scene(hs, asset_folder=ASSETS)
    assert hs.unloaded_textures == []
    assert cube.shape[1] == 1024 and len(hs.texels) > 14_000_000
    ref = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), 96, 54, spp=2, bounces=4)
    assert O.last_stats()["nmap_hits"] > 100
    for kernel in KERNELS:
        acc, rgba = gpu_render(P, gpu_ctx, hs, cube, 96, 54, 2, 4, kid(P, kernel))
        assert_same(acc, rgba, *ref, f"crate_land textured/{kernel}")


@pytest.mark.parametrize("name", ["indoor_textured", "crate_land"])
def test_textured_scenes_at_full_size(P, O, gpu_ctx, name):
    """The dependent texel gathers of sampleTexture (intersection.cuh:20-65,216-242) at BASELINE.json's frame size:
    indoor.obj with the textures its MTL names (parquet / concrete / wooden_planck albedo + normal maps: backslash
    paths normalised) and crate_land.obj (1024^2 RGBA + normal maps, bilinear 1024^2 cubemap), 1920x1080,
    4 spp, 4 bounces: default kernel (one batched launch and four consecutive launches) == one-thread-per-pixel kernel
    on every pixel, plus the oracle on a crop of full-width rows."""
    import torch
    if name == "indoor_textured":
        hs = P.HostScene.load(os.path.join(ASSETS, "indoor.scene"), normalise_backslashes=True)
        assert hs.unloaded_textures == [] and len(hs.texels) > 16_000_000
    else:
        hs = P.HostScene.load(os.path.join(ASSETS, "crate_land.scene"))
        assert hs.unloaded_textures == []
    cube = P.cubemap_for_scene(hs, asset_folder=ASSETS)
    ids = (gpu_ctx.upload_scene(hs), gpu_ctx.upload_cubemap(cube))
    W, H, spp, B = 1920, 1080, 4, 4
    acc, rgba = gpu_render(P, gpu_ctx, hs, cube, W, H, spp, B, P.KERNEL_AUTO, ids=ids)
    t_acc, t_rgba = gpu_render(P, gpu_ctx, hs, cube, W, H, spp, B, P.KERNEL_BVH, ids=ids)
    assert_same(acc, rgba, t_acc, t_rgba, f"{name} 1080p default kernel vs tile kernel")
    if batched_ok():
        fr = P.FrameRenderer(gpu_ctx, ids[0], ids[1], hs.camera_struct(), W, H)
        fr.render(spp=spp, bounces=B, batched=True)
        torch.cuda.synchronize()
        assert_same(fr.accum.cpu().numpy(), fr.surface.cpu().numpy(), acc, rgba, f"{name} 1080p batched")
    rows = (600, 604)
    buf = np.zeros((H, W, 3), np.float32)
    ref_acc, ref_rgba = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), W, H,
                                 spp=spp, bounces=B, rows=rows, accum=buf)
    st = O.last_stats()
    assert st["mesh_hits"] > 1000 and st["nmap_hits"] > 100
    np.testing.assert_array_equal(rgba[rows[0]:rows[1]], ref_rgba[rows[0]:rows[1]])
    np.testing.assert_array_equal(acc[H - rows[1]:H - rows[0]].view(np.uint32), ref_acc[H - rows[1]:H - rows[0]].view(np.uint32))


def test_random_scenes_fuzz(P, O, gpu_ctx):
    """Seeded random scenes (triangle soups with textures, normal maps, refractive materials, 0-4 lights, random camera
    lens, ragged frame sizes, 1-7 bounces, every post-process) through the default kernel and the tile kernel."""
    for seed in range(14):
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.integers(20, 400))
        tris = random_soup(rng, n, extent=float(rng.uniform(0.8, 3.0)), size=float(rng.uniform(0.2, 1.2)))
        uvs = rng.uniform(-1.0, 2.0, size=(n, 3, 2)).astype(np.float32)
        textures = [rng.uniform(0.05, 0.95, size=(int(rng.integers(1, 9)), int(rng.integers(1, 9)), 4)).astype(np.float32),
                    rng.uniform(0.0, 1.0, size=(int(rng.integers(1, 6)), int(rng.integers(1, 6)), 3)).astype(np.float32),
                    np.float32([[[0.9, 0.8, 0.7, float(rng.uniform(0, 1))]]])]
        materials = [(0, 1, 1.0), (0, -1, 1.0), (2, -1, float(rng.uniform(1.1, 1.8))), (2, 1, 1.0)]
        lights = [(tuple(rng.uniform(-2, 2, 3)), tuple(rng.uniform(0.2, 1, 3)), float(rng.uniform(1, 6)), float(rng.uniform(0.1, 0.8)))
                  for _ in range(int(rng.integers(0, 5)))]
        hs = make_scene(P, tris, uvs=uvs, material_ids=rng.integers(0, 4, size=n), materials=materials, textures=textures, lights=lights,
                        mesh_sizes=None)
        hs.camera["aperture"] = np.float32(rng.uniform(0.0, 0.2))
        hs.camera["focus_dist"] = np.float32(rng.uniform(0.5, 4.0))
        cube = synthetic_cubemap(rng, int(rng.choice([1, 2, 4])))
        W, H = int(rng.integers(1, 70)), int(rng.integers(1, 50))
        spp, B, post = int(rng.integers(1, 4)), int(rng.integers(1, 8)), int(rng.integers(0, 4))
        ref = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), W, H, spp=spp, bounces=B, post_id=post)
        for kernel in ("persistent", "restart", "bvh"):
            acc, rgba = gpu_render(P, gpu_ctx, hs, cube, W, H, spp, B, kid(P, kernel), post_id=post)
            assert_same(acc, rgba, *ref, f"fuzz seed {seed} {W}x{H} spp{spp} B{B} post{post}/{kernel}")
        # the same frames as ONE batched launch on a third of the GPU (what a host with frames in flight issues)
        import torch
        if batched_ok():
            sid, cid = gpu_ctx.upload_scene(hs), gpu_ctx.upload_cubemap(cube)
            fr = P.FrameRenderer(gpu_ctx, sid, cid, hs.camera_struct(), W, H, machine_share=3)
            fr.render(spp=spp, bounces=B, post_id=post, batched=True)
            torch.cuda.synchronize()
            assert_same(fr.accum.cpu().numpy(), fr.surface.cpu().numpy(), *ref, f"fuzz seed {seed} batched, machine_share 3")


def test_scene_with_huge_coordinates(P, O, gpu_ctx):
    """Coordinates around 1e5 (the reference's MAX_DIST): box margins scale with the coordinate (1e-6 relative), the 16-bit
    LDS link addresses do not depend on scene size — same pixels on every variant."""
    rng = np.random.default_rng(33)
    tris = random_soup(rng, 60, extent=1.2, size=0.9) * np.float32(1.0e5)
    lights = [((0.0, 0.5e5, 1.0e5), (1.0, 0.9, 0.8), 4.0, 0.8e5)]
    hs = make_scene(P, tris, lights=lights)
    hs.camera["position"] *= np.float32(1.0e5)
    hs.camera["focus_dist"] *= np.float32(1.0e5)
    cube = synthetic_cubemap(rng, 2)
    ref = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), 48, 32, spp=2, bounces=4)
    assert (ref[0] > 0).any()
    for kernel in KERNELS:
        acc, rgba = gpu_render(P, gpu_ctx, hs, cube, 48, 32, 2, 4, kid(P, kernel))
        assert_same(acc, rgba, *ref, f"far scene/{kernel}")


def test_gamma_table_matches_the_pow_sequence(P, gpu_ctx):
    """The resolve pass reads the surface byte of a tonemapped channel off a 256-step table (pt_kernels.hip: gamma_byte) instead
    of evaluating pt_powf per channel.  On the device, for EVERY positive binary32 value below the table's last step (about 2^30
    of them) plus samples of the values it hands back to the pow sequence (large, negative, NaN): table form == sequence."""
    checked, bad = gpu_ctx.gamma_table_selftest()
    if os.environ.get("PTAMD_GAMMA_TABLE", "1") == "0":
        assert checked == 0 and bad == 0
        return
    assert checked > 1_000_000_000, checked
    assert bad == 0, f"{bad} of {checked} values differ"


def test_one_texel_cubemaps(P, O, gpu_ctx):
    """1x1 cubemaps: six different texels (the face choice matters), six identical ones (the launcher's one-colour shortcut:
    env_lookup returns the colour without a lookup), identical rgb with different alpha (still one colour), and rgb that differs
    only in the sign of a zero (NOT one colour: bit patterns decide) — static and preview launches, every kernel variant."""
    rng = np.random.default_rng(91)
    lights = [((0.0, 0.5, 1.0), (1.0, 0.9, 0.8), 4.0, 0.8)]
    hs = make_scene(P, random_soup(rng, 30, extent=1.5, size=0.7), lights=lights)      # sparse: many rays escape
    six = synthetic_cubemap(rng, 1)
    same = np.repeat(six[:1], 6, axis=0).copy()
    same_alpha = same.copy(); same_alpha[:, 0, 0, 3] = np.arange(6, dtype=np.float32)
    zero_sign = same.copy(); zero_sign[:, 0, 0, 1] = 0.0; zero_sign[3, 0, 0, 1] = -0.0
    for name, cube in (("six texels", six), ("one colour", same), ("one colour, alpha differs", same_alpha), ("zero of either sign", zero_sign)):
        for moved in (False, True):
            ref = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), 48, 32, spp=2, bounces=4, moved=moved)
            assert (ref[0] > 0).any()
            for kernel in KERNELS:
                acc, rgba = gpu_render(P, gpu_ctx, hs, cube, 48, 32, 2, 4, kid(P, kernel), moved=moved)
                assert_same(acc, rgba, *ref, f"{name}/moved={moved}/{kernel}")


def test_scenes_outside_the_short_reciprocal_range(P, O, gpu_ctx):
    """The restart kernel divides by the Moller-Trumbore determinant with a 7-instruction exact reciprocal only where the
    launcher can bound the determinant (all vertices finite and <= 1e8, ptamd_api.cpp: small_det).  One far vertex (3e9), an
    infinite one (either sign: infinite coordinates stay out of the boxes, a face with one can never be hit), a NaN one and one
    near the end of the float range switch it off: every variant still renders the oracle's pixels."""
    rng = np.random.default_rng(77)
    cube = synthetic_cubemap(rng, 2)
    lights = [((0.0, 0.5, 1.0), (1.0, 0.9, 0.8), 4.0, 0.8)]
    base = random_soup(rng, 60, extent=1.2, size=0.9)
    for name, bad in (("far vertex", 3.0e9), ("infinite vertex", np.inf), ("nan vertex", np.nan),
                      ("minus infinite vertex", -np.inf), ("vertex at -3e38", -3.0e38)):
        tris = base.copy()
        tris[7, 1, 0] = np.float32(bad)          # one coordinate of one face
        if name == "far vertex":
            tris[9] = tris[9] * np.float32(1.0e9)   # and a whole face out there: huge determinants for every ray
        hs = make_scene(P, tris, lights=lights)
        ref = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), 48, 32, spp=2, bounces=4)
        assert (ref[0] > 0).any()
        for kernel in KERNELS:
            acc, rgba = gpu_render(P, gpu_ctx, hs, cube, 48, 32, 2, 4, kid(P, kernel))
            assert_same(acc, rgba, *ref, f"{name}/{kernel}")


def test_stats_are_consistent(P, O, gpu_ctx, indoor):
    """Instrumented launch: ray/mesh-hit counts equal the oracle's, BVH tests far fewer triangles."""
    import torch
    cube = P.cubemap_for_scene(indoor)
    ids = (gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube))
    W, H = 96, 64
    fr = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H)
    l = gpu_ctx.make_launch(fr.surface, fr.accum, *ids, indoor.camera_struct(), W, H, frame_nb=1, bounces=4, kernel=P.KERNEL_BVH)
    s_bvh = gpu_ctx.raytrace_stats(l)
    fr.reset()
    for k in (P.KERNEL_BVH_PERSISTENT, P.KERNEL_BVH_RESTART, P.KERNEL_BVH_BLOCKWISE, P.KERNEL_BVH_SPLIT):
        fr.reset()
        l.kernel = k
        s_k = gpu_ctx.raytrace_stats(l)
        for key in ("rays", "nodes_visited", "tris_tested", "mesh_hits", "nmap_hits", "samples"):
            assert s_k[key] == s_bvh[key]                  # scheduling changes, the work does not
    fr.reset()
    l.kernel = P.KERNEL_BRUTE_FORCE
    s_bf = gpu_ctx.raytrace_stats(l)
    O.render(O.OracleScene.from_host_scene(indoor, cube), O.camera_from_record(indoor.camera), W, H, spp=1, bounces=4)
    st = O.last_stats()
    assert s_bvh["samples"] == s_bf["samples"] == W * H
    assert s_bvh["rays"] == s_bf["rays"] == st["calls"]
    assert s_bvh["mesh_hits"] == s_bf["mesh_hits"] == st["mesh_hits"]
    assert s_bf["tris_tested"] >= s_bvh["tris_tested"] * 20 and s_bvh["nodes_visited"] > 0


def test_launch_timeline_stamps(P, gpu_ctx, indoor):
    """ptamd_set_timeline: the time-stamp instantiation of the default kernel renders the same pixels and records, per wave,
    entry <= scene staged <= (no ticket left) <= exit; switched off, nothing is recorded."""
    import torch
    cube = P.cubemap_for_scene(indoor)
    ids = (gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube))
    W, H, B = 640, 360, 4
    want = gpu_render(P, gpu_ctx, indoor, cube, W, H, 1, B, P.KERNEL_BVH_RESTART, ids=ids)
    n_waves = 256 * 24
    gpu_ctx.set_timeline(n_waves)
    try:
        got = gpu_render(P, gpu_ctx, indoor, cube, W, H, 1, B, P.KERNEL_BVH_RESTART, ids=ids)
        tl, khz = gpu_ctx.read_timeline(n_waves)
    finally:
        gpu_ctx.set_timeline(0)
    assert_same(*got, *want, "time-stamp instantiation")
    live = tl[:, 3] != 0
    assert khz > 0 and 12 <= live.sum() <= n_waves
    t = tl[live].astype(np.int64)
    assert (t[:, 0] <= t[:, 1]).all() and (t[:, 1] <= t[:, 3]).all()
    dry = t[:, 2] != 0
    assert dry.any() and (t[dry, 1] <= t[dry, 2]).all() and (t[dry, 2] <= t[dry, 3]).all()
    assert (t[:, 3].max() - t[:, 0].min()) / khz < 50.0          # the launch lasted less than 50 ms
    # off again: launches record nothing (the buffer is gone; reading it is an error)
    gpu_render(P, gpu_ctx, indoor, cube, W, H, 1, B, P.KERNEL_BVH_RESTART, ids=ids)
    with pytest.raises(P.PtamdError):
        gpu_ctx.read_timeline(16)
    torch.cuda.synchronize()


def test_error_behaviour(P, gpu_ctx, indoor):
    """Bad arguments come back as status codes with a message; nothing is launched."""
    import torch
    cube = P.cubemap_for_scene(indoor)
    ids = (gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube))
    fr = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), 32, 32)
    def launch(**kw):
        base = dict(frame_nb=1, bounces=3)
        base.update(kw)
        return gpu_ctx.make_launch(fr.surface, fr.accum, ids[0], ids[1], indoor.camera_struct(), 32, 32, **base)
    for bad in (dict(frame_nb=0), dict(bounces=0), dict(post_id=4), dict(rows=(5, 40)), dict(rows=(9, 3)), dict(kernel=9)):
        with pytest.raises(P.PtamdError) as e:
            gpu_ctx.raytrace_ex(launch(**bad))
        assert e.value.status == P.native.PTAMD_ERR_ARG
    l = launch()
    l.scene_id = 999
    with pytest.raises(P.PtamdError):
        gpu_ctx.raytrace_ex(l)
    with pytest.raises(ValueError):
        gpu_ctx.make_launch(torch.zeros(4), fr.accum, *ids, indoor.camera_struct(), 32, 32, frame_nb=1)   # CPU tensor
    bad_scene = P.HostScene(indoor.faces.copy(), indoor.mesh_sizes, indoor.materials, indoor.lights, indoor.textures, indoor.texels)
    bad_scene.faces["material_id"][3] = 77
    with pytest.raises(P.PtamdError):
        gpu_ctx.upload_scene(bad_scene)
    # empty row band is a no-op
    gpu_ctx.raytrace_ex(launch(rows=(7, 7)))
    torch.cuda.synchronize()


def test_contracted_kernel_stays_inside_the_stated_tolerance(P, gpu_ctx, indoor):
    """PTAMD_KERNEL_BVH_RESTART_FMA — the opt-in instantiation with floating-point contraction allowed, what the reference's own
    nvcc build permits (cuda_opengl/CMakeLists.txt:20-22) — is NOT bit-exact and is not meant to be.  It is held to the measured
    drift between faithful builds of the integrator (tests/test_tolerance_study.py): on the headline scene a
    pixel is a function of the sequence of surfaces its path hits, so all but <= 1e-4 of the pixels must be IDENTICAL to the exact
    kernel's and the mean image must agree to 2e-6; on a scene whose radiance is continuous in the hit point (crate_land) >= 99 % of
    the pixels must lie within 1e-4 and >= 99.85 % within +-1 LSB, the mean image within 1e-5.  Batched and single-frame launches of
    the contracted kernel must agree with each other bit for bit (it is deterministic; only its rounding differs)."""
    import torch
    from test_tolerance_study import (INDOOR_MAX_MEAN_IMAGE_DELTA, INDOOR_MAX_SHARE_BEYOND_1E4, INDOOR_MAX_SHARE_BEYOND_1LSB,
                                      TEXTURED_MAX_MEAN_IMAGE_DELTA, TEXTURED_MAX_SHARE_BEYOND_1E4, TEXTURED_MAX_SHARE_BEYOND_1LSB)
    needs_batched_default()

    def both(hs, cube, W, H, spp, bounces):
        sid, cid = gpu_ctx.upload_scene(hs), gpu_ctx.upload_cubemap(cube)
        out = []
        for kernel in (P.KERNEL_BVH_RESTART, P.KERNEL_BVH_RESTART_FMA):
            fr = P.FrameRenderer(gpu_ctx, sid, cid, hs.camera_struct(), W, H)
            fr.render(spp=spp, bounces=bounces, kernel=kernel, batched=True)
            torch.cuda.synchronize()
            out.append((fr.accum.cpu().numpy().astype(np.float64) / spp, fr.surface.cpu().numpy()))
        # the contracted kernel once more, one launch per frame: must equal its own batched launch bit for bit
        fr = P.FrameRenderer(gpu_ctx, sid, cid, hs.camera_struct(), W, H)
        fr.render(spp=spp, bounces=bounces, kernel=P.KERNEL_BVH_RESTART_FMA, batched=False)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(fr.surface.cpu().numpy(), out[1][1])
        np.testing.assert_array_equal((fr.accum.cpu().numpy().astype(np.float64) / spp), out[1][0])
        return out

    def drift(exact, fma):
        d = np.nan_to_num(np.abs(fma[0] - exact[0]), nan=1.0).max(axis=2)
        lsb = np.abs(fma[1][..., :3].astype(np.int32) - exact[1][..., :3].astype(np.int32)).max(axis=2)
        mean_delta = np.abs(fma[0].mean(axis=(0, 1)) - exact[0].mean(axis=(0, 1))).max()
        return float((d > 1e-4).mean()), float((lsb > 1).mean()), float(mean_delta), float((d == 0).mean())

    e, f = both(indoor, P.cubemap_for_scene(indoor), 1920, 1080, 4, 4)
    beyond, lsb, mean_delta, identical = drift(e, f)
    assert beyond <= INDOOR_MAX_SHARE_BEYOND_1E4 and lsb <= INDOOR_MAX_SHARE_BEYOND_1LSB and mean_delta <= INDOOR_MAX_MEAN_IMAGE_DELTA, \
        (beyond, lsb, mean_delta)
    assert identical >= 1.0 - INDOOR_MAX_SHARE_BEYOND_1E4, identical
    assert identical < 1.0, "the contracted kernel rendered the exact kernel's image: is it the contracted code object?"

    hs2 = P.HostScene.load(os.path.join(ASSETS, "crate_land.scene"))
    cube2 = P.cubemap_for_scene(hs2, asset_folder=ASSETS)
    e, f = both(hs2, cube2, 480, 270, 4, 4)
    beyond, lsb, mean_delta, identical = drift(e, f)
    assert beyond <= TEXTURED_MAX_SHARE_BEYOND_1E4 and lsb <= TEXTURED_MAX_SHARE_BEYOND_1LSB and mean_delta <= TEXTURED_MAX_MEAN_IMAGE_DELTA, \
        (beyond, lsb, mean_delta)
    assert identical < 1.0


def test_captured_launch_pins_its_slab_and_the_library_enforces_it(P, gpu_ctx, indoor):
    """ptamd.h "What a captured launch pins": after a capture on a stream, a LARGER launch on that stream is refused
    (PTAMD_ERR_LIMIT) instead of reallocating the slab under the graph; replays stay bit-identical; the same or a smaller
    configuration is fine; ptamd_release_captured lifts the pin."""
    needs_batched_default()
    import torch
    W, H, spp = 640, 360, 4
    cube = P.cubemap_for_scene(indoor)
    ids = (gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube))
    eager = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H)
    eager.render(spp=spp, bounces=4, batched=True, reset=True)
    torch.cuda.synchronize()
    want = (eager.accum.cpu().numpy(), eager.surface.cpu().numpy())
    fr = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H)
    big = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), 2 * W, 2 * H)
    small = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W // 2, H // 2)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fr.render(spp=spp, bounces=4, batched=True, reset=True, stream=side)     # sizes this stream's slab
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fr.render(spp=spp, bounces=4, batched=True, reset=True, stream=torch.cuda.current_stream())
    try:
        with torch.cuda.stream(side):
            # (a) a larger launch on the capture's stream would reallocate the slab the graph writes: refused
            with pytest.raises(P.PtamdError) as err:
                big.render(spp=spp, bounces=4, batched=True, reset=True, stream=side)
            assert err.value.status == P.native.PTAMD_ERR_LIMIT and "captured" in str(err.value)
            # (b) a smaller one is fine (stream order protects the slab)
            small.render(spp=spp, bounces=4, batched=True, reset=True, stream=side)
        torch.cuda.synchronize()
        for rep in range(2):
            fr.accum.fill_(7.0)
            fr.surface.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert_same(fr.accum.cpu().numpy(), fr.surface.cpu().numpy(), *want, f"graph replay {rep} after a refused launch")
    finally:
        del g
        torch.cuda.synchronize()
        gpu_ctx.release_captured(side)
    # (c) the pin is lifted: the larger launch goes through and renders what it should
    with torch.cuda.stream(side):
        big.render(spp=spp, bounces=4, batched=True, reset=True, stream=side)
    torch.cuda.synchronize()
    ref = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), 2 * W, 2 * H)
    ref.render(spp=spp, bounces=4, batched=False, reset=True)
    torch.cuda.synchronize()
    assert_same(big.accum.cpu().numpy(), big.surface.cpu().numpy(), ref.accum.cpu().numpy(), ref.surface.cpu().numpy(), "after release")


def test_long_batches_are_issued_four_frames_at_a_time(P, gpu_ctx, indoor):
    """frame_count = N is N consecutive launches by contract; the library issues batches longer than four frames as consecutive
    launches of four, so that a stream's sample slab does not grow with N: same bits as one launch per frame, and
    no more device memory for 13 frames than for 4."""
    needs_batched_default()
    import torch
    W, H = 480, 270
    cube = P.cubemap_for_scene(indoor)
    ids = (gpu_ctx.upload_scene(indoor), gpu_ctx.upload_cubemap(cube))
    st = torch.cuda.Stream()
    a = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H)

    def batch(n):   # one call with frame_count = n, never pipelined by the library (so that the slabs in play do not depend on timing)
        l = gpu_ctx.make_launch(a.surface, a.accum, *ids, indoor.camera_struct(), W, H, frame_nb=1, bounces=4, stream=st,
                                kernel=P.KERNEL_AUTO, frame_count=n, reset_accumulation=True, no_pipelining=True)
        with torch.cuda.stream(st):
            gpu_ctx.raytrace_ex(l)
        torch.cuda.synchronize()

    batch(4)
    free_after_4 = torch.cuda.mem_get_info()[0]
    batch(13)
    assert torch.cuda.mem_get_info()[0] >= free_after_4 - (1 << 20), "the sample slab grew with frame_count"
    b = P.FrameRenderer(gpu_ctx, *ids, indoor.camera_struct(), W, H)
    b.render(spp=13, bounces=4, batched=False, reset=True)
    torch.cuda.synchronize()
    assert_same(a.accum.cpu().numpy(), a.surface.cpu().numpy(), b.accum.cpu().numpy(), b.surface.cpu().numpy(), "13 frames: batched vs one launch per frame")


def test_round4_scheduling_knobs_change_no_bit(P, O, monkeypatch, indoor):
    """Scheduling paths of round 4 that the default configuration does not take: XCD-local tile regions (PTAMD_XCD_REGIONS: ticket ->
    tile through a 4 x 2 grid of the frame, frames of a tile side by side; measured slower, kept behind the knob) on an LDS-resident
    and on a wide-walk scene, ragged frame sizes and interleaved bands included; the wide walk with its pools of fresh paths back in
    LDS (PTAMD_POOL_LDS_WIDE=1: seven stack entries per lane instead of eleven).  A path's arithmetic never depends on where or
    when it runs: every frame must equal the oracle's."""
    import torch
    monkeypatch.setenv("PTAMD_TUNING", "1")
    rng = np.random.default_rng(91)
    big = make_scene(P, random_soup(rng, 2500, extent=2.5, size=0.3), lights=[((0.0, 3.0, 1.0), (1, 1, 1), 6.0, 0.7)])
    cube_big = synthetic_cubemap(rng, 4)
    cube = P.cubemap_for_scene(indoor)
    ref_big = O.render(O.OracleScene.from_host_scene(big, cube_big), O.camera_from_record(big.camera), 136, 72, spp=3, bounces=4)
    ref_in = {}
    for (w, h) in ((130, 47), (264, 136)):
        ref_in[(w, h)] = O.render(O.OracleScene.from_host_scene(indoor, cube), O.camera_from_record(indoor.camera), w, h, spp=3, bounces=3)
    for knobs in ({"PTAMD_XCD_REGIONS": "2"}, {"PTAMD_XCD_REGIONS": "1", "PTAMD_POOL_LDS_WIDE": "1"}, {"PTAMD_POOL_LDS_WIDE": "1", "PTAMD_STACK_LDS": "3"}):
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        with P.Context(0) as ctx:
            ids = (ctx.upload_scene(big), ctx.upload_cubemap(cube_big))
            for batched in (False, True):
                fr = P.FrameRenderer(ctx, *ids, big.camera_struct(), 136, 72)
                fr.render(spp=3, bounces=4, kernel=P.KERNEL_BVH_RESTART, batched=batched)
                torch.cuda.synchronize()
                assert_same(fr.accum.cpu().numpy(), fr.surface.cpu().numpy(), *ref_big, f"wide walk, {knobs}, batched={batched}")
            ids = (ctx.upload_scene(indoor), ctx.upload_cubemap(cube))
            for (w, h), ref in ref_in.items():
                fr = P.FrameRenderer(ctx, *ids, indoor.camera_struct(), w, h)
                fr.render(spp=3, bounces=3, kernel=P.KERNEL_BVH_RESTART, batched=True)
                torch.cuda.synchronize()
                assert_same(fr.accum.cpu().numpy(), fr.surface.cpu().numpy(), *ref, f"indoor {w}x{h}, {knobs}")
            # interleaved bands of a two-rank split: rank 1's bands of 8 rows equal the same rows of the full frame
            w, h = 264, 136
            fr = P.FrameRenderer(ctx, *ids, indoor.camera_struct(), w, h, interleave=(2, 1, 8))
            fr.render(spp=3, bounces=3, kernel=P.KERNEL_BVH_RESTART, batched=True)
            torch.cuda.synchronize()
            rows = np.concatenate([np.arange(b, e) for b, e in P.interleaved_bands(h, 2, 1, 8)])
            np.testing.assert_array_equal(fr.surface.cpu().numpy(), ref_in[(w, h)][1][rows], err_msg=f"interleaved bands, {knobs}")
        for k in knobs:
            monkeypatch.delenv(k)


def test_walk_only_queue_kernel_equals_the_brute_force_oracle(P, O, gpu_ctx):
    """ptamd_trace_rays_queue (the four-wide walk fed from a ray queue, a measurement hook: scripts/gpu_trace_queue.py): its records
    equal the brute-force oracle's in every configuration (16 / 20 / 24 waves per CU, treelet sizes) and refill threshold, also when
    the queue is shorter than one wave or than the grid."""
    import torch
    rng = np.random.default_rng(123)
    soup = make_scene(P, random_soup(rng, 3000, extent=3.0, size=0.25), lights=[((0.0, 1.0, 0.5), (1, 1, 1), 3.0, 0.4)])
    sid = gpu_ctx.upload_scene(soup)
    osc = O.OracleScene.from_host_scene(soup, P.cubemap_from_color())
    for n in (1, 63, 4097, 30000):
        rays = random_rays(rng, n)
        want = O.intersect(osc, rays)
        r = torch.from_numpy(rays).cuda()
        for config in (0, 1, 2, 3):
            for refill in (1, 8, 64):
                out = torch.full((n, 4), -7, dtype=torch.int32, device="cuda")
                gpu_ctx.trace_rays_queue(sid, r, out, config, refill)
                torch.cuda.synchronize()
                np.testing.assert_array_equal(out.cpu().numpy(), want, err_msg=f"n={n} config={config} refill={refill}")


def test_coincident_triangles_resolve_to_the_lowest_face_index(P, O, gpu_ctx):
    """The (t, face index) rule where it is exercised on every hit: each triangle of a soup stored three times, the copies scattered
    through the face list (so that the walk meets them in any order and in different leaves), each copy with a material of its own —
    the image shows which copy won.  The reference tests faces in storage order and keeps the first of equal distances
    (intersection.cuh:128 `t < best`): the lowest index.  Covers the hand-scheduled triangle test's unsigned-compare form of the rule
    (mt_test_asm) in the LDS-resident kernels, the compiled form in the others, and the wide walk (900 faces do not fit in LDS)."""
    rng = np.random.default_rng(77)
    for n_base, extent in ((40, 1.2), (300, 2.5)):
        base = random_soup(rng, n_base, extent=extent, size=0.9 if n_base == 40 else 0.4)
        order = rng.permutation(3 * n_base)
        tris = np.concatenate([base, base, base])[order]
        copy_of = (np.arange(3 * n_base) // n_base)[order]                      # which copy a face is: its material
        tex = [np.float32([[[0.9, 0.1, 0.1, 0.2]]]), np.float32([[[0.1, 0.9, 0.1, 0.0]]]), np.float32([[[0.1, 0.1, 0.9, 0.4]]])]
        hs = make_scene(P, tris, materials=[(0, -1, 1.0), (1, -1, 1.0), (2, -1, 1.0)], material_ids=copy_of, textures=tex,
                        lights=[((0.0, 2.0, 1.0), (1, 1, 1), 5.0, 0.5)])
        cube = synthetic_cubemap(rng, 2)
        osc = O.OracleScene.from_host_scene(hs, cube)
        rays = random_rays(rng, 20000, extent=extent)
        want = O.intersect(osc, rays)
        hit = want[:, 0] == 1
        assert hit.sum() > 2000
        # every hit names the first stored copy of its triangle
        first_copy = {}
        for f, key in enumerate(map(bytes, tris.reshape(len(tris), -1))):
            first_copy.setdefault(key, f)
        assert all(first_copy[tris[f].tobytes()] == f for f in np.unique(want[hit, 1]))
        sid = gpu_ctx.upload_scene(hs)
        for k in (P.KERNEL_BVH, P.KERNEL_BRUTE_FORCE, P.KERNEL_BVH_RESTART):
            np.testing.assert_array_equal(gpu_ctx.trace_rays(sid, rays, k), want)
        ref = O.render(osc, O.camera_from_record(hs.camera), 96, 64, spp=2, bounces=4)
        for kernel in KERNELS:
            acc, rgba = gpu_render(P, gpu_ctx, hs, cube, 96, 64, 2, 4, kid(P, kernel))
            assert_same(acc, rgba, *ref, f"coincident triangles x3, {3 * n_base} faces, {kernel}")
