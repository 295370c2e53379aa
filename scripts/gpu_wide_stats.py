#!/usr/bin/env python3
"""Instrumented-build statistics of the wide walk on a big scene (default: the atrium, configs[3]): visits and triangle tests per ray,
lane utilisation of the box and leaf phases, rounds per path.  One batched launch of SPP frames through ptamd_raytrace_stats.

    python3 scripts/gpu_wide_stats.py [--tessellate N]   ->  one JSON line on stdout
"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import cuda_pathtracer_amd as P  # noqa: E402
from cuda_pathtracer_amd.synthetic import write_atrium  # noqa: E402

W, H = 1920, int(os.environ.get("HEIGHT", 1080))
B, SPP = int(os.environ.get("BOUNCES", 4)), int(os.environ.get("SPP", 4))
tess = int(sys.argv[sys.argv.index("--tessellate") + 1]) if "--tessellate" in sys.argv else 0
with tempfile.TemporaryDirectory(prefix="ptamd_atrium_") as d:
    if tess:
        hs = P.tessellate(P.HostScene.load(os.path.join(ROOT, "assets", "indoor.scene")), tess)
    else:
        hs = P.HostScene.load(write_atrium(d))
    with P.Context(0) as ctx:
        sid, cid = ctx.upload_scene(hs), ctx.upload_cubemap(P.cubemap_for_scene(hs, asset_folder=os.path.join(ROOT, "assets")))
        fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), W, H)
        l = ctx.make_launch(fr.surface, fr.accum, sid, cid, hs.camera_struct(), W, H, frame_nb=1, bounces=B,
                            kernel=P.KERNEL_BVH_RESTART, frame_count=SPP)
        s = ctx.raytrace_stats(l)
        cyc = ctx.phase_cycles()
        rays, samples = max(s["rays"], 1), max(s["samples"], 1)
        out = {k: v for k, v in s.items() if v}
        out.update({
            "scene": "indoor x%d^2" % tess if tess else "atrium", "spp": SPP, "bounces": B,
            "rays_per_sample": rays / samples, "node_visits_per_ray": s["nodes_visited"] / rays, "tri_tests_per_ray": s["tris_tested"] / rays,
            "box_phase_lane_utilisation": s["nodes_visited"] / (64.0 * max(s["wave_node_iters"], 1)),
            "leaf_phase_lane_utilisation": s["tris_tested"] / (64.0 * max(s["wave_tri_iters"], 1)),
            "wave_box_iterations_per_sample": s["wave_node_iters"] / samples,
            "wave_tri_iterations_per_sample": s["wave_tri_iters"] / samples,
            "rounds_per_64_samples": s["fetch_events"] * 64.0 / samples,
            "walks_completed_per_round": s["fetch_rays"] / max(s["fetch_events"], 1),
            "box_iteration_lane_slots": {"active": s["nodes_visited"] / (64.0 * max(s["wave_node_iters"], 1)),
                                         "no_path_in_this_round": s["idle_unstarted"] / (64.0 * max(s["wave_node_iters"], 1)),
                                         "walk_over_waiting_for_round_end": s["idle_finished"] / (64.0 * max(s["wave_node_iters"], 1)),
                                         "parked_at_a_leaf": s["idle_parked"] / (64.0 * max(s["wave_node_iters"], 1))},
            "max_leaf_size": ctx.scene_info(sid)["max_leaf_size"],
            "phase_cycles_summed_over_waves": cyc,
            "phase_share_of_round_loop": {k: v / max(cyc["round_loop"], 1) for k, v in cyc.items() if k in ("refill", "box_phases", "leaf_phases", "lights_and_shading")},
            "cycles_per_box_iteration": cyc["box_phases"] / max(s["wave_node_iters"], 1),
            "cycles_per_node_fetch_issue_to_data": cyc["node_fetches"] / max(s["wave_node_iters"], 1),
            "cycles_per_visit_fetch_tests_pushes_pops": cyc["visits"] / max(s["wave_node_iters"], 1),
            "cycles_per_leaf_phase": cyc["leaf_phases"] / max(cyc["leaf_phases_entered"], 1),
            "cycles_per_round_of_shading": cyc["lights_and_shading"] / max(s["fetch_events"], 1),
            "cycles_per_round": {"r1_and_light_loop": cyc["light_loop"] / max(s["fetch_events"], 1),
                                 "shading_record_fetch": cyc["shading_record_fetch"] / max(s["fetch_events"], 1),
                                 "path_post_incl_fetch_and_parking": cyc["path_post_and_parking"] / max(s["fetch_events"], 1),
                                 "path_post_up_to_the_bsdf_sample": cyc["path_post_to_bsdf"] / max(s["fetch_events"], 1),
                                 "bsdf_sample": cyc["bsdf_sample"] / max(s["fetch_events"], 1)},
            "build_id": P.native.load().ptamd_build_id().decode(),
        })
        print(json.dumps(out))
