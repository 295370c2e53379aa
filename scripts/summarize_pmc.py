"""Averages rocprofv3 --pmc counter CSVs per dispatch of the megakernel (un-instrumented variant) and writes
(1) the full per-dispatch summary to stdout and (2) the small record bench.py reads for its roofline block
(profiles/pmc_latest.json): VALU wave-instructions per sample, active lanes per VALU instruction, HBM bytes per launch.

    python3 summarize_pmc.py <dir with pass*/> <kernel> <latest.json> [bench args the passes were run with...]

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE is in KB and reports exactly half of the bytes of wide
coalesced streaming reads -> x2; WRITE_SIZE (KB) is exact.  SQ_* cycle counters are in quad-cycles."""
import argparse, csv, glob, json, os, sys

out_dir, kernel, latest = sys.argv[1], sys.argv[2], sys.argv[3]
ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--bounces", type=int, default=4)
ap.add_argument("--sequential", action="store_true")
ap.add_argument("--atrium", action="store_true")
ap.add_argument("--scene", default="indoor.scene")
ap.add_argument("--frames-in-flight", type=int, default=0)
ap.add_argument("--fix-backslashes", action="store_true")
ap.add_argument("--aperture", type=float, default=None)
bargs, _ = ap.parse_known_args(sys.argv[4:])

acc = {}
for f in glob.glob(os.path.join(out_dir, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if "pt_megakernel" not in name:
                continue
            nm = name.replace(" ", "")
            targs = nm.split("<", 1)[1].split(">", 1)[0].split(",") if "<" in nm else []
            bools = [t for t in targs if t in ("true", "false")]
            if len(bools) >= 2 and bools[1] == "true":      # <.., LDS_RESIDENT, STATS, ..>: skip instrumented launches
                continue
            # <LDS_RESIDENT, VARIANT>: 0 = the shipped kernel, 4 / 5 = the uninstrumented walks of big scenes (eight-wide, 64-byte four-wide);
            # 1 counters, 2 time stamps, 3 every-triangle search are not what a profile is about
            if "pt_megakernel_restart" in nm and len(targs) >= 2 and targs[1] in ("1", "2", "3"):
                continue
            want = {"bvh": "pt_megakernel<", "brute": "pt_megakernel<", "persistent": "pt_megakernel_persistent<",
                    "blockwise": "pt_megakernel_blockwise<", "split": "pt_megakernel_split<", "restart": "pt_megakernel_restart<"}.get(kernel)
            if want and want not in nm:
                continue
            c = row["Counter_Name"]; v = float(row["Counter_Value"])
            s = acc.setdefault(c, [0.0, 0])
            s[0] += v; s[1] += 1
res = {c: s[0] / s[1] for c, s in acc.items()}
res["_dispatches"] = {c: s[1] for c, s in acc.items()}
batched = (not bargs.sequential) and kernel in ("persistent", "split", "restart") and bargs.spp > 1
fpl = min(bargs.spp, 4) if batched else 1   # (ptamd_api.cpp: kMaxFramesPerSlab — longer batches are issued four frames per launch)
samples = bargs.width * bargs.height * fpl
d = {"samples_per_launch": samples}
if res.get("SQ_ACTIVE_INST_VALU") and "SQ_THREAD_CYCLES_VALU" in res:
    d["valu_active_lanes_per_inst(of 64)"] = res["SQ_THREAD_CYCLES_VALU"] / res["SQ_ACTIVE_INST_VALU"]
if res.get("SQ_BUSY_CYCLES"):
    d["mean_waves_resident(SQ_WAVE_CYCLES/SQ_BUSY_CYCLES)"] = res["SQ_WAVE_CYCLES"] / res["SQ_BUSY_CYCLES"]
if "SQ_INSTS_VALU" in res:
    d["valu_insts_per_sample"] = res["SQ_INSTS_VALU"] / samples
if "FETCH_SIZE" in res:
    d["hbm_read_bytes_per_launch(FETCH_SIZE KB x1024 x2 gfx950 correction)"] = res["FETCH_SIZE"] * 1024 * 2
    d["hbm_read_bytes_per_launch_uncorrected"] = res["FETCH_SIZE"] * 1024
if "WRITE_SIZE" in res:
    d["hbm_write_bytes_per_launch"] = res["WRITE_SIZE"] * 1024
if res.get("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in res:
            d[k + "/SQ_WAVE_CYCLES"] = res[k] / res["SQ_WAVE_CYCLES"]
if res.get("SQ_LDS_IDX_ACTIVE"):
    d["lds_bank_conflict_share_of_lds_cycles"] = res.get("SQ_LDS_BANK_CONFLICT", 0.0) / res["SQ_LDS_IDX_ACTIVE"]
res["_derived"] = d
res["_kernel"] = kernel
print(json.dumps(res, indent=1))

fetch_kb, write_kb = res.get("FETCH_SIZE"), res.get("WRITE_SIZE")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("PTAMD_NO_TORCH_PRELOAD", "1")
from cuda_pathtracer_amd import native as _native   # the id of the library the passes ran with (same work tree)
rec = {"build_id": _native.load().ptamd_build_id().decode(), "kernel": kernel, "scene": "atrium.scene" if bargs.atrium else os.path.basename(bargs.scene) + (" (textured)" if bargs.fix_backslashes else "") + (f" aperture {bargs.aperture:g}" if bargs.aperture is not None else ""), "workload": f"{bargs.width}x{bargs.height}", "spp": bargs.spp, "bounces": bargs.bounces,
       "frames_per_launch": fpl, "samples_per_launch": samples,
       "valu_insts_per_launch": res.get("SQ_INSTS_VALU"), "valu_insts_per_sample": d.get("valu_insts_per_sample"),
       "active_lanes": d.get("valu_active_lanes_per_inst(of 64)"),
       "salu_insts_per_launch": res.get("SQ_INSTS_SALU"), "branch_insts_per_launch": res.get("SQ_INSTS_BRANCH"), "lds_insts_per_launch": res.get("SQ_INSTS_LDS"),
       # GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = shader clocks the dispatch was resident for (MI355X_MICROARCH.md, DVFS)
       "gui_active_cycles_per_launch": None if "GRBM_GUI_ACTIVE" not in res else res["GRBM_GUI_ACTIVE"] / 8.0,
       "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb,
       "hbm_bytes_per_launch": None if fetch_kb is None or write_kb is None else fetch_kb * 1024 * 2 + write_kb * 1024,
       # walks served from the caches (PMC_EXTRA=l1x passes): L1 -> L2 line requests and their mean latency, L2 hits / misses,
       # share of wave cycles spent waiting
       "vmem_rd_insts_per_launch": res.get("SQ_INSTS_VMEM_RD"), "l1_accesses_per_launch": res.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
       "l1_to_l2_requests_per_launch": res.get("TCP_TCC_READ_REQ_sum"),
       "l1_to_l2_mean_latency_cycles": None if not res.get("TCP_TCC_READ_REQ_sum") else res.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / res["TCP_TCC_READ_REQ_sum"],
       "l2_hits_per_launch": res.get("TCC_HIT_sum"), "l2_misses_per_launch": res.get("TCC_MISS_sum"),
       "wait_any_share_of_wave_cycles": d.get("SQ_WAIT_ANY/SQ_WAVE_CYCLES"),
       "lds_bank_conflict_share_of_lds_cycles": d.get("lds_bank_conflict_share_of_lds_cycles"),
       "mean_waves_resident": d.get("mean_waves_resident(SQ_WAVE_CYCLES/SQ_BUSY_CYCLES)"),
       "source": "rocprofv3 --pmc passes (scripts/collect_pmc.sh), per-dispatch averages over the un-instrumented "
                 "megakernel launches of `bench.py --no-extra`; FETCH_SIZE x2 (gfx950), WRITE_SIZE exact"}
json.dump(rec, open(latest, "w"), indent=1)
