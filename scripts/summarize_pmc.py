"""Averages rocprofv3 --pmc counter CSVs per dispatch of the megakernel (un-instrumented variant)."""
import csv, glob, json, os, sys
out_dir, kernel = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(out_dir, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if "pt_megakernel" not in name or "true>(" in name.replace(" ", "").replace("true,true>", "true>("):
                pass
            if "pt_megakernel" not in name:
                continue
            nm = name.replace(" ", "")
            targs = nm.split("<", 1)[1].split(">", 1)[0].split(",") if "<" in nm else []
            bools = [t for t in targs if t in ("true", "false")]
            if len(bools) >= 2 and bools[1] == "true":      # <.., LDS_RESIDENT, STATS, ..>: skip instrumented launches
                continue
            want = {"bvh": "pt_megakernel<", "brute": "pt_megakernel<", "persistent": "pt_megakernel_persistent<",
                    "blockwise": "pt_megakernel_blockwise<"}.get(kernel)
            if want and want not in nm:
                continue
            c = row["Counter_Name"]; v = float(row["Counter_Value"])
            s = acc.setdefault(c, [0.0, 0])
            s[0] += v; s[1] += 1
res = {c: s[0] / s[1] for c, s in acc.items()}
res["_dispatches"] = {c: s[1] for c, s in acc.items()}
d = {}
if "SQ_THREAD_CYCLES_VALU" in res and "SQ_ACTIVE_INST_VALU" in res and res["SQ_ACTIVE_INST_VALU"]:
    d["valu_active_lanes_per_inst(of 64)"] = res["SQ_THREAD_CYCLES_VALU"] / res["SQ_ACTIVE_INST_VALU"] / 4.0 * 4.0 / 1.0
if "SQ_WAVE_CYCLES" in res and "SQ_BUSY_CYCLES" in res and res["SQ_BUSY_CYCLES"]:
    d["mean_waves_resident(SQ_WAVE_CYCLES/SQ_BUSY_CYCLES)"] = res["SQ_WAVE_CYCLES"] / res["SQ_BUSY_CYCLES"]
if "FETCH_SIZE" in res:
    d["hbm_read_bytes_per_launch(FETCH_SIZE KB x1024 x2 gfx950 correction)"] = res["FETCH_SIZE"] * 1024 * 2
    d["hbm_read_bytes_per_launch_uncorrected"] = res["FETCH_SIZE"] * 1024
if "WRITE_SIZE" in res:
    d["hbm_write_bytes_per_launch"] = res["WRITE_SIZE"] * 1024
res["_derived"] = d
res["_kernel"] = kernel
print(json.dumps(res, indent=1))
