"""Turns the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (scripts/collect_pmc.sh) into
profiles/traffic_latest.json, which bench.py reads for roofline.traffic.

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE is in KB and reports exactly half of the
bytes of wide coalesced streaming reads -> x2; WRITE_SIZE (KB) is exact.  The megakernel's reads are
mostly 4-byte-per-lane accumulator gathers, for which the guide says the factor is uncalibrated; both
the corrected and the raw figure are recorded."""
import json, os, sys
summary, kernel, out = sys.argv[1], sys.argv[2], sys.argv[3]
frames_per_launch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
d = json.load(open(summary))
fetch_kb, write_kb = d.get("FETCH_SIZE"), d.get("WRITE_SIZE")
res = {"kernel": kernel, "workload": "1920x1080", "frames_per_launch": frames_per_launch, "bounces": 4,
       "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb,
       "hbm_read_bytes_per_launch_corrected_x2": None if fetch_kb is None else fetch_kb * 1024 * 2,
       "hbm_write_bytes_per_launch": None if write_kb is None else write_kb * 1024,
       "hbm_bytes_per_launch": None if fetch_kb is None or write_kb is None else fetch_kb * 1024 * 2 + write_kb * 1024,
       "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), averaged over the un-instrumented megakernel dispatches"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
