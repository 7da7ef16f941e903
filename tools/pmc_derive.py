"""Turns the per-kernel PMC sums of tools/collect_profiles.sh into the two small files bench.py quotes:
  profiles/pmc_<WL>.json            HBM bytes per assignment problem (FETCH_SIZE doubled: gfx950 reports half of wide reads)
  profiles/<tag>_pmc_sq_lap.json    SQ counters of the dominant assignment kernel per problem (= per workgroup)
usage: python tools/pmc_derive.py <tag> <WL> <streams per launch (= per sub-batch) of the PMC passes> ["<bench command of the passes>"]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_sources_hash import kernel_sources_hash

tag, wl, S = sys.argv[1], sys.argv[2], int(sys.argv[3])
cmd = sys.argv[4] if len(sys.argv) > 4 else f"bench.py --workload {wl} --streams {S} --pipeline 1 --steps 2 --warmup 5"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
ld = lambda n: json.load(open(os.path.join(P, f"{tag}_pmc_{n}_{wl}.json")))
fetch, write, sq1, sq2 = ld("fetch"), ld("write"), ld("sq1"), ld("sq2")
# the assignment kernels of a frame: the sparse certified solver (every problem) + the exact solver (the problems it declined)
laps = [k for k in fetch if k.startswith("lap_")]
main = max((k for k in laps if k.startswith("lap_sparse")), key=lambda k: fetch[k]["FETCH_SIZE"]["sum"])
disp = fetch[main]["FETCH_SIZE"]["dispatches"]
per_frame = 3.0 * S  # ByteTrack: S first-association problems in one launch, 2S (second + unconfirmed) in the other: both launches are summed
per_launch = per_frame
fb = sum(fetch[k]["FETCH_SIZE"]["sum"] for k in laps) * 1024.0 / disp
wb = sum(write[k]["WRITE_SIZE"]["sum"] for k in laps if k in write) * 1024.0 / disp
out = {"kernel_sources_sha": kernel_sources_hash(), "tag": tag,
       "_comment": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `{cmd}` (device lifecycle, {S} streams per launch); kernels {laps}; KB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md (an upper bound for "
                   f"4-byte accesses). Raw sums: {tag}_pmc_fetch_{wl}.json, {tag}_pmc_write_{wl}.json",
       "lap": {"fetch_bytes_per_launch_raw": fb, "write_bytes_per_launch": wb, "problems_per_launch": per_launch,
               "hbm_bytes_per_problem": (2.0 * fb + wb) / per_launch}}
# the dominant kernel ALONE (bench.py's roofline family lap1_sparse: the first association's sparse solver, S problems per dispatch):
# comparable with roofline.algorithmic_bytes_per_launch, which counts that launch only
mdisp = fetch[main]["FETCH_SIZE"]["dispatches"]
mfb = fetch[main]["FETCH_SIZE"]["sum"] * 1024.0 / mdisp
mwb = (write[main]["WRITE_SIZE"]["sum"] * 1024.0 / write[main]["WRITE_SIZE"]["dispatches"]) if main in write else 0.0
out["lap1_sparse"] = {"kernel": main, "fetch_bytes_per_launch_raw": mfb, "write_bytes_per_launch": mwb, "problems_per_launch": S,
                      "hbm_bytes_per_problem": (2.0 * mfb + mwb) / S}
json.dump(out, open(os.path.join(P, f"pmc_{wl}.json"), "w"), indent=1)
c = {}
for src in (sq1, sq2):
    for name, v in src[main].items():
        c[name] = v["sum"] / v["dispatches"] / S  # (the dominant kernel = the first association's sparse solver: S problems per dispatch)
sq = {"kernel_sources_sha": kernel_sources_hash(), "tag": tag, "kernel": main, "problems_per_launch": S, "waves_per_problem": c.get("SQ_WAVES"),
      "per_problem": {k: round(v, 1) for k, v in c.items() if k != "SQ_WAVES"},
      "active_frac": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3), "wait_any_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)}
path = os.path.join(P, f"{tag}_pmc_sq_lap.json")
allsq = json.load(open(path)) if os.path.exists(path) else {}
allsq[wl] = sq
json.dump(allsq, open(path, "w"), indent=1)
print(json.dumps(out["lap"]), json.dumps(sq))
