"""pmc_<WL>.json (what bench.py's roofline.traffic reads) from the per-kernel FETCH_SIZE / WRITE_SIZE sums of two separate rocprofv3 --pmc passes.
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KB; gfx950 counts half of the wide reads: MI355X_MICROARCH.md — an upper bound for narrow
accesses), divided by the dispatches of the family's main kernel and the problems (streams) per dispatch.
usage: python tools/pmc_traffic_derive.py <fetch.json> <write.json> <WL> <family> <streams per launch> "<command>" """
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_sources_hash import kernel_sources_hash

fetch, write = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
wl, fam, S, cmd = sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
pref = {"cosine": ("embed_kernel<",), "lap": ("lap_sparse_kernel<", "lap_kernel<"), "lap1_sparse": ("lap_sparse_kernel<",), "feat": ("feat_kernel",)}[fam]
ks = [k for k in fetch if k.startswith(pref)]
if not ks:
    raise SystemExit(f"no kernel of family {fam} in the counter files: {list(fetch)[:8]}")
main = max(ks, key=lambda k: fetch[k]["FETCH_SIZE"]["sum"])
if fam == "lap1_sparse":
    ks = [main]
disp = fetch[main]["FETCH_SIZE"]["dispatches"]
fb = sum(fetch[k]["FETCH_SIZE"]["sum"] for k in ks) * 1024.0 / disp
wb = sum(write[k]["WRITE_SIZE"]["sum"] for k in ks if k in write) * 1024.0 / disp
out = {"kernel_sources_sha": kernel_sources_hash(), "tag": os.environ.get("MOT_EVIDENCE_TAG", ""),
       "_comment": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, nothing else traced) on `{cmd}`; kernels {ks}; KB -> bytes; FETCH_SIZE doubled "
                   "(MI355X_MICROARCH.md: an upper bound for narrow accesses); per dispatch of the family's largest kernel, divided by the streams of the launch",
       fam: {"kernel": main, "kernels_summed": ks, "dispatches": disp, "fetch_bytes_per_launch_raw": fb, "write_bytes_per_launch": wb, "problems_per_launch": S,
             "hbm_bytes_per_problem": (2.0 * fb + wb) / S}}
print(json.dumps(out, indent=1))
