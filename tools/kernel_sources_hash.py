"""sha256 over the kernel sources (motcpp_amd/csrc/*.hip, *.hpp, sorted by name): what a counter file under profiles/ was collected for.
tools/pmc_traffic_derive.py and tools/pmc_derive.py record it, bench.py compares it with the sources it runs on and marks a counter file
`stale` when they differ (git is not available on the GPU box; the sources are).   python tools/kernel_sources_hash.py -> prints the hash"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "motcpp_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_sources_hash())
