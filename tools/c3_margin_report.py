#!/usr/bin/env python3
"""Assignment margin of the BoT-SORT (C3) associations, SURVEY.md §7 hard part 1.

The cosine distances of the reference come out of Eigen's dot()/norm(), whose reduction order is unspecified (SIMD packets +
tree); this repository (kernel and oracle) fixes a k-ordered fp32 chain. The two can differ in the last bits (<~ 1e-6
relative on 256-d unit vectors). This tool asks whether such a difference could flip an assignment: it replays the cost
matrices of the oracle's BoT-SORT run on the C3 stream (dumped with ORC_LAP_DUMP), perturbs every cost by a uniform random
amount in [-delta, +delta] for several deltas and re-solves with the oracle's lapjv. Reported per delta: the fraction of
problems whose matches change. (A margin of delta means: no reordering of the fp32 reduction smaller than delta can change
track ids.) Runs on the CPU; test/analysis infrastructure only."""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    P, M, D = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (1024, 512, 256)))
    from tests import orclib
    orclib.build()
    with tempfile.TemporaryDirectory() as tmp:
        code = ("import sys; sys.path.insert(0, %r)\n"
                "from tests import orclib\nfrom motcpp_amd.synth import SynthStream\n"
                "orc = orclib.load(); t = orc.tracker(orclib.BOTSORT); s = SynthStream(%d, %d, 1234, %d)\n"
                "for f in range(%d):\n    d, e = s.next_frame(); t.update(d, e)\n" % (ROOT, P, M, D, frames))
        subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, ORC_LAP_DUMP=tmp))
        orc = orclib.load()
        rng = np.random.default_rng(0)
        deltas = [1e-7, 1e-6, 1e-5, 1e-4]
        flips = {d: 0 for d in deltas}
        total = 0
        sizes = []
        for fn in sorted(glob.glob(os.path.join(tmp, "*.bin"))):
            n, m = map(int, re.search(r"_(\d+)x(\d+)\.bin", fn).groups())
            raw = np.fromfile(fn, np.float32)
            th, c = float(raw[0]), raw[1:].reshape(n, m)
            x0, _ = orc.linear_assignment(c, th)
            total += 1
            sizes.append([n, m])
            for d in deltas:
                changed = False
                for _ in range(3):
                    cp = (c.astype(np.float64) + rng.uniform(-d, d, c.shape)).astype(np.float32)
                    x1, _ = orc.linear_assignment(cp, th)
                    changed = changed or not np.array_equal(x0, x1)
                flips[d] += int(changed)
    out = {"workload": f"BoT-SORT, {P} objects x {M} detections, {D}-d embeddings, stream seed 1234, {frames} frames",
           "problems": total, "problem_sizes_first5": sizes[:5],
           "fraction_of_problems_whose_matches_change": {f"{d:g}": flips[d] / max(total, 1) for d in deltas},
           "note": "costs perturbed uniformly in [-delta, +delta], 3 draws per problem, re-solved with the oracle's lapjv; a reordering of "
                   "the fp32 reductions moves a cosine distance by <~ 1e-6"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
