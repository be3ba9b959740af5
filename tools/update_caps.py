#!/usr/bin/env python
"""tests/golden/measured_caps.json from gpurun_out/test_metrics.jsonl: cap = 2 x the maximum the GPU tests measured for every
`bounded(...)` comparison -- nothing else: round 3 floored every cap at the comparison's own tolerance, which made "2 x measured"
read `max(2 x measured, 2e-3)` for most fields; the floor is gone (a comparison that measured exactly 0 gets the fp32 unit
roundoff, 1.2e-7, so that a last-bit change of a library does not fail it).  Run the GPU tests with ROBIR_RECORD_CAPS=1 first
(records without asserting the caps), under every precision policy the suite is run with: the maxima of all recorded runs count."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(ROOT, "gpurun_out", "test_metrics.jsonl")
caps, meas = {}, {}
for line in open(src):
    d = json.loads(line)
    if d["name"].startswith("bounded/"):
        n = d["name"][len("bounded/"):]
        meas[n] = max(meas.get(n, 0.0), d["max"])
        caps[n] = max(2.0 * meas[n], 1.2e-7)
dst = os.path.join(ROOT, "tests", "golden", "measured_caps.json")
prev = json.load(open(dst)) if os.path.exists(dst) else {}
if "--fresh" in sys.argv and len(caps) < len(prev):      # a metrics file of a partial run (gpurun merges the LAST call's file): keep the caps
    sys.exit(f"{src} holds {len(caps)} bounded comparisons, the caps file {len(prev)}: run the whole GPU suite (both precision policies) "
             "with ROBIR_RECORD_CAPS=1 before --fresh")
old = prev if "--fresh" not in sys.argv else {}
if "--raise-only" in sys.argv:         # a re-measurement after an arithmetic change: never tighten what another policy / run needed
    caps = {k: max(v, old.get(k, 0.0)) for k, v in caps.items()}
old.update(caps)                      # comparisons not re-measured in this run keep their recorded cap
out = {k: float("%.3g" % v) for k, v in sorted(old.items())}
json.dump(out, open(dst, "w"), indent=0, sort_keys=True)
for k in sorted(meas):
    print(f"{k:70s} measured {meas[k]:.3g}  cap {out[k]:.3g}")
