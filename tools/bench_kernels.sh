#!/bin/bash
# headline step only (no variants / other configs): ms per step, eager, the probed gather kernels
set -eu
python bench.py --no-variants --no-configs "$@" 2>&1 | tail -1 > gpurun_out/_bk.json
python - <<PY
import json
o=json.load(open("gpurun_out/_bk.json"))
print("ms/step", round(o["ms_per_step"],4), "eager", round(o["repeats"]["eager_ms_per_step"],4), {k:round(v["ms"],4) for k,v in o.get("kernels",{}).items()})
PY
