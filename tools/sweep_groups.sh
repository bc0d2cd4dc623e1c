#!/bin/bash
# experiment: chain groups x light-kernel priority (run on the GPU box)
for G in 1 2 3 4; do for P in 0 1; do
  RMHMC_GROUPS=$G RMHMC_PRIO=$P timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "G=$G P=$P failed"; tail -3 gpurun_out/sw.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/sw.json"))
print("G=$G P=$P value=%.0f ms/step=%.2f"%(d["value"], d["ms_per_step"]), {k:round(v["seconds"]*1000/6,2) for k,v in d["kernel_seconds"].items()})
PY
done; done
