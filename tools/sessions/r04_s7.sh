#!/bin/bash
# round 4, session 7: the whole GPU suite (velocity models, batched structural ops, diff-driven frames, fuzz slices), then the bench line without PMC / cpu baseline
set -u
O=gpurun_out/r04_s7; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -v "$F" $O/pytest_gpu.txt | tail -8 | cut -c1-400
timeout 900 python bench.py --no-cpu-baseline --no-traffic --steps 30 > $O/bench_quick.json 2> $O/bench_quick.err; echo "bench rc=$?"; tail -3 $O/bench_quick.err | cut -c1-300
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04_s7/bench_quick.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"])
    for e in d["scale_sweep"]["sizes"]:
        print("sweep", e["ragdolls"], round(e["ms_per_step"],4), round(e["value"]/1e9,2), "G", e["clusters"], round(e["clusters_per_cu"],2), round(e["memory_stream_frac_of_peak"],3))
    print({k: round(v,3) for k, v in d["boundary"].items() if k.endswith("_ms")})
    for k, v in d["connected_scenes"].items(): print(k, v["ms_per_step"])
    print("widened", d["widened_types"]["ms_per_step"])
except Exception as e:
    print("parse failed", e)
PY
