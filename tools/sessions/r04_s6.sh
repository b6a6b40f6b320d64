#!/bin/bash
# round 4, session 6: batched structural ops / swap / diff-driven frames; publish-without-wait A/B; the sweep with whole rounds of clusters
set -u
O=gpurun_out/r04_s6; mkdir -p $O
export TMPDIR=/tmp
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL'
V=tools/experiments/variants/libbepuhip_nowait.so
echo "== structural tests"; timeout 900 python -m pytest tests/test_gpu_structural.py tests/test_gpu_schedule_fuzz.py -x -q 2>&1 | grep -v "$F" | tail -6 | cut -c1-400 | tee $O/pytest_structural.txt
echo "== A/B headline"; for rep in 1 2; do
  BEPUHIP_ROW_POLICY=0 timeout 200 python tools/ab_scene.py ragdoll "product:" 2>&1 | grep "ms/step" | tee -a $O/ab_nowait.txt
  BEPUHIP_LIB=$V BEPUHIP_ROW_POLICY=0 timeout 200 python tools/ab_scene.py ragdoll "nowait:" 2>&1 | grep "ms/step" | tee -a $O/ab_nowait.txt
done
BEPUHIP_ROW_POLICY=0 timeout 200 python tools/ab_scene.py pile "product:" 2>&1 | grep "ms/step" | tee -a $O/ab_nowait.txt
BEPUHIP_LIB=$V BEPUHIP_ROW_POLICY=0 timeout 200 python tools/ab_scene.py pile "nowait:" 2>&1 | grep "ms/step" | tee -a $O/ab_nowait.txt
echo "== nowait under schedule fuzzing"; BEPUHIP_LIB=$V timeout 600 python -m pytest tests/test_gpu_schedule_fuzz.py tests/test_gpu_parity.py -x -q 2>&1 | grep -v "$F" | tail -4 | cut -c1-300 | tee $O/pytest_nowait.txt
BEPUHIP_LIB=$V timeout 150 python tools/fuzz_device.py 3001 90 2>&1 | grep -v "$F" | tail -2 | cut -c1-400 | tee $O/fuzz_nowait.txt
echo "== bench (no PMC, no cpu baseline)"; timeout 900 python bench.py --no-cpu-baseline --no-traffic --steps 30 > $O/bench_quick.json 2> $O/bench_quick.err; echo "rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04_s6/bench_quick.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"])
    for e in d["scale_sweep"]["sizes"]:
        print("sweep", e["ragdolls"], round(e["ms_per_step"],4), round(e["value"]/1e9,2), "G", e["clusters"], round(e["clusters_per_cu"],2), round(e["memory_stream_frac_of_peak"],3))
    print({k: round(v,3) for k, v in d["boundary"].items() if k.endswith("_ms")})
    for k, v in d["connected_scenes"].items(): print(k, v["ms_per_step"])
except Exception as e:
    print("parse failed", e)
PY
