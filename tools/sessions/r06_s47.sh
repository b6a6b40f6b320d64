# round 6, session 47: rehearsal of the N > 1 line's device-group leg (one child per rank on a rendezvous of its own) on a one-GPU box; --gpus 2 on one GPU must refuse
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s47
mkdir -p $O
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 BEPU_BENCH_FORCE_DIST=1 BEPU_BENCH_GROUP_LEG=1 timeout 900 python bench.py --no-traffic --no-scale-sweep --no-cpu-baseline --steps 20 --warmup 3 --full-report $O/bench_group_leg_full.json > $O/bench_group_leg.json 2> $O/bench_group_leg.err
echo "rc $?"; tail -3 $O/bench_group_leg.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/s47/bench_group_leg.json"))
print("value", d["value"], "ms", d["ms_per_step"], "n_gpus", d["n_gpus"])
print("lattice_device_group", d.get("lattice_device_group"))
print("lattice", d.get("lattice"))
PY
timeout 120 python bench.py --gpus 2 --steps 5 > $O/gpus2_on_one.txt 2>&1; echo "--gpus 2 on one GPU: rc $?"; tail -2 $O/gpus2_on_one.txt
