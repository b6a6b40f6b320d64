"""bench.py's rank bookkeeping without a GPU (`python bench.py --gpus 8` without a launcher ran ONE rank and printed an n_gpus 1 line)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench


def test_one_gpu_or_a_launcher_runs_in_this_process():
    assert bench.launch_plan(1, {}, 0, []) == ("run", None)
    assert bench.launch_plan(4, {"WORLD_SIZE": "4", "RANK": "2"}, 8, []) == ("run", None)
    assert bench.launch_plan(1, {"WORLD_SIZE": "1"}, 1, []) == ("run", None)


def test_more_gpus_without_a_launcher_starts_the_ranks_itself():
    action, cmd = bench.launch_plan(8, {}, 8, ["--gpus", "8", "--steps", "20", "--warmup", "3"])
    assert action == "spawn"
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "3"] and cmd[-7].endswith("bench.py")
    # the port is the caller's if it named one
    assert bench.launch_plan(2, {"MASTER_PORT": "31111"}, 2, [])[1][bench.launch_plan(2, {"MASTER_PORT": "31111"}, 2, [])[1].index("--master-port") + 1] == "31111"


def test_never_an_n1_line_for_an_n8_request():
    action, text = bench.launch_plan(8, {}, 1, ["--gpus", "8"])
    assert action == "error" and "8" in text and "1 GPU" in text
    action, text = bench.launch_plan(8, {"WORLD_SIZE": "4"}, 8, [])
    assert action == "error" and "WORLD_SIZE 4" in text
    action, text = bench.launch_plan(2, {"WORLD_SIZE": "1"}, 8, [])  # a launcher that started one rank for a two-GPU request
    assert action == "error"
    assert bench.launch_plan(0, {}, 8, [])[0] == "error"


def test_the_compact_line_carries_the_n_gt_1_legs():
    out = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 8, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic", "config": {"workload": "w"}, "roofline": None, "cpu_baseline": None,
           "lattice_device_group": {"value": 2.0e10, "ms_per_step": 0.4, "scaling": "strong", "n_gpus": 8, "config": {"exchanges_per_step": 1, "finite": True, "schedule_is_island": True, "workload": "long text"}},
           "self_checks": {"world": 8, "peer_access_all_pairs": True}}
    line = bench.compact_line(out, None)
    assert line["lattice_device_group"] == {"value": 2.0e10, "ms_per_step": 0.4, "scaling": "strong", "n_gpus": 8, "exchanges_per_step": 1, "finite": True, "schedule_is_island": True}
    assert line["self_checks"]["world"] == 8
