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


def test_the_device_group_leg_is_a_job_of_its_own():
    """At N > 1 the configs[4] leg runs as one child per rank (a standalone `--lattice --lattice-exact` run on a rendezvous of its own): a child that aborts or hangs
    costs the leg's report, never the headline's line."""
    import argparse
    args = argparse.Namespace(steps=20, warmup=3, ragdolls=15000, no_prewarm=False)
    cmd = bench.lattice_group_command(args, 8)
    assert cmd[1].endswith("bench.py") and cmd[cmd.index("--gpus") + 1] == "8" and "--lattice" in cmd and "--lattice-exact" in cmd
    assert cmd[cmd.index("--steps") + 1] == "20" and cmd[cmd.index("--warmup") + 1] == "3" and cmd[cmd.index("--ragdolls") + 1] == "15000"
    # the child is a rank of a launcher-made world: launch_plan runs it in-process instead of spawning again
    assert bench.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3", "MASTER_PORT": "40000"}, 8, cmd[2:]) == ("run", None)
    out = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 8, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic", "config": {"workload": "w"}, "roofline": None, "cpu_baseline": None,
           "lattice_device_group": {"error": "the device-group leg did not complete on every rank", "ranks": [{"rank": 5, "exit_code": -6}], "partial": None}}
    line = bench.compact_line(out, None)
    assert line["value"] == 1.0 and line["lattice_device_group"]["ranks"] == [{"rank": 5, "exit_code": -6}]


def test_stdout_carries_the_line_and_nothing_else():
    """RCCL writes a version banner to the C library's stdout when its first communicator comes up (flushed at exit: behind the JSON line in a pipe). bench.py moves
    descriptor 1 to stderr for the run and writes its one line to the descriptor stdout used to be."""
    import subprocess
    code = ("import bench, ctypes\n"
            "bench.claim_stdout(); print('python noise'); ctypes.CDLL(None).puts(b'c library noise'); bench.emit_line({'metric': 'm', 'value': 1})\n")
    done = subprocess.run([sys.executable, "-c", code], cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert done.returncode == 0, done.stderr.decode()
    assert done.stdout.decode() == '{"metric": "m", "value": 1}\n'
    assert "python noise" in done.stderr.decode() and "c library noise" in done.stderr.decode()
