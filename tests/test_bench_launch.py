"""`python bench.py --gpus N` must be able to start its own N ranks (the driver's invocation has no torchrun around it):
the parent builds a torch.distributed.run command, runs it as a CHILD and relays the exit code -- and never touches the
GPU itself. Checked here on CPU through --dry-launch and by running the parent path with a stand-in child."""
import json
import os
import subprocess
import sys

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_dry_launch_command():
    out = subprocess.check_output([sys.executable, BENCH, "--gpus", "4", "--steps", "7", "--warmup", "2", "--dry-launch"],
                                  env=_clean_env(), text=True)
    cmd = json.loads(out)["launch"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    tail = cmd[cmd.index(BENCH):]
    assert tail == [BENCH, "--gpus", "4", "--steps", "7", "--warmup", "2"]        # same args, --dry-launch dropped


def test_single_gpu_and_torchrun_children_do_not_relaunch():
    out = subprocess.check_output([sys.executable, BENCH, "--dry-launch"], env=_clean_env(), text=True)
    assert json.loads(out)["launch"] is None
    env = dict(_clean_env(), WORLD_SIZE="4", RANK="1", LOCAL_RANK="1")       # a rank started by torchrun
    out = subprocess.check_output([sys.executable, BENCH, "--gpus", "4", "--dry-launch"], env=env, text=True)
    assert json.loads(out)["launch"] is None


def test_parent_relays_child_and_never_imports_torch(tmp_path):
    """The parent path itself: bench.main() with --gpus 2 and a stand-in child command. It must return the child's
    exit code and must not have imported torch (let alone initialised a device) on the way."""
    probe = tmp_path / "probe.py"
    probe.write_text(
        "import sys, json\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "bench.launch_command = lambda n, argv, port=None: [sys.executable, '-c', 'import sys; print(\"child ran\"); sys.exit(7)']\n"
        "sys.argv = ['bench.py', '--gpus', '2']\n"
        "rc = bench.main()\n"
        "print(json.dumps({'rc': rc, 'torch': 'torch' in sys.modules}))\n" % REPO)
    out = subprocess.check_output([sys.executable, str(probe)], env=_clean_env(), text=True)
    lines = out.strip().splitlines()
    assert lines[0] == "child ran"
    assert json.loads(lines[-1]) == {"rc": 7, "torch": False}
