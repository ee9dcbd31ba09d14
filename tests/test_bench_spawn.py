"""bench.py --gpus N from a plain shell: the parent spawns the ranks itself (CPU part: no GPU here, so the ranks fail -
cleanly, with the parent relaying the status and touching neither torch nor a GPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parent_spawns_ranks_and_fails_cleanly_without_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-also"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert res.stdout.strip() == ""                                   # no half-written JSON line
    assert "needs a GPU" in res.stderr                                # the ranks' own message came through the launcher
    assert "2-rank run failed" in res.stderr


def test_spawn_command_line_drops_the_spawn_flag(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class R:
        returncode = 0
        stdout = 'noise\n{"metric": "x"}\n'

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw["env"]
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    assert bench.spawn_ranks(4, ["--gpus", "4", "--spawn", "--steps", "3"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and "--spawn" not in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
