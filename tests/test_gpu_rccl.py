"""GPU: the RCCL collective path of the multi-GPU scorer, executed for real in a process group of one rank
(SURVEY.md section 4.1 "world-size-1 RCCL path on the box", section 8e).  Each case runs in a fresh child process:
a process group cannot be torn down and re-created reliably inside the pytest process, and RCCL must not see a GPU
context that other tests have been using in ways it does not expect."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "children", "rccl_ws1.py")


def _run(cmd, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NOMAD_TEST_PORT=str(port))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert "RCCL_WS1_OK" in res.stdout, res.stdout[-1500:]
    return res.stdout


def test_rccl_world_size_1_plain_process(built_lib):
    out = _run([sys.executable, CHILD], 29547)
    assert "RCCL version" in out


def test_rccl_world_size_1_under_the_distributed_launcher(built_lib):
    """The launcher the driver uses for N > 1 (python -m torch.distributed.run ...), with one process."""
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
          "--master-port", "29548", CHILD], 29548)


def test_cli_under_the_distributed_launcher_shards_predict(built_lib, tmp_path):
    """python -m torch.distributed.run --nproc-per-node 1 -m nomad_amd ...: the CLI joins the process group ("nccl" = RCCL),
    predict runs with its file sharding and collectives on (NOMAD_FORCE_COLLECTIVE=1 in this group of one rank) and
    writes the same CSV files as the plain CLI."""
    wavs = os.path.join(ROOT, "tests", "golden", "wavs")
    outs = []
    for k, launcher in enumerate((False, True)):
        out = tmp_path / f"out{k}"
        out.mkdir()
        tail = ["-m", "nomad_amd", "--mode", "dir", "--nmr", os.path.join(wavs, "nmr-data"), "--deg", os.path.join(wavs, "test-data"),
                "--results_path", str(out), "--weights", "seeded"]
        cmd = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29549"] + tail) if launcher else [sys.executable] + tail
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NOMAD_FORCE_COLLECTIVE="1" if launcher else "0")
        for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(key, None)
        res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
        assert "Nomad average scores" in res.stdout
        outs.append(out)
    for name in ("nomad_avg.csv", "nomad_scores.csv"):
        assert (outs[0] / name).read_bytes() == (outs[1] / name).read_bytes()
