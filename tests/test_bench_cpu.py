"""Host logic of bench.py that needs no GPU: the multi-GPU entry point must launch its own ranks
when it is started as a plain process, and must refuse to report a rank count it was not given."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spawn_command_shape():
    import bench
    cmd = bench.spawn_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "1"], port=23456)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "23456"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    port = bench.spawn_command(2, [])[bench.spawn_command(2, []).index("--master-port") + 1]
    assert 1024 < int(port) < 65536                       # a free port was picked


def test_plain_process_with_gpus_n_spawns_n_ranks(monkeypatch):
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    assert seen["cmd"][seen["cmd"].index("--nproc-per-node") + 1] == "4"
    assert seen["cmd"][-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_rank_count_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode != 0
    assert "WORLD_SIZE 3" in r.stderr and "--gpus 2" in r.stderr
    assert r.stdout.strip() == ""                          # no JSON line for an unmeasured configuration


def test_csrc_sha_is_stable():
    import bench
    assert bench.csrc_sha() == bench.csrc_sha() and len(bench.csrc_sha()) == 16
