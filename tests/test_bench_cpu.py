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


# ---- bench.main() end to end with TWO ranks (gloo, CPU stand-ins for the device work) ----------------------------------
def _main_worker(rank, world, port, q, fault=None, out_path=None):
    import contextlib
    import io
    import json
    import types

    import torch
    import torch.distributed as dist

    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    calls = []

    def collective(tag):                      # a section every rank must enter: a mismatch would deadlock (test timeout)
        t = torch.ones(1)
        dist.all_reduce(t)
        assert int(t.item()) == world, tag
        calls.append(tag)

    class Stepper:
        def __init__(self):
            self.n = 0

        def prefill(self, prompt):
            assert prompt.shape == (1, 16)

        def step(self):
            self.n += 1

    class CpuHooks(bench.Hooks):
        backend = "gloo"

        def device(self, local_rank):
            return torch.device("cpu")

        def init_process_group(self, dev, timeout_s=120.0):
            import datetime
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s))

        def sync(self, dev):
            pass

        def empty_cache(self):
            pass

        def load_lib(self):
            pass

        def build_model(self, cfg, seed, dev):
            calls.append(("build", seed))
            return types.SimpleNamespace(config=cfg)

        def make_stepper(self, model, max_len):
            self.stepper = Stepper()
            return self.stepper

        @staticmethod
        def measure_roofline(model, dev, ms, tok_bytes):
            assert dist.get_rank() == 0
            return {"bound": "hbm", "frac": 0.1}

        @staticmethod
        def measure_prefill_sharded(cfg, dev, world_, rank_):
            collective("prefill_sharded")
            return {"k_shards": world_}

        @staticmethod
        def measure_continuous_batch(model, dev):
            assert dist.get_rank() == 0
            return {"slots": 32}

        @staticmethod
        def measure_prefill_model(model, dev):
            assert dist.get_rank() == 0
            if fault == "raise":                                 # an injected failure of a secondary leg: recorded, nothing else lost
                raise RuntimeError("injected failure")
            return {"ms": 1.0}

        @staticmethod
        def measure_decode_ctx(model, dev):
            assert dist.get_rank() == 0
            return {"single_stream": {}}

        @staticmethod
        def measure_eval(model, dev):
            assert dist.get_rank() == 0
            return {"perplexity": {}}

        @staticmethod
        def measure_train_layer(dev):
            assert dist.get_rank() == 0
            return {"TFLOPs": 1.0}

        @staticmethod
        def measure_prefill_model_tp(model, dev, world_, rank_):
            collective("prefill_model_tp")
            return {"tp_degree": world_}

        @staticmethod
        def measure_k_sharded_decode(cfg, dev, world_, rank_, steps, prompt):
            if fault == "hang" and rank_ == 1:                   # one rank never arrives: rank 0 is stuck in the collective below
                import time
                time.sleep(3600)
            collective("k_sharded_decode")
            assert cfg.hidden_size == 5120                      # config 4 runs on 13B shapes at every N
            return {"k_shards": world_}

        @staticmethod
        def measure_cpu_baseline(cfg):
            assert dist.get_rank() == 0
            return {"value": 1.0, "unit": "tokens/s", "cores": 1, "kind": "port", "sample": "stand-in"}

    hooks = CpuHooks()
    if fault == "hang":
        # the deadline fires while the main thread hangs in a collective: the line goes to a file (the process leaves with os._exit)
        sys.stdout = open(out_path + ".%d" % rank, "w")
        bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--model", "tiny", "--deadline", "6", "--pg-timeout", "600"], hooks=hooks)
        raise AssertionError("main() returned although a leg hangs")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--model", "tiny"], hooks=hooks)
    if fault == "raise":
        if rank == 0:
            q.put(json.loads(buf.getvalue().strip().splitlines()[-1]))
        return
    assert hooks.stepper.n == 4                                 # warm-up + timed steps, every rank
    assert [c for c in calls if isinstance(c, str)] == ["prefill_sharded", "prefill_model_tp", "k_sharded_decode"]
    # replicas are seeded per rank; the tensor-parallel leg rebuilds the SAME checkpoint on every rank
    assert ("build", 1000 * rank) in calls and ("build", 4242) in calls
    out = buf.getvalue().strip()
    if rank == 0:
        q.put(json.loads(out.splitlines()[-1]))
    else:
        assert out == ""                                        # one JSON line per job, from rank 0 only


def test_main_control_flow_two_ranks_gloo():
    """bench.py's main() with --gpus 2 executed by two gloo ranks (device work replaced by stand-ins that keep the
    collectives): which ranks run what, where the barriers sit, the JSON line -- in particular `cpu_baseline` is
    present at N > 1 and config 4 / tensor-parallel prefill are entered by every rank."""
    import socket

    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_main_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    line = q.get(timeout=5)
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["metric"] == "decode_tokens_per_sec" and line["value"] > 0
    assert line["cpu_baseline"] is not None and line["cpu_baseline"]["kind"] == "port"        # emitted at every N
    assert line["roofline"]["bound"] == "hbm"
    for k in ("prefill_k_sharded", "decode_k_sharded", "continuous_batch", "prefill_model", "prefill_model_tp", "eval_ppl", "train_layer"):
        assert k in line, k
    assert line["decode_k_sharded"]["k_shards"] == 2 and line["decode_k_sharded"]["model"] == "LLaMA-13B shapes"


def _spawn2(fault, out_path=None):
    import socket

    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_main_worker, args=(r, 2, port, q, fault, out_path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0, p.exitcode
    return q


def test_failed_secondary_leg_is_recorded_and_nothing_else_is_lost():
    """A secondary leg that raises (injected into rank 0's prefill_model leg): its field carries the error, every other field --
    the headline, the legs every rank enters, cpu_baseline -- is there, and the job ends normally on both ranks."""
    line = _spawn2("raise").get(timeout=5)
    assert line["prefill_model"] == {"error": "RuntimeError: injected failure"}
    assert line["value"] > 0 and "incomplete" not in line
    for k in ("roofline", "cpu_baseline", "prefill_k_sharded", "decode_k_sharded", "continuous_batch", "prefill_model_tp", "eval_ppl", "decode_ctx"):
        assert k in line and line[k] is not None, k


def test_hung_secondary_leg_cannot_suppress_the_headline_line(tmp_path):
    """Rank 1 never enters the config-4 leg's collective, rank 0 hangs in it: at the deadline rank 0 prints the ONE JSON line with
    everything measured so far and `incomplete` naming the leg, both ranks exit with status 0."""
    import json
    out_path = str(tmp_path / "bench_out")
    _spawn2("hang", out_path)
    text = open(out_path + ".0").read().strip()
    assert len(text.splitlines()) == 1
    line = json.loads(text)
    assert line["metric"] == "decode_tokens_per_sec" and line["value"] > 0 and line["n_gpus"] == 2
    assert line["incomplete"]["legs_not_finished"] == ["decode_k_sharded"]
    assert line["roofline"]["bound"] == "hbm" and line["prefill_k_sharded"]["k_shards"] == 2     # the legs before the hang survived
    assert "decode_k_sharded" not in line
    assert open(out_path + ".1").read().strip() == ""
