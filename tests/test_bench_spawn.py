"""bench.py's self-launch path (CPU side, child process mocked): `python bench.py --gpus N`
without torchrun must start the N ranks as a child BEFORE anything touches a GPU, relay rank 0's
single JSON line and return the child's exit code; a WORLD_SIZE / --gpus mismatch is an error."""
import io
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class _Run:
    def __init__(self, stdout, rc=0, stderr=""):
        self.calls = []
        self.result = types.SimpleNamespace(stdout=stdout, stderr=stderr, returncode=rc)

    def __call__(self, cmd, **kw):
        self.calls.append((cmd, kw))
        return self.result


def _spawn(argv, environ, run):
    out, err = io.StringIO(), io.StringIO()
    rc = bench.maybe_spawn(bench.parse(argv), argv, environ=environ, run=run, out=out, err=err)
    return rc, out.getvalue(), err.getvalue()


def test_single_gpu_and_ranks_do_not_spawn():
    run = _Run("")
    assert _spawn(["--gpus", "1"], {}, run)[0] is None
    assert _spawn(["--gpus", "4"], {"WORLD_SIZE": "4", "RANK": "2"}, run)[0] is None
    assert run.calls == []


def test_world_size_mismatch_is_a_hard_error():
    run = _Run("")
    rc, out, err = _spawn(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0"}, run)
    assert rc == 2 and "disagrees" in err and out == "" and run.calls == []
    assert _spawn(["--gpus", "1"], {"WORLD_SIZE": "8"}, run)[0] == 2


def test_plain_gpus_n_spawns_torchrun_child_and_relays_one_line():
    line = json.dumps({"metric": "scenes/sec", "value": 1.0, "n_gpus": 4})
    run = _Run("W0 noise from a rank\n" + line + "\n", rc=0, stderr="warn\n")
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    rc, out, err = _spawn(argv, {"PATH": "/usr/bin"}, run)
    assert rc == 0
    assert out.strip() == line                       # exactly rank 0's line on stdout
    assert "noise" in err and "warn" in err          # everything else goes to stderr
    (cmd, kw), = run.calls
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                       # the ranks get the caller's arguments
    assert kw["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and kw["env"]["MASTER_ADDR"] == "127.0.0.1"
    assert "WORLD_SIZE" not in kw["env"]             # torchrun sets it for the ranks


def test_child_failure_and_missing_line_are_reported():
    rc, out, _ = _spawn(["--gpus", "2"], {}, _Run("", rc=3, stderr="boom"))
    assert rc == 3 and out == ""
    rc, out, err = _spawn(["--gpus", "2"], {}, _Run("no json here\n", rc=0))
    assert rc == 1 and "expected one JSON line" in err


def test_importing_and_spawning_never_loads_torch():
    """The parent of a self-launched run must not initialise the GPU: it does not even import torch."""
    code = ("import sys, types; sys.path.insert(0, %r); import bench\n"
            "r = types.SimpleNamespace(stdout='{}\\n', stderr='', returncode=0)\n"
            "rc = bench.maybe_spawn(bench.parse(['--gpus', '2']), ['--gpus', '2'], environ={}, "
            "run=lambda *a, **k: r)\n"
            "assert rc == 0 and 'torch' not in sys.modules, sorted(m for m in sys.modules if 'torch' in m)\n"
            % ROOT)
    subprocess.run([sys.executable, "-c", code], check=True, timeout=120)


def test_traffic_is_null_with_a_reason_when_the_stamp_differs(tmp_path, monkeypatch):
    f = tmp_path / "traffic.json"
    f.write_text(json.dumps({"k": {"traffic_bytes": 5, "source": "pmc", "source_stamp": "0" * 16}}))
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(f))
    tb, src, why = bench.load_traffic("k")
    assert tb is None and "other sources" in why
    f.write_text(json.dumps({"k": {"traffic_bytes": 5, "source": "pmc", "source_stamp": bench.source_stamp()}}))
    assert bench.load_traffic("k") == (5, "pmc", None)
    assert bench.load_traffic("absent")[0] is None


def test_percentile():
    assert bench.percentile([1, 2, 3, 4, 5], 0.5) == 3
    assert abs(bench.percentile([10, 20], 0.1) - 11) < 1e-9
