"""`python bench.py --gpus N` without a torch.distributed.run environment (the shape of the driver's N = 1 command) must start its
N ranks itself, as child processes, and relay rank 0's one JSON line.  CPU tests of the command, of the relay and of the failure
path (here no GPU exists, so the children fail: the parent must exit non-zero and print no line)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_launch_command():
    b = _bench()
    cmd = b.self_launch_command(4, ["--gpus", "4", "--steps", "7"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "7"]     # the script's own arguments follow it unchanged


def test_self_launch_relays_one_json_line(monkeypatch, capfd):
    b = _bench()
    line = json.dumps({"metric": "m", "n_gpus": 2, "world_size": 2})
    child = "import sys; print('noise from a rank'); print(%r); sys.stderr.write('warn\\n')" % line
    monkeypatch.setattr(b, "self_launch_command", lambda n, argv, port: [sys.executable, "-c", child])
    assert b.self_launch(2, []) == 0
    out, err = capfd.readouterr()
    assert out.strip().splitlines() == [line]
    assert "noise from a rank" in err and "warn" in err


def test_self_launch_reports_failure(monkeypatch, capfd):
    b = _bench()
    monkeypatch.setattr(b, "self_launch_command", lambda n, argv, port: [sys.executable, "-c", "import sys; print('{\"half\": 1}'); sys.exit(3)"])
    assert b.self_launch(2, []) == 3
    out, err = capfd.readouterr()
    assert out.strip() == "" and "failed (exit code 3)" in err
    monkeypatch.setattr(b, "self_launch_command", lambda n, argv, port: [sys.executable, "-c", "pass"])
    assert b.self_launch(2, []) == 1
    assert "no result line" in capfd.readouterr()[1]


def test_plain_invocation_without_gpus_fails_loudly():
    """The real thing on this GPU-less host: two children under torch.distributed.run, both fail at the first GPU call; the parent
    returns their failure and prints nothing on stdout (and it did not need a torchrun environment to get that far)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    import torch
    if torch.cuda.device_count() >= 2:
        return   # a multi-GPU box: tests/test_multi_gpu.py runs the plain invocation for real
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-sample-div", "0", "--workload", "r2"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "torch.distributed.run with 2 ranks failed" in p.stderr


def test_self_launch_retries_once_when_the_port_was_taken(monkeypatch, capfd, tmp_path):
    """The rendezvous port is chosen by binding and closing a socket; somebody else may take it before torch.distributed.run binds it.
    A child that dies with 'Address already in use' is started once more on another port; a second such failure is reported."""
    b = _bench()
    line = json.dumps({"metric": "m", "n_gpus": 2})
    marker = tmp_path / "first"
    child = ("import sys, os\n"
             "p = %r\n"
             "if not os.path.exists(p):\n"
             "    open(p, 'w').close(); sys.stderr.write('RuntimeError: The server socket has failed to listen ... EADDRINUSE Address already in use\\n'); sys.exit(1)\n"
             "print(%r)\n" % (str(marker), line))
    ports = []
    monkeypatch.setattr(b, "self_launch_command", lambda n, argv, port: (ports.append(port), [sys.executable, "-c", child])[1])
    assert b.self_launch(2, []) == 0
    out, err = capfd.readouterr()
    assert out.strip().splitlines() == [line] and len(ports) == 2 and "once more with another port" in err
    always = "import sys; sys.stderr.write('Address already in use\\n'); sys.exit(1)"
    monkeypatch.setattr(b, "self_launch_command", lambda n, argv, port: [sys.executable, "-c", always])
    assert b.self_launch(2, []) == 1
    assert "failed (exit code 1)" in capfd.readouterr()[1]


def test_self_launch_timeout_keeps_the_ranks_stderr(monkeypatch, capfd):
    """A run that hangs at N ranks: the children's stderr has been passed through as it came (a reader thread), so what they said before
    they hung is on the terminal when the time limit kills them; exit code 124."""
    b = _bench()
    child = "import sys, time; sys.stderr.write('rank 1: waiting for the rendezvous\\n'); sys.stderr.flush(); time.sleep(60)"
    monkeypatch.setattr(b, "self_launch_command", lambda n, argv, port: [sys.executable, "-c", child])
    assert b.self_launch(2, [], timeout=3) == 124
    out, err = capfd.readouterr()
    assert out.strip() == "" and "waiting for the rendezvous" in err and "did not finish within 3 s" in err
