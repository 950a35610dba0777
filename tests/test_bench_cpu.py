"""bench.py's launcher contract (no GPU needed): a world size that differs from --gpus is refused, and --gpus N outside a
launcher re-executes under torch.distributed.run with one rank per GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE is 3" in r.stderr


def test_gpus_n_becomes_the_launcher(monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    try:
        bench.spawn_ranks(4)
        raise AssertionError("spawn_ranks must exit with the launcher's return code")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and os.path.samefile(cmd[-5], BENCH)
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_clock_sampler_without_a_card_samples_nothing():
    """bench.ClockSampler reads sysfs of the card whose PCI address torch reports; with no such card (this container) it
    starts no thread and reports None - it never falls back to forking a child from the benchmark process."""
    sys.path.insert(0, ROOT)
    import importlib
    import torch
    bench = importlib.import_module("bench")
    assert bench.ClockSampler.pci_address_of(torch.device("cpu")) is None
    c = bench.ClockSampler(None).start()
    assert c._thread is None
    c.stop()
    assert c.summary() is None
    c = bench.ClockSampler("0000:ff:1f.0").start()      # an address no card has
    c.stop()
    assert c.summary() is None
