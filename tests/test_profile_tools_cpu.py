"""CPU: the two scripts that turn rocprofv3 output into what bench.py's roofline is checked against (tools/trace_sum.py,
tools/pmc_summarise.py): the marker window / tail selection must pick exactly the timed steps - the untimed steps in
front of them run another workload (the loss scale settling from 65536: inf / NaN gradients through float atomics)."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, steps=6, timed=2):
    rows, t = [], 0
    for step in range(steps):
        if step == steps - timed:
            rows.append(("spin_kernel(long)", t, t + 10)); t += 10
        slow = step < steps - timed
        for k, dur in (("void (anonymous namespace)::k_bin_emit<true>((anonymous namespace)::PointSet)", 9_000_000 if slow else 5_000_000),
                       ("(anonymous namespace)::k_bin_reduce(char const*)", 2_000_000)):
            rows.append((k, t, t + dur)); t += dur
        if step == steps - 1:     # a kernel only the refresh step launches
            rows.append(("(anonymous namespace)::k_packbits(float const*)", t, t + 1000)); t += 1000
    rows.append(("spin_kernel(long)", t, t + 10))
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writerows(rows)


def test_trace_sum_window_and_tail(tmp_path):
    f = tmp_path / "x_kernel_trace.csv"
    _trace(str(f))
    run = lambda *a: subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_sum.py"), str(tmp_path), *a],
                                    capture_output=True, text=True, check=True).stdout.strip().splitlines()
    win = {l.split(",")[0]: l.split(",") for l in run("--window", "spin_kernel", "--steps", "2")[1:]}
    assert win["k_bin_emit"][1] == "2" and float(win["k_bin_emit"][-1]) == 5.0     # the timed steps only
    assert float(win["k_bin_reduce"][-1]) == 2.0 and "spin_kernel" not in " ".join(win)
    tail = {l.split(",")[0]: l.split(",") for l in run("--tail", "2/6")[1:]}
    assert tail["k_bin_emit"][1] == "2" and float(tail["k_bin_emit"][-1]) == 5.0
    assert tail["k_packbits*"][1] == "1" and tail["k_packbits*"][-1] == ""                      # not a per-step kernel
    plain = {l.split(",")[0]: l.split(",") for l in run()[1:]}
    assert plain["k_bin_emit"][1] == "6"


def test_pmc_summarise_tail_and_write_calibration(tmp_path):
    # six steps, two emit launches each; the first four steps "settle": 10 x the WRITE_SIZE of the timed ones
    rows = []
    d = 0
    for step in range(6):
        for _ in range(2):
            d += 1
            rows.append((d, "void k_bin_emit<true>(PointSet)", "WRITE_SIZE", 258.0 if step >= 4 else 2580.0))
            rows.append((d, "void k_bin_emit<true>(PointSet)", "FETCH_SIZE", 50.0))
        d += 1
        rows.append((d, "void k_mlp_bwd_g<F16, 2, 3, true>(float const*)", "FETCH_SIZE", 100.0))
        rows.append((d, "void k_mlp_bwd_g<F16, 2, 3, true>(float const*)", "WRITE_SIZE", 80.0))
    f = tmp_path / "p_counter_collection.csv"
    with open(f, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writerows(rows)
    out = tmp_path / "pmc.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summarise.py"), str(out), "c2", "1000", "--tail", "2/6",
                    str(f)], capture_output=True, text=True, check=True)
    db = json.load(open(out))
    emit = db["k_bin_emit"]["c2"]
    assert emit["WRITE_SIZE"]["launches_full"] == 4 and emit["WRITE_SIZE"]["per_launch_full"] == 258.0   # timed steps only
    assert abs(emit["write_bytes_per_launch"] - 258.0 * 1024 / 1.29) < 1e-6          # 12-byte runs: sectors -> bytes
    assert emit["fetch_bytes_per_launch_corrected_x2"] == 2 * 50.0 * 1024
    mlp = db["k_mlp_backward"]["c2"]
    assert mlp["write_bytes_per_launch"] == 80.0 * 1024 and "write_size_calibration_divisor" not in mlp
    assert abs(db["scatter_binned"]["c2"]["hbm_bytes_per_eval"] - 2 * (2 * 50.0 * 1024 + 258.0 * 1024 / 1.29) / 1000) < 1e-6
