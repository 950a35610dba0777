"""rocprofv3 --kernel-trace CSV -> one line per kernel (calls, average ms, total ms), short names.
    python tools/trace_sum.py <dir or csv> [--skip N] [--tail K/T] [--out file.csv]
--skip N drops the first N dispatches of every kernel (warm-up calls).
--tail K/T keeps the LAST K/T of every kernel's dispatches, in start order: a trace of `bench.py --profile-run` holds T =
`steps_run_total` identical steps of which the last K = `steps` are the timed ones (a kernel whose dispatch count is not
a multiple of T - the grid refresh every 16th step - is listed with all its calls and marked '*'); the extra column
ms_per_step = kept total / K is what bench.py's kernels_ms_per_step must agree with.
--window MARKER --steps K keeps the dispatches that START between the first and the last dispatch whose kernel name
contains MARKER: `bench.py --profile-run` brackets its timed region with torch.cuda._sleep (kernel `spin_kernel`), so
`--window spin_kernel --steps K` lists exactly the timed steps' kernels, ms_per_step = total / K."""
import csv
import glob
import os
import re
import sys


def main():
    args = sys.argv[1:]
    skip, out = 0, None
    if "--skip" in args:
        i = args.index("--skip"); skip = int(args[i + 1]); del args[i:i + 2]
    tail = None
    if "--tail" in args:
        i = args.index("--tail"); k, t = args[i + 1].split("/"); tail = (int(k), int(t)); del args[i:i + 2]
    window, steps = None, None
    if "--window" in args:
        i = args.index("--window"); window = args[i + 1]; del args[i:i + 2]
    if "--steps" in args:
        i = args.index("--steps"); steps = int(args[i + 1]); del args[i:i + 2]
    if "--out" in args:
        i = args.index("--out"); out = args[i + 1]; del args[i:i + 2]
    path = args[0]
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    agg = {}
    lo_t, hi_t = None, None
    if window is not None:
        marks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for f in files for r in csv.DictReader(open(f))
                       if window in r["Kernel_Name"])
        if len(marks) < 2:
            raise SystemExit(f"fewer than two '{window}' dispatches in the trace")
        lo_t, hi_t = marks[0][1], marks[-1][0]
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if lo_t is not None and not (lo_t <= int(r["Start_Timestamp"]) < hi_t):
                continue
            m = re.search(r"(k_[a-z0-9_]+)", name)
            short = m.group(1) + ("<" + name.split("<", 1)[1].split(">(")[0][:40] + ">" if m and "<" in name.split("(")[0] else "") if m else name[:50]
            agg.setdefault(short, []).append((int(r["Start_Timestamp"]),
                                              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    rows = []
    for k, sv in agg.items():
        v = [d for _, d in sorted(sv)]
        v = v[skip:] if len(v) > skip else v
        mark = ""
        if tail is not None and window is None:
            if len(v) % tail[1] == 0:
                v = v[len(v) - len(v) // tail[1] * tail[0]:]
            else:
                mark = "*"
        rows.append((sum(v), k + mark, len(v), sum(v) / len(v), min(v), max(v)))
    rows.sort(reverse=True)
    per_k = steps if window is not None else (tail[0] if tail else None)
    lines = ["kernel,calls,avg_ms,total_ms,min_ms,max_ms" + (",ms_per_step" if per_k else "")]
    for tot, k, n, avg, lo, hi in rows[:40]:
        per = f",{tot / per_k:.3f}" if per_k and not k.endswith("*") else ("," if per_k else "")
        lines.append(f"{k},{n},{avg:.4f},{tot:.3f},{lo:.4f},{hi:.4f}{per}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
