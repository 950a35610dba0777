"""rocprofv3 --kernel-trace CSV -> one line per kernel (calls, average ms, total ms), short names.
    python tools/trace_sum.py <dir or csv> [--skip N] [--out file.csv]
--skip N drops the first N dispatches of every kernel (warm-up calls)."""
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
    if "--out" in args:
        i = args.index("--out"); out = args[i + 1]; del args[i:i + 2]
    path = args[0]
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    agg = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            m = re.search(r"(k_[a-z0-9_]+)", name)
            short = m.group(1) + ("<" + name.split("<", 1)[1].split(">(")[0][:40] + ">" if m and "<" in name.split("(")[0] else "") if m else name[:50]
            agg.setdefault(short, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    rows = []
    for k, v in agg.items():
        v = v[skip:] if len(v) > skip else v
        rows.append((sum(v), k, len(v), sum(v) / len(v), min(v), max(v)))
    rows.sort(reverse=True)
    lines = ["kernel,calls,avg_ms,total_ms,min_ms,max_ms"]
    for tot, k, n, avg, lo, hi in rows[:40]:
        lines.append(f"{k},{n},{avg:.4f},{tot:.3f},{lo:.4f},{hi:.4f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
