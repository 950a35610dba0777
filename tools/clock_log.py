"""Samples the GPU's clocks and power beside another process (the scatter's box-to-box and
run-to-run spread was attributed to power management without a measurement): every `--period` seconds one line
{"t": unix time, "sclk_mhz", "mclk_mhz", "power_w", "temp_c", "busy"} from rocm-smi's JSON output, until killed.
    python tools/clock_log.py --out gpurun_out/clocks.jsonl &   # kill $! when the measured process is done"""
import argparse
import json
import re
import subprocess
import time


def sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showuse", "--json"],
                         capture_output=True, text=True, timeout=10).stdout
    card = next(iter(json.loads(out).values()))
    num = lambda v: float(re.sub(r"[^0-9.]", "", str(v)) or "nan")   # noqa: E731
    rec = {"t": time.time()}
    for k, v in card.items():
        lk = k.lower()
        if "sclk" in lk and "level" not in lk:      # "sclk clock speed:": "(2400Mhz)"; "sclk clock level:" is an index
            rec["sclk_mhz"] = num(v)
        elif "mclk" in lk and "level" not in lk:
            rec["mclk_mhz"] = num(v)
        elif "power" in lk and "w" in lk and "power_w" not in rec:
            rec["power_w"] = num(v)
        elif "temperature" in lk and "temp_c" not in rec:
            rec["temp_c"] = num(v)
        elif "gpu use" in lk:
            rec["busy"] = num(v)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/clocks.jsonl")
    ap.add_argument("--period", type=float, default=0.25)
    a = ap.parse_args()
    with open(a.out, "w") as f:
        while True:
            try:
                f.write(json.dumps(sample()) + "\n")
                f.flush()
            except Exception as e:  # noqa: BLE001 - keep sampling
                f.write(json.dumps({"t": time.time(), "error": repr(e)}) + "\n")
            time.sleep(a.period)


if __name__ == "__main__":
    main()
