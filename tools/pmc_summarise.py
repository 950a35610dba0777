"""Turns rocprofv3 --pmc counter_collection CSVs into profiles/pmc_r01.json entries.
    python tools/pmc_summarise.py <kernel-substring> <workload> <evals-per-launch> <FETCH csv> <WRITE csv> [out json]
FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their bytes
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so fetched bytes are doubled; WRITE_SIZE is taken as reported."""
import csv, json, os, sys


def per_launch(path, sub, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals)


def main():
    sub, workload, evals, fcsv, wcsv = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5]
    out = sys.argv[6] if len(sys.argv) > 6 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "profiles", "pmc_r01.json")
    f_kb, nf = per_launch(fcsv, sub, "FETCH_SIZE")
    w_kb, nw = per_launch(wcsv, sub, "WRITE_SIZE")
    fetched, written = 2.0 * f_kb * 1024.0, w_kb * 1024.0
    rec = {"launches_sampled": [nf, nw], "FETCH_SIZE_KB_per_launch_raw": f_kb, "WRITE_SIZE_KB_per_launch_raw": w_kb,
           "fetch_bytes_per_launch_corrected_x2": fetched, "write_bytes_per_launch": written,
           "evals_per_launch": evals, "hbm_bytes_per_eval": (fetched + written) / evals}
    db = json.load(open(out)) if os.path.exists(out) else {}
    db.setdefault(sub, {})[workload] = rec
    json.dump(db, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
