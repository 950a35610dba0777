"""rocprofv3 --pmc counter_collection CSVs -> profiles/pmc_rNN.json.
    python tools/pmc_summarise.py <out json> <workload> <evals per full launch> [--tail K/T] <csv> [<csv> ...]
--tail K/T: the passes wrap `bench.py --profile-run`, T = its steps_run_total identical steps of which the LAST K are the
timed ones; only the last K/T of every family's dispatches (in dispatch order) are averaged - the untimed steps in front of
them settle the loss scale from 65536, where the binary16 gradients are full of inf / NaN and the scatter's emit hands
those to float atomics (its WRITE_SIZE is 10 x the settled steps' there).  A family whose dispatch count is not a multiple
of T (kernels the grid refresh adds every 16th step) keeps all its dispatches.
Every CSV is one --pmc pass of tools/field_bench.py (FETCH_SIZE and WRITE_SIZE need a pass each: TCC slots).  For each
kernel family the per-launch averages of every counter are recorded, plus
  hbm_bytes_per_eval = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 / evals   (gfx950: FETCH_SIZE counts wide reads at half
                       their bytes, /opt/skills/guides/MI355X_MICROARCH.md HBM section; WRITE_SIZE as reported)
  mfma_busy_frac     = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs)   (rocprofv3 sums
                       GRBM_GUI_ACTIVE over the 8 XCDs: 288.8 M for a 16.0 ms launch at ~2.1 GHz = 8 x 36 M)
  lds_conflict_frac  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
Only the LARGEST launches of a family (>= half of its maximum grid-independent duration proxy: the counter itself)
would skew nothing here: the two backward passes differ 13x in size, so families are split by launch size into
'full' (all 13 points) and 'point0'."""
import csv
import json
import os
import sys

FAMILIES = [   # (family key, substring of the kernel name); round 3's generic MLP kernels keep the family names
    ("k_grid_encode_planes_lds", "k_grid_encode_planes_lds"), ("k_grid_encode_planes", "k_grid_encode_planes"), ("k_mlp_forward", "k_mlp_forward"), ("k_mlp_forward", "k_mlp_fwd_g"),
    ("k_mlp_backward", "k_mlp_backward"), ("k_mlp_backward", "k_mlp_bwd_g"),
    ("k_bin_emit16", "k_bin_emit16"), ("k_bin_emit", "k_bin_emit("), ("k_bin_reduce", "k_bin_reduce"),
    ("k_head_forward", "k_head_forward"), ("k_head_backward", "k_head_backward"), ("k_march_train", "k_march_train"),
    ("k_composite_train_fwd", "k_composite_train_fwd"), ("k_composite_train_bwd", "k_composite_train_bwd"),
]


def family(kernel_name):
    for fam, sub in FAMILIES:
        if sub in kernel_name or (fam == "k_bin_emit" and "k_bin_emit" in kernel_name and "emit16" not in kernel_name):
            return fam
    return None


# WRITE_SIZE counts 32-byte sectors touched, not bytes (tools/write_calib.hip, profiles/write_calib_r04.txt): exact for
# coalesced streams of 4 / 12 / 16 bytes per lane, x 1.29 for 12-byte records in runs of 8 at 4-byte-aligned places (the
# emit's sorted runs), x 1.125 for 16-byte records in such runs, x 3.3 for lone 12-byte records.
WRITE_CALIBRATION = {"k_bin_emit": 1.29}


def main():
    args = sys.argv[1:]
    tail = None
    if "--tail" in args:
        i = args.index("--tail"); k, t = args[i + 1].split("/"); tail = (int(k), int(t)); del args[i:i + 2]
    out, workload, evals = args[0], args[1], float(args[2])
    acc = {}
    for path in args[3:]:
        for r in csv.DictReader(open(path)):
            fam = family(r["Kernel_Name"])
            if fam is None:
                continue
            key = (fam, r["Counter_Name"])
            acc.setdefault(key, []).append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    db = json.load(open(out)) if os.path.exists(out) else {}
    fams = sorted({k[0] for k in acc})
    for fam in fams:
        rec = {}
        for (f, counter), vals in acc.items():
            if f != fam:
                continue
            v = [x for _, x in sorted(vals)]
            if tail is not None and len(v) % tail[1] == 0:
                v = v[len(v) - len(v) // tail[1] * tail[0]:]
            big = max(v)
            full = [x for x in v if x >= 0.5 * big] or v
            small = [x for x in v if x < 0.5 * big]
            rec[counter] = {"per_launch_full": sum(full) / len(full), "launches_full": len(full)}
            if small:
                rec[counter].update({"per_launch_point0": sum(small) / len(small), "launches_point0": len(small)})
        g = lambda c: rec.get(c, {}).get("per_launch_full")
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            rec["fetch_bytes_per_launch_corrected_x2"] = 2.0 * g("FETCH_SIZE") * 1024.0
            rec["write_bytes_per_launch"] = g("WRITE_SIZE") * 1024.0 / WRITE_CALIBRATION.get(fam, 1.0)
            if fam in WRITE_CALIBRATION:
                rec["write_size_calibration_divisor"] = WRITE_CALIBRATION[fam]
            rec["evals_per_launch"] = evals
            rec["hbm_bytes_per_eval"] = (rec["fetch_bytes_per_launch_corrected_x2"] + rec["write_bytes_per_launch"]) / evals
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("GRBM_GUI_ACTIVE"):
            rec["mfma_busy_frac"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8.0 * 256 * 4)
        if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
            rec["lds_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
        db.setdefault(fam, {})[workload] = rec
    # the scatter as bench.py times it: every emit (+ emit16) and reduce launch of one full-size call (all slices);
    # the number of full-size calls in the trace = the number of full-size MLP backward launches
    parts = [db[f][workload] for f in ("k_bin_emit", "k_bin_emit16", "k_bin_reduce") if f in db and workload in db[f]]
    calls = db.get("k_mlp_backward", {}).get(workload, {}).get("FETCH_SIZE", {}).get("launches_full", 0)
    if parts and calls and all("FETCH_SIZE" in p and "WRITE_SIZE" in p for p in parts):
        total = sum((p["fetch_bytes_per_launch_corrected_x2"] * p["FETCH_SIZE"]["launches_full"] +
                     p["write_bytes_per_launch"] * p["WRITE_SIZE"]["launches_full"]) for p in parts)
        db.setdefault("scatter_binned", {})[workload] = {
            "hbm_bytes_per_eval": total / calls / evals, "calls": calls,
            "note": "emit (+ emit16) + reduce launches of one full-size scatter call (all slices), FETCH x2 + WRITE"}
    json.dump(db, open(out, "w"), indent=1)
    print(json.dumps({f: {k: v for k, v in db[f][workload].items() if not isinstance(v, dict)} for f in db
                      if workload in db[f]}, indent=1))


if __name__ == "__main__":
    main()
