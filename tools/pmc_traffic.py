#!/usr/bin/env python3
"""tools/pmc_traffic.py PROF_DIR ROUND -- distil rocprofv3 output (see
tools/collect_profiles.sh) into small committed summaries:

  summary/<ROUND>_kernel_stats.csv   rocprofv3 --stats table, our kernels only
  summary/<ROUND>_pmc.json           per-launch FETCH_SIZE / WRITE_SIZE averages
  summary/traffic.json               what bench.py reports as roofline.traffic

Correction (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports exactly HALF the bytes of a wide coalesced
streaming read (16 B/lane), WRITE_SIZE is exact for 16 B/lane streaming stores.
Both kernels here read and write 16 B per lane, so
    hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import glob
import json
import os
import re
import sys


def find(d, suffix):
    fs = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return fs[0] if fs else None


def main():
    prof, rnd = sys.argv[1], sys.argv[2]
    out = os.path.join(prof, "summary")
    os.makedirs(out, exist_ok=True)
    # 1. kernel stats
    ks = find(os.path.join(prof, "kt"), "kernel_stats.csv")
    rows = []
    if ks:
        with open(ks) as f:
            for r in csv.DictReader(f):
                if r["Name"].startswith("void mms::") or "mms::" in r["Name"]:
                    rows.append(r)
        with open(os.path.join(out, rnd + "_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()) if rows else ["Name"])
            w.writeheader()
            for r in rows:
                w.writerow(r)
    # 2. counters: one row per dispatch and counter
    pmc = {}
    for tag, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        cc = find(os.path.join(prof, tag), "counter_collection.csv")
        if not cc:
            continue
        with open(cc) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != cname:
                    continue
                k = r["Kernel_Name"]
                if "mms::" not in k:
                    continue
                e = pmc.setdefault(k, {}).setdefault(cname, [0.0, 0])
                e[0] += float(r["Counter_Value"])
                e[1] += 1
    summary, traffic = {}, {}
    for k, cs in pmc.items():
        fetch = cs.get("FETCH_SIZE", [0.0, 0])
        write = cs.get("WRITE_SIZE", [0.0, 0])
        f_kib = fetch[0] / fetch[1] if fetch[1] else None
        w_kib = write[0] / write[1] if write[1] else None
        hbm = None
        if f_kib is not None and w_kib is not None:
            hbm = (2.0 * f_kib + w_kib) * 1024.0
        summary[k] = {"launches_fetch_pass": fetch[1], "launches_write_pass": write[1],
                      "FETCH_SIZE_KiB_per_launch_raw": f_kib, "WRITE_SIZE_KiB_per_launch_raw": w_kib,
                      "hbm_bytes_per_launch": hbm,
                      "correction": "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of 16B/lane streaming reads)"}
        kn = k.replace("(bool)1", "true").replace("(bool)0", "false")
        # euclid_pair32_kernel / euclid_block_kernel<D4C, FWD, BWD, EXACT, WPB> / euclid_rows_wave_kernel<NIT, RW, FWD, BWD>
        m = re.search(r"euclid_(?:pair32|block)_kernel<\s*\d+,\s*(true|false),\s*(true|false)", kn) or \
            re.search(r"euclid_rows_wave_kernel<\s*\d+,\s*\d+,\s*(true|false),\s*(true|false)", kn)
        if m:
            fwd, bwd = m.group(1) == "true", m.group(2) == "true"
            ent = {"kernel": k, "hbm_bytes_per_launch": hbm}
            if fwd and bwd:
                traffic["fused"] = ent
            elif fwd:
                traffic.setdefault("layers", {})["forward"] = ent
            elif bwd:
                traffic.setdefault("layers", {})["backward"] = ent
    lay = traffic.get("layers")
    if lay and "forward" in lay and "backward" in lay and lay["forward"]["hbm_bytes_per_launch"] and lay["backward"]["hbm_bytes_per_launch"]:
        # one step of the Layer-API path = one Forward launch + one Backward launch
        lay["hbm_bytes_per_step"] = lay["forward"]["hbm_bytes_per_launch"] + lay["backward"]["hbm_bytes_per_launch"]
    json.dump(summary, open(os.path.join(out, rnd + "_pmc.json"), "w"), indent=1)
    if traffic:
        # keep entries of other paths collected in earlier runs (e.g. the fused variant)
        old = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
        try:
            for kk, vv in json.load(open(old)).items():
                traffic.setdefault(kk, vv)
        except Exception:
            pass
        traffic["_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, round " + rnd
        json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))
    for r in rows:
        print(r["Name"][:90], r["Calls"], r["AverageNs"])


if __name__ == "__main__":
    main()
