#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of `tools/gpu_run.sh pmc` into profiles/r01_pmc_traffic.json.

    python tools/pmc_summarize.py gpurun_out/pmc_fetch gpurun_out/pmc_write [frames_per_launch] [tag] [pool]

Each pass wrote one *counter_collection.csv with a row per (dispatch, counter).  FETCH_SIZE / WRITE_SIZE are in KB.
gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE under-reports coalesced streaming reads
by 2x -- calibrated on the level-0 border kernel (k_border0 / k_border0_fast), which reads each source byte once -- so HBM bytes = 2 * FETCH + WRITE.
Also copies the two CSVs (orbgpu kernels only) next to the JSON.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + d)
    f = max(files, key=os.path.getmtime)
    acc, rows = {}, []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or "orbgpu::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("orbgpu::")[1].split("(")[0].split("<")[0]
        s = acc.setdefault(name, [0, 0.0])
        s[0] += 1
        s[1] += float(r["Counter_Value"])
        rows.append(r)
    return acc, rows, f


def main():
    dfetch, dwrite = sys.argv[1], sys.argv[2]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    tag = sys.argv[4] if len(sys.argv) > 4 else "r01"
    pool = int(sys.argv[5]) if len(sys.argv) > 5 else 256
    fe, frows, _ = collect(dfetch, "FETCH_SIZE")
    wr, wrows, _ = collect(dwrite, "WRITE_SIZE")
    kernels = {}
    for name in fe:
        nf, sf = fe[name]
        nw, sw = wr.get(name, [1, 0.0])
        fkb, wkb = sf / nf, sw / max(nw, 1)
        kernels[name] = {"launches": nf, "FETCH_SIZE_KB": round(fkb, 1), "WRITE_SIZE_KB": round(wkb, 1),
                         "hbm_bytes_per_frame": int(round((2 * fkb + wkb) * 1024 / frames))}
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/gpu_run.sh pmc), bench.py "
                   "--batch %d --pool %d; values are KB per launch (%d frames), averaged over launches. gfx950 "
                   "correction (MI355X_MICROARCH.md): FETCH_SIZE under-reports coalesced streaming reads by 2x; "
                   "calibrated on the level-0 border kernel (k_border0 / k_border0_fast), which reads each of the %d*307200 source bytes once. "
                   "hbm_bytes_per_frame = (2*FETCH + WRITE)*1024/%d. Made by tools/pmc_summarize.py." % (
                       frames, pool, frames, frames, frames),
           "how": "separate --pmc passes at B=%d, pool %d" % (frames, pool),
           "frames_per_launch": frames, "kernels": kernels}
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % tag), "w"), indent=1)
    for rows, nm in ((frows, "%s_pmc_fetch_size_b%d.csv" % (tag, frames)), (wrows, "%s_pmc_write_size_b%d.csv" % (tag, frames))):
        with open(os.path.join(ROOT, "profiles", nm), "w", newline="") as fo:
            w = csv.DictWriter(fo, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    for k, v in kernels.items():
        print("%-18s %4d launches  fetch %10.1f KB  write %10.1f KB  -> %9d B/frame" % (
            k, v["launches"], v["FETCH_SIZE_KB"], v["WRITE_SIZE_KB"], v["hbm_bytes_per_frame"]))


if __name__ == "__main__":
    main()
