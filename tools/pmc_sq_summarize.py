#!/usr/bin/env python3
"""Per-kernel SQ counters of `tools/gpu_run.sh pmcsq` (two rocprofv3 --pmc passes) -> profiles/<tag>_pmc_sq.json.

    python tools/pmc_sq_summarize.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 [tag]

Units (MI355X_MICROARCH.md, per-instruction constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves; SQ_INSTS_* count wave-instructions; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Derived per kernel:
  duration_us          from the dispatch timestamps of the counter pass (profiled clock, a few % slower than unprofiled)
  clock_GHz            GRBM_GUI_ACTIVE / 8 / duration for launches of >= 100 us.  The counter window of a dispatch is wider
                       than the dispatch (it includes the serialisation the counter pass puts around every launch), so for
                       short kernels that quotient is not a clock (round 3 printed 5.2 GHz for k_trig): kernels under
                       100 us per launch take the median clock of the long ones and say so (clock_source)
  valu_per_simd_cycle  SQ_INSTS_VALU / (1024 SIMDs x duration x clock)
  cycles_per_valu      its inverse: SIMD cycles available per VALU wave-instruction (the issue ceiling measured by
                       tools/ubench/valu_rate is 2.5 cycles for the full-rate class and 4.3 for the half-rate class)
  wave_*_frac          ACTIVE_INST_ANY, WAIT_ANY, WAIT_INST_ANY as fractions of WAVE_CYCLES
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 256 * 4


def collect(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + d)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        if "orbgpu::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("orbgpu::")[1].split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[name][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc, dur


def main():
    a1, d1 = collect(sys.argv[1])
    a2, _ = collect(sys.argv[2])
    tag = sys.argv[3] if len(sys.argv) > 3 else "r02"
    out = {}
    # clock: only launches long enough for the counter window to be the launch
    raw_clock = {}
    for name in a1:
        g = sum(a1[name].get("GRBM_GUI_ACTIVE", []))
        us = sum(d1[name].values())
        raw_clock[name] = g / 8 / us / 1e3 if g and us else 0.0
    longk = sorted(raw_clock[n] for n in a1 if raw_clock[n] and sum(d1[n].values()) / len(d1[n]) >= 100.0)
    ref_clock = longk[len(longk) // 2] if longk else 2.4
    for name in a1:
        c = {k: sum(v) for k, v in a1[name].items()}
        c.update({k: sum(v) for k, v in a2.get(name, {}).items()})
        n = len(d1[name])
        us = sum(d1[name].values())
        own = us / n >= 100.0 and raw_clock[name] > 0
        clock = raw_clock[name] if own else ref_clock  # GHz
        simd_cycles = SIMDS * us * 1e3 * clock
        k = {"launches": n, "duration_us": round(us / n, 1), "clock_GHz": round(clock, 2),
             "clock_source": "GRBM_GUI_ACTIVE of its own launches" if own else "median of the kernels >= 100 us per launch (own quotient %.2f is not a clock)" % raw_clock[name],
             "valu_insts_per_launch": int(c.get("SQ_INSTS_VALU", 0) / n), "salu_insts_per_launch": int(c.get("SQ_INSTS_SALU", 0) / n),
             "waves_per_launch": int(c.get("SQ_WAVES", 0) / n)}
        if simd_cycles and c.get("SQ_INSTS_VALU"):
            k["valu_per_simd_cycle"] = round(c["SQ_INSTS_VALU"] / simd_cycles, 3)
            k["cycles_per_valu"] = round(simd_cycles / c["SQ_INSTS_VALU"], 2)
            k["active_inst_valu_quad_frac"] = round(4 * c.get("SQ_ACTIVE_INST_VALU", 0) / simd_cycles, 3)
        wc = c.get("SQ_WAVE_CYCLES", 0)
        if wc:
            k["wave_active_frac"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
            k["wave_wait_any_frac"] = round(c.get("SQ_WAIT_ANY", 0) / wc, 3)
            k["wave_wait_inst_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
            k["vmem_rd_insts_per_launch"] = int(c.get("SQ_INSTS_VMEM_RD", 0) / len(a2[name]["SQ_WAVE_CYCLES"]))
            k["lds_insts_per_launch"] = int(c.get("SQ_INSTS_LDS", 0) / len(a2[name]["SQ_WAVE_CYCLES"]))
        out[name] = k
    res = {"note": __doc__.split("Units")[1].strip(), "how": "tools/gpu_run.sh pmcsq: bench.py --batch 256 --pool 1024 --no-overlap-match, "
           "two --pmc passes, no trace domains", "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["duration_us"] * kv[1]["launches"]))}
    json.dump(res, open(os.path.join(ROOT, "profiles", "%s_pmc_sq.json" % tag), "w"), indent=1)
    for name, k in res["kernels"].items():
        print("%-28s %7.1f us x%3d  clk %.2f  VALU/launch %11d  cyc/VALU %6s  active %5s wait %5s instwait %5s" % (
            name[:28], k["duration_us"], k["launches"], k["clock_GHz"], k["valu_insts_per_launch"], k.get("cycles_per_valu", "-"),
            k.get("wave_active_frac", "-"), k.get("wave_wait_any_frac", "-"), k.get("wave_wait_inst_frac", "-")))


if __name__ == "__main__":
    main()
