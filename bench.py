#!/usr/bin/env python3
"""Headline benchmark: frames/s of ORB extract + brute-force Hamming match (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N=1 directly)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N>1)

A *step* = one batch of `--batch` synthetic 640x480 frames already resident in HBM: 8-level ORB
extraction at 1000 features (pyramid, blur, per-cell FAST, quadtree, orientation, rBRIEF) followed by
`batch` brute-force matches of consecutive frames (frame b vs b-1; the first against the last frame
of the previous step).  Everything runs through the C ABI of liborbgpu.so on the calling stream;
torch only owns device memory, the stream and (N>1) the RCCL process group.

Multi-GPU: frames shard by sequence, one independent sequence per rank, no data-path collective
 -> "weak" scaling; RCCL carries only the barrier and the max-time / frame-count
reductions.

Besides the contract's JSON line the script reports
  roofline     -- the dominant kernel (longest average duration, HIP events on the launch stream,
                  recorded by the library around every stage of every timed step) priced against
                  the 8 TB/s HBM peak with SURVEY.md 8d's algorithmic bytes;
  cpu_baseline -- the CPU oracle ("port" of the reference algorithm) timed on this host, rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

# SURVEY.md 8d algorithmic bytes per 640x480 / 1000-feature frame
PYR_PX = [307200, 213200, 147852, 102860, 71379, 49601, 34454, 23986]


def algorithmic_bytes(stage, n_kp, n_cand):
    tot = sum(PYR_PX)
    if stage == "pyramid":
        return sum(PYR_PX[:-1]) + sum(PYR_PX[1:])  # R levels 0..6 + W levels 1..7 = 1 569 878 B
    if stage == "fast":
        return tot + 4 * n_cand  # every level read once (score + NMS + cell logic in one kernel) + surviving keys written
    if stage == "blur":
        return 2 * tot
    if stage == "orient":
        return n_kp * (749 + 28 + 16)  # disc gather + key point + aux record
    if stage == "describe":
        return n_kp * (512 + 32)
    if stage == "quadtree":
        return 4 * n_cand * 2 + 4 * n_kp  # candidate keys compacted (R+W) + selected keys
    raise KeyError(stage)


# kernels of a stage in direct mode (aligned device input: level 0 is read from the image); the
# round-3 data path had k_border0_fast in front of the seven resize launches and one k_blur launch
STAGE_KERNELS = {"pyramid": ["k_resize_fast"] * 7, "fast": ["k_fast_detect"],
                 "quadtree": ["k_quadtree"], "orient": ["k_orient", "k_trig"],
                 "blur": ["k_blur0_direct", "k_blur"], "describe": ["k_describe"]}


PMC_FILES = ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")  # newest first (the round-1 file predates the fused FAST kernel)


def pmc_traffic(stage, frames_per_launch):
    """(HBM bytes per launch of `stage`, source) from the newest committed PMC pass under profiles/ (separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this script, gfx950 x2 FETCH correction). (None, None) if absent."""
    for name in PMC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        try:
            j = json.load(open(path))
            k = j["kernels"]
            per_frame = float(sum(k[kn]["hbm_bytes_per_frame"] for kn in STAGE_KERNELS[stage]))
            return per_frame * frames_per_launch, "SCALED to %d frames per launch from the committed per-frame counters of profiles/%s (%s); not collected in this run" % (
                frames_per_launch, name, j.get("how", "separate --pmc passes"))
        except Exception:
            continue
    return None, None


def valu_issue(stage, frames_per_launch, ms_per_launch):
    """Second bound of the dominant kernel (it is VALU-issue bound, not HBM bound): VALU wave-instructions per launch from
    the committed SQ counter pass (profiles/r04_pmc_sq.json, rocprofv3 --pmc SQ_INSTS_VALU ... at B = 256), scaled to
    this launch, against the live launch duration; the ceiling is tools/ubench/valu_rate (profiles/r02_valu_rate.txt)."""
    try:
        sq = [n for n in ("r04_pmc_sq.json", "r03_pmc_sq.json") if os.path.exists(os.path.join(ROOT, "profiles", n))][0]
        j = json.load(open(os.path.join(ROOT, "profiles", sq)))
        insts = sum(j["kernels"][kn]["valu_insts_per_launch"] for kn in j["kernels"]
                    if kn.split("<")[0] in STAGE_KERNELS[stage]) / 256.0 * frames_per_launch
        cyc = 1024 * ms_per_launch * 1e-3 * 2.4e9 / insts  # 1024 SIMDs at the nominal 2.4 GHz, as in r02_valu_rate.txt
        return {"valu_wave_insts_per_launch": int(insts), "simd_cycles_per_valu_inst": round(cyc, 2),
                "measured_issue_ceiling_cycles": {"full_rate_class": 2.5, "half_rate_class": 4.3},
                "note": "cycles a SIMD has per VALU wave-instruction of this kernel at the live launch time; the kernel's "
                        "instructions are mostly of the half-rate class (packed 16-bit extrema, v_perm), whose measured issue "
                        "cost is 4.3 cycles",
                "scaled": "instruction count SCALED from the committed B = 256 counter pass to %d frames per launch; the launch "
                          "duration is live" % frames_per_launch,
                "source": "profiles/%s (SQ_INSTS_VALU, GRBM_GUI_ACTIVE), profiles/r02_valu_rate.txt" % sq}
    except Exception:
        return None


VALU_CLASS_COST = {"full_rate": 2.5, "half_rate": 4.3}  # SIMD cycles per wave-instruction, tools/ubench/valu_rate (profiles/r02_valu_rate.txt)


def valu_issue_step(frames_per_step, ms_per_step):
    """VALU issue budget of the WHOLE step: the VALU wave-instructions of every kernel of a step (SQ
    counter pass under profiles/, per 256 frames, scaled to this step) priced at the measured issue cost of the two
    instruction classes, over the SIMD cycles of the live step (1024 SIMDs at 2.4 GHz)."""
    for name in ("r04_pmc_sq.json", "r03_pmc_sq.json"):
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", name)))
            nsteps = max(v["launches"] for kn, v in j["kernels"].items() if kn.split("<")[0] == "k_fast_detect")  # one launch per step
            per = {kn: v["valu_insts_per_launch"] * v["launches"] / nsteps / 256.0 * frames_per_step
                   for kn, v in j["kernels"].items()}
            insts = sum(per.values())
            simd_cycles = 1024 * ms_per_step * 1e-3 * 2.4e9
            top = sorted(per.items(), key=lambda kv: -kv[1])
            return {"valu_wave_insts_per_step": int(insts),
                    "share": {k: round(v / insts, 3) for k, v in top[:8]},
                    "simd_cycles_per_valu_inst": round(simd_cycles / insts, 2),
                    "issue_busy_frac": {"if_all_half_rate": round(insts * VALU_CLASS_COST["half_rate"] / simd_cycles, 3),
                                        "if_all_full_rate": round(insts * VALU_CLASS_COST["full_rate"] / simd_cycles, 3)},
                    "note": "fraction of the step's SIMD cycles the VALU issue ports are busy, bracketed by the two measured "
                            "instruction-class costs (%.1f / %.1f cycles); the instruction mix of the dominant kernel is mostly "
                            "half rate" % (VALU_CLASS_COST["full_rate"], VALU_CLASS_COST["half_rate"]),
                    "scaled": "instruction counts SCALED from the committed B = 256 counter pass (profiles/%s) to %d frames; "
                              "the step time is live" % (name, frames_per_step),
                    "source": "profiles/%s (SQ_INSTS_VALU), profiles/r02_valu_rate.txt" % name}
        except Exception:
            continue
    return None


def _cpu_worker(libpath, frames, seconds_budget, want_stages):
    """extract + BF match vs the previous frame over `frames` (cyclically) until the budget is spent.
    ctypes releases the GIL inside the C oracle, so N of these run on N cores from N threads."""
    import ctypes as C
    from oracle import oracle_py as O
    e = O.Extractor(1000, libpath=libpath)
    L = O.lib(libpath)
    t_match = 0.0
    n = 0
    prev = None
    t0 = time.perf_counter()
    while True:
        img = frames[n % len(frames)]
        k, d = e.extract(img)
        if prev is not None:
            pk, pd = prev
            out = np.zeros(max(1, len(d)), np.int32)
            tm = time.perf_counter()
            L.ora_match_bf(pd.ctypes.data_as(C.c_void_p), np.ascontiguousarray(pk["angle"]).ctypes.data_as(C.c_void_p),
                           None, len(pd), d.ctypes.data_as(C.c_void_p),
                           np.ascontiguousarray(k["angle"]).ctypes.data_as(C.c_void_p), len(d), 50, 0.7, 1,
                           out.ctypes.data_as(C.c_void_p))
            t_match += time.perf_counter() - tm
        prev = (k, d)
        n += 1
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    stages = None
    if want_stages:
        stages = {k: v / n * 1e3 for k, v in e.stage_seconds().items()}
        stages["match_bf"] = t_match / max(n - 1, 1) * 1e3
    return n, dt, stages


def cpu_baseline(frames, seconds_budget=9.0):
    """Oracle (kind 'port'): extract + BF match of consecutive frames.  `value` = one core (the reference runs
    Tracking on one thread, Examples/RGB-D/rgbd_tum.cc:77-119); `all_cores` = the same loop frame-parallel on the
    threads this process may use (capped at the GPU box's per-GPU CPU share, 16)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_py as O
    libpath = None
    try:  # -march=native build on THIS host's CPU; falls back to the portable build
        tmp = os.path.join(tempfile.gettempdir(), "liborb_oracle_native_%d.so" % os.getpid())
        O.build(out=tmp, extra_cflags="-O3 -march=native -fPIC -std=c11 -ffp-contract=off -fno-fast-math")
        libpath = tmp
    except Exception:
        O.build()
    n1, dt1, stages = _cpu_worker(libpath, frames, seconds_budget, True)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    nthreads = max(1, min(avail, 16))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(nthreads) as ex:
        res = list(ex.map(lambda i: _cpu_worker(libpath, frames[i::nthreads] if len(frames) >= 2 * nthreads else frames,
                                                seconds_budget, False), range(nthreads)))
    dt_all = time.perf_counter() - t0
    n_all = sum(r[0] for r in res)
    return {"value": n1 / dt1, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d synthetic 640x480 frames, 1000 features, extract + BF match vs previous frame, "
                      "oracle built -O3 -march=native -ffp-contract=off, host has %d cores" % (n1, os.cpu_count()),
            "stage_ms_per_frame": {k: round(v, 4) for k, v in stages.items()},
            "all_cores": {"value": n_all / dt_all, "unit": "frames/s", "cores": nthreads,
                          "sample": "%d frames over %d threads (frame-parallel, one extractor per thread), %.1f s"
                                    % (n_all, nthreads, dt_all)}}


def cpu_baseline_c3(wl, seconds_budget=12.0):
    """Oracle (kind 'port') on the C3 frame: extraction at 1280x960 / 2000 features, ComputeStereoFromRGBD + the grid, then
    Tracking::SearchLocalPoints (isInFrustum per point + SearchByProjection) against the sequence's local map, one core."""
    from oracle import oracle_py as O
    libpath = None
    try:
        tmp = os.path.join(tempfile.gettempdir(), "liborb_oracle_native_%d.so" % os.getpid())
        O.build(out=tmp, extra_cflags="-O3 -march=native -fPIC -std=c11 -ffp-contract=off -fno-fast-math")
        libpath = tmp
    except Exception:
        O.build()
    st = wl.st
    e = O.Extractor(wl.NF, libpath=libpath)
    sf = e.scale_factors()
    log_sf = float(np.log(np.float32(sf[1])))
    cam = (float(st.fx), float(st.fy), float(st.cx), float(st.cy), float(st.bf))
    depth = wl.depth[0].cpu().numpy()
    n, t_ext, t_search, nm = 0, 0.0, 0.0, 0
    t0 = time.perf_counter()
    while True:
        img = wl.host_frames[-1] if n % 2 == 0 else wl.host_frames[n % (len(wl.host_frames) - 1)]
        ta = time.perf_counter()
        k, d = e.extract(img)
        tb = time.perf_counter()
        ur, _ = O.compute_stereo_from_rgbd(k["x"], k["y"], k["x"], depth, cam[4])
        fr = O.Frame(k["x"], k["y"], k["octave"], k["angle"], ur, d, wl.W, wl.H, sf)
        nm, _, _, _ = O.search_local_points(fr, wl.Tcw, *cam, wl.table, log_sf, libpath=libpath)
        tc = time.perf_counter()
        t_ext += tb - ta
        t_search += tc - tb
        n += 1
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d synthetic 1280x960 frames, 2000 features: extract, ComputeStereoFromRGBD + grid, isInFrustum + "
                      "SearchByProjection(th=3) of %d map points (%d matches on the last frame); oracle built -O3 "
                      "-march=native -ffp-contract=off; host has %d cores" % (n, wl.M, nm, os.cpu_count()),
            "stage_ms_per_frame": {"extract": round(t_ext / n * 1e3, 3), "glue_search_local_points": round(t_search / n * 1e3, 3)}}


def main_c3_batch(args, rank, local_rank, world):
    """--workload c3_batch: north_star's 1280x960 line under --gpus N.  One set of --batch independent sequences per rank
    (frames shard by sequence: no data-path collective), the same barrier + max-time / frame-count reductions as C2."""
    import torch
    from orb_slam2_map_amd import dist as D
    from orb_slam2_map_amd import lib as G
    from orb_slam2_map_amd import workloads
    B = args.batch if args.batch != 512 else 128  # 512 is C2's default; 128 sequences x 1280x960 is the C3 batch (2.3 GB of state)
    wl = workloads.C3Batch(B, local_rank, D.sequence_seed(1234, rank))
    s_ext = wl.stream

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            D.barrier(world)
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        wl.step()
    barrier()
    cnt = wl.counts.cpu().numpy()
    n_host = wl.nout.cpu().numpy()
    assert n_host.min() > 0 and cnt[:, 0].min() > 0, "benchmark produced empty frames"
    # the timed region three times (one 41-ms sample has no spread): the contract's K steps each, barrier on
    # both sides of every repetition; `value` comes from the median repetition
    PROF_EVERY = 4
    reps = []
    for r in range(3):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            wl.ext.set_profiling(r == 1 and i % PROF_EVERY == 0)
            wl.step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            D.barrier(world)
        torch.cuda.synchronize()
        e_max, total = D.aggregate(elapsed, B * args.steps, world, device="cuda" if args.backend == "nccl" else "cpu")
        reps.append((e_max, total))
    stage_ms = wl.ext.stage_times()
    wl.ext.set_profiling(False)
    order = sorted(range(3), key=lambda r: reps[r][0])
    e_med, total = reps[order[1]]
    if rank == 0:
        N = float(n_host.mean())
        n_cand = float(sum(len(wl.ext.debug_read(G.DBG_CANDIDATES, 0, lvl)[0]) for lvl in range(8)))
        launches = {k: len(v) for k, v in STAGE_KERNELS.items()}
        dom = max(stage_ms, key=lambda k: stage_ms[k] / launches[k])
        dom_bytes = wl.stage_bytes(dom, N, n_cand) * B
        ach = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9
        alg_m2, alg_ext = wl.algorithmic_bytes(N)
        ext_ms = sum(stage_ms.values())
        out = {"metric": "frames/sec ORB extract + SearchByProjection (1280x960, 2000 feat, ~10 k local MapPoints)",
               "value": total / e_med, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
               "ms_per_step": e_med / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u8", "data": "synthetic",
               "timed_region_repeats": {"frames_per_s": [round(t / e, 1) for e, t in reps],
                                        "spread_pct": round(100 * (max(e for e, _ in reps) - min(e for e, _ in reps)) / e_med, 2),
                                        "value_is": "median of 3 repetitions of the K-step region"},
               "config": {"workload": "C3 (BASELINE.json configs[2]) in throughput mode: synthetic 1280x960 RGB-D, 2000 features, 8 "
                                      "levels; per step and sequence one frame: extract + frame glue (ComputeStereoFromRGBD, grid) "
                                      "+ Tracking::SearchLocalPoints (isInFrustum + SearchByProjection th=3) against the sequence's "
                                      "%d local map points" % wl.M,
                          "sequences_per_gpu": B, "frames_per_step_per_gpu": B, "sequences": B * world,
                          "parallelism": "%d sequences per GPU, sequences never cross GPUs" % B, "schedule": "serial, 1 stream"},
               "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach / HBM_PEAK_GBS, "traffic": None,
                            "traffic_source": "no PMC pass at 1280x960 (the C2 passes under profiles/ are per 640x480 frame)",
                            "algorithmic_bytes": dom_bytes, "ms_per_launch": stage_ms[dom], "frames_per_launch": B},
               "stages": {k: {"ms": round(v, 4), "GB/s": round(wl.stage_bytes(k, N, n_cand) * B / (max(v, 1e-6) * 1e-3) / 1e9, 1)}
                          for k, v in stage_ms.items()},
               "extract_ms_per_step": round(ext_ms, 4),
               "extract_roofline": {"achieved": alg_ext / (ext_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": alg_ext / (ext_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg_ext},
               "search_algorithmic_bytes_per_step": alg_m2,
               "keypoints_per_frame": N, "matches_per_frame": float(cnt[:, 0].mean()), "map_points": wl.M}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_c3(wl)
        print(json.dumps(out), flush=True)
    D.finalize(world)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512,
                    help="frames per step and per GPU (225 k frames/s at 512, 216 k at 256, 197 k at 128: launch tails amortise)")
    ap.add_argument("--pool", type=int, default=1024, help="distinct resident frames cycled through (> Infinity Cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-self-check", action="store_true",
                    help="skip the serial re-run that checks the timed schedule (profiling passes: keeps every launch of a "
                         "kernel the same size)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the C3 / C4 measurements and the copy-kernel roofline reported next to the contract line")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--rehearse-on-device0", action="store_true",
                    help="N>1 rehearsal on a 1-GPU box: every rank uses device 0 (use with --backend gloo)")
    ap.add_argument("--match-after", default="fast", choices=["start", "pyramid", "fast", "quadtree", "orient", "blur"],
                    help="overlapped schedule: the matcher of the previous step starts when the extraction has passed this stage "
                         "(with the blur next to the FAST pass on its own stream, 'fast' measured 277.7 k against 275.6 k frames/s "
                         "for 'quadtree' in three A/B pairs on one box; round 3, blur in line: 'quadtree')")
    ap.add_argument("--parts", type=int, default=1,
                    help="extraction of a step as this many staggered sub-batches on streams of their own (orbgpu_pipeline); "
                         "1 = one plain batched call (default: next to the matcher more parts gain 0-6 % and not reliably, "
                         "DESIGN.md section 5).  Only with the overlapped matcher schedule")
    ap.add_argument("--match-part", type=int, default=-1, help="part whose --match-after stage starts the previous step's matcher")
    ap.add_argument("--no-overlap-match", dest="overlap_match", action="store_false",
                    help="serialise the matcher behind the extraction (default: the matcher of step i runs on a "
                         "second stream next to the extraction of step i+1)")
    ap.add_argument("--schedule", default="best", choices=["best", "overlap", "serial"],
                    help="schedule of the timed region: best (default) = matcher of step i-1 on its own stream next to extraction "
                         "i + the blur on a stream of the handle's own; overlap = the matcher stream only (round-3 default); "
                         "serial = one stream.  Per-kernel times always come from a serial calibration pass in front of it")
    ap.add_argument("--concurrent-blur", action="store_true",
                    help="with --schedule overlap: run the blur of every extraction on a stream of the handle's own next to FAST / "
                         "quadtree (orbgpu_extractor_set_concurrent_blur); implied by --schedule best")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3_batch"],
                    help="c2 (default, the contract line): 640x480 / 1000 features, extract + BF match.  c3_batch: the 1280x960 "
                         "stream of BASELINE.json configs[2] in throughput mode -- --batch independent sequences per GPU, one "
                         "frame each per step: extract + frame glue + SearchLocalPoints against ~10 k map points per sequence")
    ap.add_argument("--force-group", action="store_true",
                    help="build the process group for a single rank too: barrier and reductions then run through the "
                         "backend (RCCL) exactly as on the multi-GPU node -- what a one-GPU box can execute of that path")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start one rank per GPU as a CHILD process group before this process
        # imports torch / loads liborbgpu.so / touches HIP (never exec from a process that saw the GPU).
        # Rank 0 of the child prints the JSON line on the inherited stdout; the child's exit code is ours.
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch
    import torch.distributed as dist
    from orb_slam2_map_amd import dist as D
    from orb_slam2_map_amd import lib as G
    from orb_slam2_map_amd.synth import Stream

    rank, local_rank, world = D.env_rank()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    G.lib()  # fail loudly if the HIP extension is missing
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: bench.py measures the HIP path only (no CPU fallback)")
    if args.rehearse_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # RCCL; carries the barrier and two scalar reductions only (--force-group: through the backend even for one rank)
    D.init(args.backend, rank, world, force_group=args.force_group)

    if args.workload == "c3_batch":
        return main_c3_batch(args, rank, local_rank, world)

    W, H, NFEAT = 640, 480, 1000
    B, POOL = args.batch, max(args.pool, args.batch)
    POOL = (POOL // B) * B
    # schedule of the TIMED region.  "best" = the fastest one measured: the matcher of step i-1 on a
    # stream of its own next to extraction i, and the blur of every extraction on a stream of the handle's own next to
    # FAST / quadtree.  Per-kernel times and the roofline come from a serial CALIBRATION pass in front of it, where no
    # kernel shares the device, so the per-kernel figures and the headline no longer constrain each other.
    sched = args.schedule if args.overlap_match else "serial"
    overlap = sched != "serial"
    cblur = overlap and (sched == "best" or args.concurrent_blur)
    # one independent sequence per rank
    st = Stream(W, H, D.sequence_seed(1234, rank))
    host_pool = np.stack([st.frame(t)[0] for t in range(POOL)])
    frames = torch.from_numpy(host_pool).cuda(local_rank)

    ext = G.ORBextractor(NFEAT, max_batch=B, device_id=local_rank)
    cap = ext.max_keypoints(W, H)
    P = args.parts if overlap else 1
    pl = G.ExtractorPipeline(NFEAT, max_batch=B, parts=P, device_id=local_rank) if P > 1 else None
    parts = pl.parts if pl else [ext]  # the handles that run the timed extraction
    matcher = G.BatchMatcher(B, cap, device_id=local_rank)
    match_b = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    nmatch = torch.zeros(B, dtype=torch.int32, device="cuda")
    KP, DS = cap * 28, cap * 32
    # Output sets (slot 0 of a set carries the previous step's last frame; this step's frames go to slots 1..B).
    # Extraction runs on the current stream; in the overlapped schedules the matcher of step i runs on a second stream
    # next to the extraction of step i+1 (no data dependency between them).
    nsets = (3 if args.match_after != "start" else 2) if overlap else 1  # a deferred matcher holds its set one step longer
    kps = [torch.zeros((B + 1, cap, 7), dtype=torch.float32, device="cuda") for _ in range(nsets)]
    desc = [torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device="cuda") for _ in range(nsets)]
    nout = [torch.zeros(B + 1, dtype=torch.int32, device="cuda") for _ in range(nsets)]
    s_ext = torch.cuda.current_stream()
    s_match = torch.cuda.Stream() if overlap else s_ext
    ev_ext = [torch.cuda.Event() for _ in range(nsets)]
    ev_match = [torch.cuda.Event() for _ in range(nsets)]
    ev_copy = [torch.cuda.Event() for _ in range(nsets)]  # slot B of set k has been carried over to the other set

    def run_match(k, stream):
        matcher.match(B, cap, desc[k].data_ptr(), kps[k].data_ptr() + 12, None, nout[k].data_ptr(),
                      desc[k].data_ptr() + DS, kps[k].data_ptr() + KP + 12, nout[k].data_ptr() + 4, 28, 50, 0.7,
                      True, match_b.data_ptr(), nmatch.data_ptr(), stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            D.barrier(world)
        torch.cuda.synchronize()

    # ---- calibration: the serial schedule (one stream, plain handle), HIP events at every stage boundary of every step
    #      (recorded by the library on the launch stream) and around the matcher.  Outside the timed region.
    NCAL = 12
    cal_pairs = []
    for i in range(NCAL + 2):
        src = frames[(i * B) % POOL:(i * B) % POOL + B]
        ext.set_profiling(i >= 2)
        ext.extract_batch_device(src.data_ptr(), B, W, H, W, W * H, kps[0].data_ptr() + KP, desc[0].data_ptr() + DS, cap,
                                 nout[0].data_ptr() + 4, s_ext.cuda_stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s_ext)
        run_match(0, s_ext)
        b.record(s_ext)
        if i >= 2:
            cal_pairs.append((a, b))
        kps[0][0].copy_(kps[0][B], non_blocking=True)
        desc[0][0].copy_(desc[0][B], non_blocking=True)
        nout[0][0:1].copy_(nout[0][B:B + 1], non_blocking=True)
    torch.cuda.synchronize()
    stage_ms = ext.stage_times()  # per stage: the launches that cover the B frames of a step, nothing else on the device
    ext.set_profiling(False)
    match_ms = sum(a.elapsed_time(b) for a, b in cal_pairs) / len(cal_pairs)
    match_how = "k_bf_topk + k_bf_resolve of one step, HIP events on the launch stream, serial calibration pass (%d steps)" % NCAL

    if cblur:
        for e in parts:
            e.set_concurrent_blur(True)
    # Overlapped schedule: the matcher of step i-1 is enqueued on its own stream while extraction i runs, and starts when
    # that extraction has passed the stage named by --match-after (the library records ev_mid there).  The brute-force
    # matcher is matrix-core / LDS work; next to the FAST pass (VALU-bound) or the pyramid (HBM-bound) it takes more from
    # the extraction than next to the quadtree, whose long level-0 workgroups leave most of the device idle.
    ev_mid = torch.cuda.Event()
    if overlap and args.match_after != "start":
        ev_mid.record(s_ext)  # torch creates the hipEvent_t on first use
        assert ev_mid.cuda_event, "no event handle"
        parts[args.match_part % len(parts)].set_stage_signal(args.match_after, ev_mid.cuda_event)
    pending = [None]
    # pipeline mode: one event gates a call (the matcher that last read the output set has finished and its last frame
    # has been carried over), one event says the call's parts are all done
    s_gate = torch.cuda.Stream() if pl else None
    ev_gate = [torch.cuda.Event() for _ in range(nsets)]
    for e in ev_gate + ev_ext:
        e.record(s_ext)  # create the handles

    def flush_match():
        if pending[0] is not None:
            with torch.cuda.stream(s_match):
                run_match(pending[0], s_match)
                ev_match[pending[0]].record(s_match)
            pending[0] = None

    def step(i):
        k = i % nsets
        p = (i - 1) % nsets  # set holding the previous step's last frame (its own slot B when nsets == 1)
        src = frames[(i * B) % POOL:(i * B) % POOL + B]
        if overlap:
            # carry-over first: set p's slot B (extraction i-1, already ordered before on s_match) -> set k's slot 0,
            # which extraction i never writes.  Extraction i+1 overwrites set p and waits for ev_copy[p].
            if i > 0:
                with torch.cuda.stream(s_match):
                    kps[k][0].copy_(kps[p][B], non_blocking=True)
                    desc[k][0].copy_(desc[p][B], non_blocking=True)
                    nout[k][0:1].copy_(nout[p][B:B + 1], non_blocking=True)
                    ev_copy[p].record(s_match)
            s_ext.wait_event(ev_match[k])  # the matcher that last read this set has finished
            s_ext.wait_event(ev_copy[k])   # ... and its last frame has been copied out
        if pl:
            s_gate.wait_event(ev_match[k])
            s_gate.wait_event(ev_copy[k])
            ev_gate[k].record(s_gate)
            pl.extract_batch_device(src.data_ptr(), B, W, H, W, W * H, kps[k].data_ptr() + KP, desc[k].data_ptr() + DS,
                                    cap, nout[k].data_ptr() + 4, ev_gate[k].cuda_event, ev_ext[k].cuda_event)
        else:
            ext.extract_batch_device(src.data_ptr(), B, W, H, W, W * H, kps[k].data_ptr() + KP, desc[k].data_ptr() + DS,
                                     cap, nout[k].data_ptr() + 4, s_ext.cuda_stream)
        if overlap:
            if not pl:
                ev_ext[k].record(s_ext)
            if pending[0] is not None and args.match_after != "start":
                s_match.wait_event(ev_mid)  # extraction i has passed the chosen stage: now match step i-1
            flush_match()
            s_match.wait_event(ev_ext[k])
            if args.match_after == "start":
                pending[0] = k
                flush_match()  # the original schedule: match step i as soon as its extraction is done
            else:
                pending[0] = k
        else:
            run_match(k, s_match)
            kps[0][0].copy_(kps[0][B], non_blocking=True)
            desc[0][0].copy_(desc[0][B], non_blocking=True)
            nout[0][0:1].copy_(nout[0][B:B + 1], non_blocking=True)

    def drain():
        flush_match()
        if pl:
            pl.wait(s_ext.cuda_stream)
        if overlap:
            s_ext.wait_stream(s_match)

    it = [0]  # steps issued so far: the output sets rotate with it across warm-up and the repetitions

    def run_steps(n):
        for _ in range(n):
            step(it[0])
            it[0] += 1

    run_steps(args.warmup)
    drain()
    # ---- the timed region, NREP times: EXACTLY K steps each, barrier + synchronize on both sides, the last step's
    #      matcher inside; max over ranks per repetition; `value` is the MEDIAN repetition
    NREP = 3
    reps = []
    for _ in range(NREP):
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        drain()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            D.barrier(world)
        torch.cuda.synchronize()
        reps.append(D.aggregate(elapsed, B * args.steps, world, device="cuda" if args.backend == "nccl" else "cpu"))
    order = sorted(range(NREP), key=lambda r: reps[r][0])
    elapsed_max, total_frames = reps[order[NREP // 2]]

    # sanity of the measured work (rank-local): every frame produced key points and matches
    n_host = nout[0].cpu().numpy()
    nm_host = nmatch.cpu().numpy()
    sweeps = matcher.last_sweeps(B)
    assert n_host[1:].min() > 0 and nm_host.min() >= 0, "benchmark produced empty frames"
    if args.steps > 0 and it[0] > 1 and not args.no_self_check:
        # the pipelined schedule must give what a serial schedule gives: redo the last step from the frames, one
        # stream, fresh buffers (previous step's last frame -> slot 0, this step's frames -> slots 1..B), and compare
        last_i = it[0] - 1
        c_kps, c_desc, c_n = torch.zeros_like(kps[0]), torch.zeros_like(desc[0]), torch.zeros_like(nout[0])
        chk_b, chk_n = torch.zeros_like(match_b), torch.zeros_like(nmatch)
        s0 = torch.cuda.current_stream().cuda_stream
        prev_last = ((last_i - 1) * B) % POOL + B - 1
        ext.extract_batch_device(frames[prev_last:prev_last + 1].data_ptr(), 1, W, H, W, W * H, c_kps.data_ptr(),
                                 c_desc.data_ptr(), cap, c_n.data_ptr(), s0)
        cur = (last_i * B) % POOL
        ext.extract_batch_device(frames[cur:cur + B].data_ptr(), B, W, H, W, W * H, c_kps.data_ptr() + KP,
                                 c_desc.data_ptr() + DS, cap, c_n.data_ptr() + 4, s0)
        matcher.match(B, cap, c_desc.data_ptr(), c_kps.data_ptr() + 12, None, c_n.data_ptr(), c_desc.data_ptr() + DS,
                      c_kps.data_ptr() + KP + 12, c_n.data_ptr() + 4, 28, 50, 0.7, True, chk_b.data_ptr(),
                      chk_n.data_ptr(), s0)
        torch.cuda.synchronize()
        assert torch.equal(chk_n, nmatch) and torch.equal(chk_b, match_b), "the timed schedule changed the matches"

    n_kp = float(n_host[1:].mean())

    # What the other schedules give on this box (informational, after the contract's timed region): the same loop without
    # the concurrent blur (the r03 default); the serial schedule is the calibration pass.
    others = None
    if world == 1 and not args.no_secondary and P == 1 and cblur and args.steps >= 10:
        ext.set_concurrent_blur(False)
        nup = 30
        run_steps(3)
        drain()
        torch.cuda.synchronize()
        tu = time.perf_counter()
        run_steps(nup)
        drain()
        torch.cuda.synchronize()
        tu = time.perf_counter() - tu
        ext.set_concurrent_blur(True)
        others = {"overlapped_matcher_only": {"frames_per_s": B * nup / tu, "ms_per_step": tu / nup * 1e3, "steps": nup,
                                              "note": "bench.py --schedule overlap: matcher on its own stream, blur in line (the "
                                                      "round-3 default)"},
                  "serial_calibration": {"ms_per_step": sum(stage_ms.values()) + match_ms,
                                         "frames_per_s": B / ((sum(stage_ms.values()) + match_ms) * 1e-3),
                                         "note": "one stream, HIP events at every stage boundary (they cost the stream a bubble "
                                                 "each): bench.py --schedule serial times it without the events"}}

    if rank == 0:
        # dominant kernel of the extraction pipeline + its roofline fraction
        # FAST survivors handed to the quadtree (first frame of the self-check batch); prices the NMS / quadtree rows
        n_cand = 13700.0 if args.no_self_check else \
            float(sum(len(ext.debug_read(G.DBG_CANDIDATES, 0, lvl)[0]) for lvl in range(8)))
        # dominant KERNEL = longest average launch; the pyramid stage is a chain of 8 launches, orient is 2
        launches = {k: len(v) for k, v in STAGE_KERNELS.items()}
        dom = max(stage_ms, key=lambda k: stage_ms[k] / launches[k])
        ach = algorithmic_bytes(dom, n_kp, n_cand) * B / (stage_ms[dom] * 1e-3) / 1e9
        per_stage = {k: {"ms": round(v, 4),
                         "GB/s": round(algorithmic_bytes(k, n_kp, n_cand) * B / (max(v, 1e-6) * 1e-3) / 1e9, 1)}
                     for k, v in stage_ms.items()}
        # next to the algorithmic rate: what the stage really moved through HBM (committed PMC bytes per frame, scaled to
        # this launch) over its live duration -- the gather stages read whole planes line by line, far more than N x 1321 B
        for k in per_stage:
            tb, _ = pmc_traffic(k, B)
            if tb:
                per_stage[k]["hbm_traffic_GB/s_scaled_pmc"] = round(tb / (max(stage_ms[k], 1e-6) * 1e-3) / 1e9, 1)
        traffic, traffic_src = pmc_traffic(dom, B)
        ext_bytes = sum(algorithmic_bytes(k, n_kp, n_cand) for k in ("pyramid", "fast", "blur", "orient", "describe"))
        ext_ms = sum(stage_ms.values())  # serial calibration: the six stages follow each other
        ext_ach = ext_bytes * B / (ext_ms * 1e-3) / 1e9
        step_ms = elapsed_max / args.steps * 1e3
        out = {
            "metric": "frames/sec ORB extract+match (640x480, 1000 feat)",
            "value": total_frames / elapsed_max,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "timed_region_repeats": {"n": NREP, "frames_per_s": [round(t / e, 1) for e, t in reps],
                                     "ms_per_step": [round(e / args.steps * 1e3, 4) for e, _ in reps],
                                     "spread_pct": round(100 * (max(e for e, _ in reps) - min(e for e, _ in reps)) / elapsed_max, 2),
                                     "value_is": "the median repetition; every repetition is K steps between barrier + synchronize"},
            "config": {"workload": "C2: synthetic 640x480 RGB-D stream, 1000 features, 8 levels, extract + BF-Hamming "
                                   "match of consecutive frames", "frames_per_step_per_gpu": B,
                       "resident_frame_pool": POOL, "sequences": world, "parallelism": "1 sequence per GPU",
                       "schedule": (("extraction as %d staggered sub-batches on %d streams (orbgpu_pipeline: a part starts when the "
                                     "previous one has passed its pyramid stage); " % (P, P) if pl else "") +
                                    ("matcher of step i-1 on a stream of its own, started when extraction i has passed its '%s' stage"
                                     % args.match_after if args.match_after != "start" else
                                     "matcher of step i overlapped with extraction of step i+1 (2 streams)") +
                                    ("; blur of every extraction on a stream of the handle's own next to FAST / quadtree" if cblur else ""))
                       if overlap else "serial, 1 stream", "schedule_name": sched, "extraction_parts": P},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes": algorithmic_bytes(dom, n_kp, n_cand) * B,
                         "ms_per_launch": stage_ms[dom], "frames_per_launch": B, "launches": 1,
                         "measured_in": "serial calibration pass of this run (%d steps, nothing else on the device), HIP events "
                                        "on the launch stream" % NCAL},
            "stages": dict(per_stage, **{"match": {"ms": round(match_ms, 4), "how": match_how}}),
            "stages_measured_in": "serial calibration pass (not the timed region, whose schedule overlaps kernels)",
            "valu_issue": valu_issue(dom, B, stage_ms[dom]),
            "valu_issue_step": valu_issue_step(B, step_ms),
            "extract_ms_per_step": round(ext_ms, 4),
            "extract_roofline": {"achieved": ext_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ext_ach / HBM_PEAK_GBS,
                                 "algorithmic_bytes": ext_bytes * B,
                                 "note": "all six extraction stages of one step (serial calibration) against SURVEY.md 8d's 5.74 MB/frame"},
            "fast_candidates_per_frame": n_cand,
            "match_ms_per_step": round(match_ms, 4),
            "keypoints_per_frame": n_kp,
            "matches_per_frame": float(nm_host.mean()),
            "bf_sweeps_max": int(sweeps.max()),
        }
        if not args.no_secondary and world == 1:
            # free the C2 working set first, then: the practical roofline (plain copy kernel over 2 x 1 GiB, far past
            # the Infinity Cache) and the other single-GPU configurations of BASELINE.json
            del frames, kps, desc, match_b
            torch.cuda.empty_cache()
            from orb_slam2_map_amd import workloads
            copy_gbs = G.measure_copy_bandwidth(1 << 30, 10, local_rank)
            out["roofline"]["practical_peak"] = {"copy_kernel_GB/s": copy_gbs, "frac_of_practical": ach / copy_gbs,
                                                 "what": "16 B/lane device-to-device copy, 1 GiB, read + write bytes"}
            out["secondary"] = {}
            if others:
                out["secondary"]["c2_schedules"] = others
            for name, fn in (("c3", workloads.c3), ("c3_batch", workloads.c3_batch), ("c4", workloads.c4),
                             ("c2_fast_early_out", workloads.c2_fast_early_out)):
                try:
                    r = fn(device_id=local_rank)
                    r.pop("per_keyframe", None)
                    out["secondary"][name] = r
                except Exception as ex:  # the contract line must survive a secondary failure; say so loudly
                    out["secondary"][name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline([host_pool[i] for i in range(min(POOL, 400))])
        print(json.dumps(out), flush=True)
    D.finalize(world)

if __name__ == "__main__":
    main()
