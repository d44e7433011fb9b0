"""Section timestamps of k_proj_resolve (C3 single frame).  Needs a build with -DORBGPU_PJ_TIMING (see qt_sections.py)."""
import os, re, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from orb_slam2_map_amd import workloads as W, lib as G
    r = W.c3(reps=2)
    G.lib().orbgpu_pj_dbg_dump()
    sys.exit(0)
out = subprocess.run([sys.executable, __file__, "--child"], stdout=subprocess.PIPE, text=True).stdout
rows = [(int(a), int(b)) for a, b in re.findall(r"mark (\d+) t (\d+)", out)]
# last kernel only: find last mark 8 and walk back to its first mark 1
end = max(i for i, r in enumerate(rows) if r[0] == 8)
start = max(i for i, r in enumerate(rows[:end]) if r[0] == 8) + 1 if any(r[0] == 8 for r in rows[:end]) else 0
rows = rows[start:end + 1]
t0, prev = rows[0][1], rows[0][1]
for m, t in rows:
    print("mark %2d  %8.2f us  +%7.2f" % (m, (t - t0) / 100.0, (t - prev) / 100.0))
    prev = t
