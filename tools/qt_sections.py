"""Section timestamps of k_quadtree's (frame 0, level 0) workgroup.  Needs a build with the timing marks:
    ORBGPU_EXTRA_FLAGS=-DORBGPU_QT_TIMING python -c 'from orb_slam2_map_amd import build; build.build(force=True)'
    python tools/qt_sections.py [W H NFEATURES BATCH]
Marks: 0 start, 1 after step 0 (compaction), 2 after step 1 (initial nodes), 10..15 sections (2)..(7) of a pass,
3 before step 3, 4 end."""
import os
import re
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    from orb_slam2_map_amd import lib as G
    from orb_slam2_map_amd.synth import Stream
    W, H, NF, B = (int(x) for x in sys.argv[2:6])
    st = Stream(W, H, 1234)
    ext = G.ORBextractor(NF, max_batch=B)
    imgs = np.stack([st.frame(t)[0] for t in range(B)])
    for i in range(3):
        ext.extract_batch(imgs)
    G.lib().orbgpu_qt_dbg_dump()
    sys.exit(0)
args = sys.argv[1:5] if len(sys.argv) >= 5 else ["1280", "960", "2000", "1"]
child = subprocess.run([sys.executable, __file__, "--child"] + args, stdout=subprocess.PIPE, text=True)
if child.returncode != 0:
    sys.exit("qt_sections: the child process failed with exit code %d (its stderr is above)" % child.returncode)
rows = [(int(a), int(b)) for a, b in re.findall(r"mark (\d+) t (\d+)", child.stdout)]
if not rows:
    sys.exit("qt_sections: no marks -- is liborbgpu.so built with -DORBGPU_QT_TIMING?")
t0, prev = rows[0][1], rows[0][1]
for m, t in rows:
    print("mark %2d  %8.2f us  +%7.2f" % (m, (t - t0) / 100.0, (t - prev) / 100.0))
    prev = t
