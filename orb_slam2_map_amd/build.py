"""Builds orb_slam2_map_amd/liborbgpu.so (HIP, gfx950 only) with hipcc.

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the build container;
the resulting .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liborbgpu.so")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")

# -ffp-contract=off: the float stages must round after every operation to match the reference's
# scalar arithmetic.  HIP's correctly-rounded f32 divide/sqrt default stays on.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + os.environ.get(
    "ORBGPU_EXTRA_FLAGS", "").split()  # e.g. -DORBGPU_QT_TIMING (tools/qt_sections.py)


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in (src,) + tuple(extra))


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = tuple(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + (
        os.path.join(ROOT, "include", "orbgpu.h"), os.path.join(ROOT, "include", "orbgpu_pattern.inc"))
    objs, jobs = [], []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _newer(src, obj, headers):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        return r.stdout

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(OUT):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
