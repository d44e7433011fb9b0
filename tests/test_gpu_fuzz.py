"""A bounded, seeded slice of every randomised parity sweep (tools/fuzz_*.py: GPU path vs the oracle on random sizes,
parameters and data) inside the driver-run suite.  The long campaigns are run with tools/fuzz_campaign.sh and their
summaries are committed under profiles/ (rNN_fuzz_campaign.txt)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWEEPS = ["fuzz_extract", "fuzz_projection", "fuzz_proj_variants", "fuzz_table", "fuzz_bf", "fuzz_bow", "fuzz_m6", "fuzz_cloud"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", SWEEPS)
def test_fuzz_slice(gpu, oracle, name):
    seconds, seed = 5, 20261004
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", name + ".py"), str(seconds), str(seed)], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and tail.startswith("fuzz ok"), r.stdout[-2000:]
    counts = [int(x) for x in re.findall(r"\d+", tail.split(" in ")[0])]
    assert sum(counts) >= 3, "the slice ran almost nothing: %s" % tail
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "fuzz_slices.log"), "a") as f:
        f.write("%s seed %d: %s\n" % (name, seed, tail))
