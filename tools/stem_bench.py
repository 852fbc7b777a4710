#!/usr/bin/env python3
"""The stem + max-pool stage (bench.py's stage table, 10 instrumented steps) for the library named by AIRPOSE_HIP_LIB: the
A/B and timing-only builds of stem.hip (-DSTEM_ABLATE=<bits>) are compared with it."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-sample", "0",
                      "--parity-steps", "0", "--repeat-blocks", "0", "--stage-steps", "10", "--parity-sweep", "0", "--airpose-plus", "0", "--b64", "0",
                      "--parity-pairs", "0", "--other-form", "0"] + sys.argv[1:], capture_output=True, text=True)
if not out.stdout.strip():
    sys.exit("bench.py failed: " + out.stderr.strip().splitlines()[-1][:300])
d = json.loads(out.stdout.strip().splitlines()[-1])
print("stem_maxpool %.3f ms   (step %.3f ms)" % (d["stage_ms_per_step"]["stem_maxpool"], d["ms_per_step"]))
