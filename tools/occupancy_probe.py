"""Developer tool: is k_mesh_raster_shade latency-bound or throughput-bound?  Pads its dynamic LDS (diagnostic build,
MVR_LDS_PAD) so that only 3 / 2 / 1 workgroups fit per CU and reports the kernel time at each occupancy."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvtn_amd import build as b  # noqa: E402

diag = os.path.join(ROOT, "mvtn_amd", "libmvr_hip_diag.so")
b.build_hip(out=diag, extra_flags=["-DMVR_DIAG"])
for pad in (0, 8 * 1024, 20 * 1024, 50 * 1024, 100 * 1024):
    env = dict(os.environ, MVR_LIB=diag, MVR_LDS_PAD=str(pad))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "50", "--warmup", "10"],
                         env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    import json
    d = json.loads(out)
    print("LDS pad %6d B -> kernel %.1f us, step %.1f us" % (pad, d["roofline"]["kernel_ms"] * 1e3, d["ms_per_step"] * 1e3))
