"""Short render used under rocprofv3: veach-mis (or --workload), a few passes, reference-order or ordered traversal."""
import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from pt_hip import api  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="veach-mis")
ap.add_argument("--spp", type=int, default=36)
ap.add_argument("--traversal", default="exact")
ap.add_argument("--repeat", type=int, default=1)
a = ap.parse_args()
sd, label = bench.load_workload(a.workload)
scene = api.Scene(sd)
mode = api.TRAVERSE_EXACT if a.traversal == "exact" else api.TRAVERSE_FAST
for _ in range(a.repeat):
    film, st = scene.render(0, a.spp, traversal=mode)
print(label, {k: round(v, 3) for k, v in st["kernel_ms"].items()}, st["closest_rays"] + st["any_rays"], "rays", round(st["seconds"], 4), "s")
