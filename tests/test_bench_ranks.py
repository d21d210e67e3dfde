"""The multi-rank path of bench.py and the order in which a process may load torch and the backend.

bench.py --gpus N is what the scaling measurement runs.  Launched bare (no WORLD_SIZE) it has to start its own ranks as
fresh child processes; with PT_BENCH_REHEARSE=1 the ranks share whatever GPUs there are and reduce through gloo and host
memory, which exercises the whole code path (tree built once and shared through /dev/shm, sample split, film reduce,
max-over-ranks timing, the JSON line) on a one-GPU box.  Its line is marked as a rehearsal and carries no value.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _bench(args, env_extra):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("split", ["spp", "tile"])
def test_bench_started_bare_with_two_ranks_spawns_them_and_matches_one_rank(split):
    common = ["--steps", "1", "--warmup", "0", "--spp", "8", "--workload", "cornell-box-synth", "--no-cpu-baseline"]
    one = _bench(["--gpus", "1"] + common, {})
    two = _bench(["--gpus", "2", "--split", split] + common, {"PT_BENCH_REHEARSE": "1"})
    assert one["n_gpus"] == 1 and one["value"] > 0 and "rehearsal" not in one
    assert two["n_gpus"] == 2 and two["rehearsal"] is True and two["value"] is None and two["metric"].startswith("REHEARSAL")
    assert two["scaling"] == "strong"
    # the two ranks traced the same rays as the single one: rays per path and paths per second x seconds agree
    rays_one = one["rays_per_path"] * one["samples_per_s"] * one["ms_per_step"]
    rays_two = two["rays_per_path"] * two["samples_per_s"] * two["ms_per_step"]
    assert abs(rays_one - rays_two) <= 1e-6 * rays_one
    assert two["roofline"]["frac"] > 0 and two["roofline"]["peak"] > 0


_ORDER = """
import sys
sys.path.insert(0, %r)
first = sys.argv[1]
if first == "torch":
    import torch
    assert torch.cuda.is_available()
    x = torch.zeros(4, device="cuda:0") + 1
from pt_hip import api, synth
scene = api.Scene(synth.cornell_box(32, 32), device=0)
film, st = scene.render(0, 2, traversal=api.TRAVERSE_FAST)
assert st["closest_rays"] > 0
if first == "backend":
    import torch
    assert torch.cuda.is_available(), "torch lost the GPU after the backend had loaded its runtime"
    x = torch.zeros(4, device="cuda:0") + 1
assert float(x.sum()) == 4.0
film2, st2 = scene.render(0, 2, traversal=api.TRAVERSE_FAST)
assert st2["closest_rays"] == st["closest_rays"]
print("ok", first)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("first", ["torch", "backend"])
def test_torch_and_the_backend_share_one_hip_runtime_in_either_load_order(first):
    """PyTorch-ROCm ships its own copy of the HIP runtime; a process in which both that copy and the system's initialise
    loses the GPU in the second.  pt_hip.api loads torch's copy itself when torch is installed, so the order of the two
    imports must not matter (bench.py imports torch first, the tests the backend)."""
    src = _ORDER % os.path.join(ROOT, "path-tracer_amd", "python")
    r = subprocess.run([sys.executable, "-c", src, first], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and ("ok " + first) in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bare_multi_gpu_launch_builds_the_torchrun_command(monkeypatch):
    """CPU-side half of the same: without WORLD_SIZE, --gpus 4 must hand over to torch.distributed.run on 127.0.0.1 with the
    caller's own arguments, and exit with the launcher's status."""
    import bench
    seen = {}

    def fake_call(cmd):
        seen["cmd"] = cmd
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
