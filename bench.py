#!/usr/bin/env python3
"""Throughput of one path-traced frame on N MI355X GPUs of one node.

A step is one frame: every pixel of the workload's camera, `spp` Sobol' samples per pixel, MIS path tracing, rendered
through the C ABI (libpt_hip.so) into device-resident films, followed for N > 1 by one RCCL reduce of the film.
Strong scaling is the default for N > 1: the frame of --spp samples is split by sample number, rank k renders its
contiguous share of [0, spp).  --scaling weak gives every rank --spp samples of its own (an N*spp frame).
The tree is built once per node (rank 0) and handed to the other ranks through /dev/shm.

Prints one JSON line (rank 0).  value = closest-hit + any-hit queries of all ranks per second, in Mrays/s.
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python"))

import numpy as np  # noqa: E402

VALU_MIX_RATE = 0.25    # wave64 vector instructions per SIMD cycle a traversal kernel's instruction mix can issue at (tools/valu_roof.hip)
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak
# What binds the traversal kernels is the rate at which a CU serves per-lane loads of DIFFERENT addresses: about one lane
# and 16 bytes per clock, whichever cache level answers (tools/gather_roof.hip: dependent chains of 64-byte records at 6
# waves per SIMD; profiles/r3_gather_roof.jsonl).  The fallback is that file's figure for an L2-resident table.
GATHER_PEAK_FALLBACK = 2.27e9   # 16-byte lane loads per second and CU
N_CUS = 256
# 16-byte (or narrower) per-lane loads and stores of each kernel per unit besides the BVH records: a navigation node is
# 4 loads when it comes from memory (a binary node 2), a triangle record 3 (layouts: path-tracer_amd/csrc/kernels.h)
LANE_ACCESSES = {
    "generate": 3 + 7,            # three Sobol' table words; radiance slot, path state (4) and hit entry (2) written
    "extend": 3 + 2,              # ray read (a, b, c); hit entry written
    "shadow": 2 + 0.5 * 3,        # ray read; contribution read and radiance slot read and written when unoccluded (~half)
    "shade": 2 + 4 + 8 + 6 + 6 + 7,  # hit entry, path state, triangle record, Sobol' window, light record in; shadow entry + next state out
    "film": 1 + 3,
}
NODE_LOADS = {"exact": 2, "fast": 4}
TRI_LOADS = 3
NODE_BYTES = {"exact": 32, "fast": 64}   # node record read per visit: binary node / 4-wide navigation node
TRI_BYTES = 48             # leaf-array record: triangle {p0, e1, e2, prim} or leaf header (parent box)
# streamed bytes per unit of each kernel besides the BVH records (layouts: path-tracer_amd/csrc/kernels.h)
UNIT_BYTES = {
    "generate": 68 + 16 + 24,        # path state + hit written, radiance slot zeroed, 2 Sobol' dimensions x 3 table words
    "extend": 32 + 16,               # ray read, hit written
    "shadow": 32 + 16 + 32 * 0.5,    # ray + contribution read, radiance slot read-modify-write when unoccluded (~half)
    "shade": 68 + 52 + 48 + 84 + 32 + 232,  # state+hit in, state out, shadow entry out, 7 Sobol' dims, radiance slot, mesh/material/light gathers
    "film": 16 + 24 + 8,             # radiance slot, 2 Sobol' dimensions, share of the patch flush
}


def load_workload(name):
    from pt_hip import synth
    from pt_hip.scene import SceneData
    if name == "veach-mis":
        return SceneData.load(os.path.join(ROOT, "tests", "golden", "veach_mis_scene.npz")), "veach-mis (reference mesh, 3092 tris, 3040 emitters) 1280x720"
    if name == "cornell-box-synth":
        return synth.cornell_box(), "cornell-box-synth (stand-in: reference mesh not in its tree) 768x768"
    if name == "bathroom-synth":
        sd = synth.bathroom_synth()
        return sd, "bathroom-synth (stand-in: reference mesh not in its tree, %d tris, 5 procedural textures) 768x768" % sd.n_tris
    if name.startswith("library-synth"):
        # "library-synth@D" picks the tessellation (triangles ~ D^2; the default 8 gives 1.64 M, 44 gives 49 M);
        # "library-synth@D:WxH" also the frame size (default 1280x720; 3840x2160 switches the sampler to its 4096 grid)
        spec, _, res = (name.split("@")[1] if "@" in name else "8").partition(":")
        w, h = (int(v) for v in res.split("x")) if res else (1280, 720)
        sd = synth.library_synth(detail=int(spec), width=w, height=h)
        return sd, "library-synth (stand-in: reference mesh not in its tree, %d tris) %dx%d" % (sd.n_tris, w, h)
    raise SystemExit("unknown workload " + name)


def kernel_source_hash():
    """Digest of the device sources: counter figures quoted from profiles/traffic.json are stale once it changes."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "path-tracer_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def shared_tree(sd, rank, world, barrier, device=0):
    """The reference-topology tree of the scene, built once per node: rank 0 builds and writes it to /dev/shm, the other
    ranks map it.  Returns (nodes, prim_ids) or None when there is only one rank (Scene builds its own)."""
    if world == 1:
        return None
    import numpy as np
    from pt_hip import api
    stem = "/dev/shm/pt_bench_tree_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.getuid())
    if rank == 0:
        nodes, ids, _ = api.build_bvh_gpu(np.ascontiguousarray(sd.verts, np.float32), np.ascontiguousarray(sd.vi, np.uint32), device=device)
        np.save(stem + "_nodes.npy", nodes.view(np.uint32).reshape(-1, 8))
        np.save(stem + "_ids.npy", ids)
    barrier()
    if rank != 0:
        nodes = np.load(stem + "_nodes.npy", mmap_mode="r").reshape(-1).view(api.NODE_DTYPE)
        ids = np.load(stem + "_ids.npy", mmap_mode="r")
    tree = (np.ascontiguousarray(nodes), np.ascontiguousarray(ids))
    barrier()
    if rank == 0:
        for f in (stem + "_nodes.npy", stem + "_ids.npy"):
            os.remove(f)
    return tree


def gather_peak():
    """Measured roof of a CU's dependent-gather rate (16-byte lane loads per second): the 64-byte-record chain on a 1 MB
    table at six waves per SIMD from the committed run of tools/gather_roof.hip."""
    try:
        for line in open(os.path.join(ROOT, "profiles", "r3_gather_roof.jsonl")):
            d = json.loads(line)
            if d["mode"] == "g4" and d["table_bytes"] == 1 << 20 and d["waves_per_simd"] == 6 and d.get("valu_per_step", 0) == 0:
                return d["lane_records_per_cu_per_us"] * 1e6 * 4, "profiles/r3_gather_roof.jsonl"
    except (OSError, ValueError, KeyError):
        pass
    return GATHER_PEAK_FALLBACK, "fallback constant"


def counters_ok(c):
    return bool(c) and c.get("bytes_per_launch") and c.get("avg_launch_ms")


def traffic_of(c):
    return c["bytes_per_launch"]


def host_cores():
    """(threads the CPU baseline uses, cores this process may run on, cores of the machine).  The GPU box gives a one-GPU
    job a share of a much larger host, and says so through OMP_NUM_THREADS."""
    machine = os.cpu_count() or 1
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else machine
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    return max(1, min(allowed, share) if share > 0 else allowed), allowed, machine


def cpu_baseline(sd, budget_s=15.0):
    """The oracle (a port of the reference algorithm) on the host cores, on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    cores, allowed, machine = host_cores()
    o = orc.Scene(sd)
    _, st = o.render(0, 1, nthreads=cores)              # probe: 1 spp, whole frame
    rate = st["paths"] / max(st["seconds"], 1e-6)
    spp = int(max(1, min(64, budget_s * rate / st["paths"])))
    _, st = o.render(0, spp, nthreads=cores)
    rays = st["closest_rays"] + st["any_rays"]
    return {"value": rays / st["seconds"] / 1e6, "unit": "Mrays/s", "cores": cores, "cores_allowed": allowed, "cores_of_machine": machine,
            "cores_note": "threads used = this job's share of the host (OMP_NUM_THREADS on the GPU box), never more than its affinity mask",
            "kind": "port",
            "sample": "whole frame, %d spp of the same workload (%.1f s)" % (spp, st["seconds"]),
            "samples_per_s": st["paths"] / st["seconds"],
            "ref_order_nodes_per_ray": st["nodes_visited"] / rays, "ref_order_tris_per_ray": st["tris_tested"] / rays}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("PT_BENCH_WORKLOAD", "library-synth"))
    ap.add_argument("--spp", type=int, default=1024, help="samples per pixel of the frame (per GPU with --scaling weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default): N GPUs share one frame of --spp samples; weak: every GPU renders --spp samples of its own")
    ap.add_argument("--traversal", choices=["exact", "fast"], default="fast")
    ap.add_argument("--split", choices=["spp", "tile"], default="spp",
                    help="multi-GPU partition: spp = by Sobol' sample number (all pixels on every rank); "
                         "tile = the ranks share the 32x32 pixel blocks of one frame of --spp samples (always strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Launched bare: start one rank per GPU as fresh child processes.  Nothing in this process has touched a GPU yet
        # (a process that has must not be replaced or forked into ranks), and it only waits for them.
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    from pt_hip import api
    from pt_hip.dist import block_share, reduce_film, spp_range_strong, spp_range_weak

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (a.gpus, world, a.gpus))
    # PT_BENCH_REHEARSE=1: ranks share whatever GPUs there are and reduce through gloo and host memory -- a dry run of the
    # multi-rank code path on a box with fewer GPUs than ranks (its numbers mean nothing)
    rehearse = os.environ.get("PT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    sd, label = load_workload(a.workload)
    scene = api.Scene(sd, device=local_rank, bvh=shared_tree(sd, rank, world, dist.barrier if world > 1 else None, local_rank))
    film = torch.empty((sd.height + 4, sd.width + 4, 4), dtype=torch.float32, device="cuda:%d" % local_rank)
    mode = api.TRAVERSE_EXACT if a.traversal == "exact" else api.TRAVERSE_FAST
    tiles = a.split == "tile" and world > 1
    strong = tiles or a.scaling == "strong"
    s0, s1 = (0, a.spp) if tiles else (spp_range_strong(rank, world, a.spp) if strong else spp_range_weak(rank, a.spp))
    share = block_share(rank, world) if tiles else (0, 0)
    if s1 <= s0:
        raise SystemExit("--spp %d leaves rank %d without samples" % (a.spp, rank))

    def step():
        st = scene.render_device(film.data_ptr(), s0, s1, traversal=mode, block=share)
        if rehearse and world > 1:
            host = film.cpu()
            reduce_film(host, dst=0)
            film.copy_(host)
        else:
            reduce_film(film, dst=0)
        if world > 1:
            torch.cuda.synchronize()   # the next step clears the film on the renderer's own stream: the reduce must be done with it
        return st

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = [step() for _ in range(a.steps)]
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, float(sum(s["closest_rays"] + s["any_rays"] for s in stats)), float(sum(s["paths"] for s in stats))],
                     dtype=torch.float64, device="cpu" if rehearse else "cuda")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
    rays_total, paths_total = float(t[1]), float(t[2])

    if rank == 0:
        # per-kernel device time from HIP events recorded on the launch stream inside the timed region
        kms = {k: sum(s["kernel_ms"].get(k, 0.0) for s in stats) for k in api.KERNEL_NAMES[:5]}
        extra_ms = {k: sum(s["kernel_ms"].get(k, 0.0) for s in stats) for k in ("finish", "compact")}   # tail launch, path gathering
        kn = {k: sum(s["kernel_launches"].get(k, 0) for s in stats) for k in api.KERNEL_NAMES[:5]}
        dominant = max(kms, key=kms.get)
        # average BVH records per ray of each traversal kernel from an instrumented (untimed) pass of the size of the timed
        # region's passes (which kernels carry the late bounces depends on the pass size)
        pass_spp = max(1, stats[0]["samples_per_pass"])
        cst = scene.render_device(film.data_ptr(), s0, min(s1, s0 + pass_spp), traversal=mode, flags=api.FLAG_COUNT_WORK, block=share)
        units = {"generate": sum(s["paths"] for s in stats), "film": sum(s["paths"] for s in stats),
                 "extend": sum(s["closest_rays"] - s["paths"] for s in stats), "shadow": sum(s["any_rays"] for s in stats),
                 "shade": sum(s["shaded_hits"] for s in stats)}    # misses leave the queue before shading
        cunits = {"generate": cst["paths"], "extend": cst["closest_rays"] - cst["paths"], "shadow": cst["any_rays"]}
        per_unit = dict(UNIT_BYTES)
        lane_acc = dict(LANE_ACCESSES)
        bvh = {}
        for k in ("generate", "extend", "shadow"):
            n_node = cst["nodes_fetched"][k] / max(cunits[k], 1)
            n_lds = cst["nodes_from_lds"][k] / max(cunits[k], 1)
            n_tri = cst["tris_fetched"][k] / max(cunits[k], 1)
            bvh[k] = {"nodes_per_ray": n_node, "of_them_from_lds": n_lds, "tris_per_ray": n_tri}
            per_unit[k] += NODE_BYTES[a.traversal] * n_node + TRI_BYTES * n_tri
            lane_acc[k] += NODE_LOADS[a.traversal] * (n_node - n_lds) + TRI_LOADS * n_tri
        # Hardware counters cannot be read from inside this process: the counter figures of the dominant kernel come from the
        # committed rocprofv3 --pmc passes of the same workload and pass size (profiles/traffic.json), stamped with the digest
        # of the device sources they were measured on; a different digest marks them stale.
        counters, stale = {}, None
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("%s/%s" % (a.workload, a.traversal), {}).get("k_" + dominant)
            if t:
                counters = t
                stale = t.get("kernel_source_hash") != kernel_source_hash()
        except (OSError, ValueError, KeyError):
            pass
        seconds = kms[dominant] * 1e-3
        achieved_hbm = units[dominant] * per_unit[dominant] / seconds / 1e9 if seconds > 0 else 0.0
        achieved_gather = units[dominant] * lane_acc[dominant] / seconds / 1e9 if seconds > 0 else 0.0   # G lane accesses per second
        peak_cu, peak_src = gather_peak()
        peak_gather = peak_cu * N_CUS / 1e9
        # The traversal kernels are priced against the gather rate.  k_shade streams its queues (coalesced) and gathers one
        # 128-byte record per hit: its lane accesses are mostly of the cheap, coalesced kind, and what it moves is priced
        # against HBM instead.
        gather_bound = dominant in ("generate", "extend", "shadow")
        hbm_measured = (traffic_of(counters) / counters["avg_launch_ms"] / 1e-3 / 1e9) if counters_ok(counters) else None   # GB/s that crossed the fabric
        launch_s = kms[dominant] / max(kn[dominant], 1) * 1e-3
        traffic = counters.get("bytes_per_launch")
        valu, lanes = counters.get("valu_issue_frac"), counters.get("lanes_per_valu_inst_frac")
        # Vector instructions per SIMD cycle of this run: the committed counter pass's figure (instructions x 2 / SIMD cycles)
        # scaled by the launch time measured here (same kernel, same instructions).  What a SIMD can take: tools/valu_roof.hip.
        valu_rate = None
        if valu and counters.get("avg_launch_ms") and kn[dominant]:
            valu_rate = valu / 2.0 * counters["avg_launch_ms"] / (kms[dominant] / kn[dominant])
        # The binding resource is the one the kernel stands closest to: vector issue, or the gather rate (traversal kernels)
        # / the HBM peak (k_shade) when no counter pass of this workload is committed or the kernel is nearer to those.
        other_frac = achieved_gather / peak_gather if gather_bound else (hbm_measured if hbm_measured is not None else achieved_hbm) / HBM_PEAK_GBS
        valu_bound = valu_rate is not None and valu_rate / VALU_MIX_RATE >= other_frac
        out = {
            "metric": "Mrays/s", "value": rays_total / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if "synth" in a.workload else "reference mesh, no textures",
            "config": {"workload": label, "integrator": "MIS path tracer, 32 bounces, Sobol'",
                       "spp_per_frame" if strong else "spp_per_gpu": a.spp, "traversal": a.traversal,
                       "traversal_note": ("ordered 4-wide walk: identical to the reference-order walk in every bit on all tested rays (parity tests, fuzzers), "
                                          "not by construction: a ray that grazes a large triangle below 1e-4 rad with |det| just above the reference's 1e-5 "
                                          "threshold can differ; --traversal exact follows the reference there too")
                                         if a.traversal == "fast" else "reference visit order: identical to the reference by construction",
                       "samples_per_pass": pass_spp,
                       "parallelism": (("32x32-block split x%d" if tiles else ("spp-range split of one frame x%d" if strong else "spp-range x%d, one frame share each"))
                                       + " + 1 film reduce, tree built once per node") % world},
            "samples_per_s": paths_total / dt,
            "rays_per_path": rays_total / max(paths_total, 1.0),
            "kernel_ms_per_step": dict({k: v / a.steps for k, v in kms.items()}, **{k: v / a.steps for k, v in extra_ms.items()}),
            "kernel_launches_per_step": {k: v / a.steps for k, v in kn.items()},
            "bvh_records_per_ray": bvh,
            # The dominant kernel against the resource that binds it.  The traversal kernels gather: every lane loads its own
            # node and triangle records, 16 bytes per instruction, and a CU serves such loads at about one lane per clock
            # whatever cache level they hit in (measured by tools/gather_roof.hip) -- far below what HBM, L2 or the vector ALU
            # could take.  achieved = 16-byte lane accesses the kernel's queries need (counted by an instrumented pass) per
            # second of kernel time; peak = the measured rate of a chain of dependent 64-byte gathers from an L2-resident table
            # x 256 CUs; frac = achieved / peak.  The contract's HBM view of the same kernel is under "hbm".
            "roofline": {"bound": "valu-issue" if valu_bound else ("l1-gather" if gather_bound else "hbm"),
                         "bound_note": ("vector instruction issue: the SIMDs issue this kernel's instructions at the rate its mix allows "
                                        "(multiply-adds, adds, moves and bit operations 2.5-3.3 cycles per wave64 instruction, conversions, min / max, "
                                        "compares, selects and three-operand integer instructions 4.4-5: tools/valu_roof.hip); its time follows its "
                                        "instruction count, one for one, and not its loads -- the second limit is "
                                        + ("the gather rate under 'gather'" if gather_bound else "the fabric traffic under 'hbm' (measured_frac)")) if valu_bound else
                                       ("per-lane gathers of different addresses: the CU's address/data path takes one lane x 16 B per clock "
                                        "(tools/gather_roof.hip)") if gather_bound else
                                       ("queues streamed once per bounce plus one gathered 128-byte record per hit; algorithmic bytes count what "
                                        "the caches serve too, the measured share of the HBM peak is hbm.measured_frac"),
                         "kernel": "k_" + dominant,
                         "achieved": valu_rate if valu_bound else (achieved_gather if gather_bound else (hbm_measured if hbm_measured is not None else achieved_hbm)),
                         "achieved_is": "wave64 vector instructions (committed counter pass) / SIMD cycles of this run's launches" if valu_bound else
                                        ("counted lane accesses / kernel time" if gather_bound else
                                         ("measured fabric bytes / kernel time (committed counter pass)" if hbm_measured is not None else "algorithmic bytes / kernel time")),
                         "peak": VALU_MIX_RATE if valu_bound else (peak_gather if gather_bound else HBM_PEAK_GBS),
                         "unit": "wave64 vector instructions per SIMD cycle" if valu_bound else ("G lane-accesses/s (16 B each)" if gather_bound else "GB/s"),
                         "frac": min(1.0, valu_rate / VALU_MIX_RATE) if valu_bound else
                                 (achieved_gather / peak_gather if gather_bound else (hbm_measured if hbm_measured is not None else achieved_hbm) / HBM_PEAK_GBS),
                         "valu": {"achieved": valu_rate, "peak": VALU_MIX_RATE, "unit": "wave64 vector instructions per SIMD cycle",
                                  "frac": (valu_rate / VALU_MIX_RATE) if valu_rate is not None else None,
                                  "cycles_per_instruction": (1.0 / valu_rate) if valu_rate else None,
                                  "peak_source": "profiles/r3_valu_roof.jsonl: six waves per SIMD sustain one wave64 instruction per 2.5-3.3 cycles "
                                                 "(v_fma / v_mul / v_add / v_mov / v_and / shifts) or per 4.4-5 (v_cvt, v_min / v_max, v_cmp, v_cndmask, "
                                                 "v_alignbit, three-operand integer, double precision, the pieces of a division) or per 10 (v_rcp, v_sqrt); "
                                                 "a traversal step is two thirds of the second kind, the shading code's mix weighs the same: one per 4 cycles"},
                         "gather": {"achieved": achieved_gather, "peak": peak_gather, "unit": "G lane-accesses/s (16 B each)", "frac": achieved_gather / peak_gather,
                                    "peak_source": peak_src, "lane_accesses_per_unit": lane_acc[dominant],
                                    "note": "every access counted as a gather; coalesced queue accesses cost a quarter of that"},
                         "traffic": traffic, "traffic_unit": "HBM-side bytes per launch (fabric requests x 64 B)",
                         "hbm": {"achieved": achieved_hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_hbm / HBM_PEAK_GBS,
                                 "note": "algorithmic bytes per unit x units / kernel time: counts gathers that L2 serves, so it may exceed the peak",
                                 "measured_frac": (traffic / counters["avg_launch_ms"] / 1e-3 / 1e9 / HBM_PEAK_GBS) if traffic and counters.get("avg_launch_ms") else None,
                                 "algorithmic_bytes_per_launch": per_unit[dominant] * units[dominant] / max(kn[dominant], 1),
                                 "bytes_per_unit": per_unit[dominant]},
                         "valu_issue_frac": valu, "lane_frac": lanes, "wave_wait_frac": counters.get("wave_wait_frac"), "l2_hit_rate": counters.get("l2_hit_rate"),
                         "ta_busy_frac": counters.get("ta_busy_frac"), "td_busy_frac": counters.get("td_busy_frac"),
                         "counters_source": counters.get("source"), "counters_how": counters.get("how"),
                         "counters_measured_at": {k: counters.get(k) for k in ("git", "kernel_source_hash", "spp_per_pass", "avg_launch_ms")} if counters else None,
                         "stale": stale,
                         "units_per_launch": units[dominant] / max(kn[dominant], 1),
                         "avg_launch_ms": launch_s * 1e3},
        }
        if rehearse:
            # ranks shared GPUs and reduced through host memory: the code path ran, the figures mean nothing
            out.update({"rehearsal": True, "metric": "REHEARSAL " + out["metric"], "rehearsal_value": out["value"], "value": None, "vs_baseline": None})
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
