#!/usr/bin/env python3
"""Copies the results of `tools/gpu_round.sh <tag>` from gpurun_out/ into profiles/ as r3_<letter>_*, replacing the files of
the previous letter, refreshes profiles/traffic.json from the counter passes and prints the figures the design notes quote.
Usage: tools/install_round.py <tag> <new letter> <old letter>"""
import glob, json, os, re, shutil, subprocess, sys
tag, new, old = sys.argv[1:4]
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
pairs = [("%s_bench_%s.json" % (tag, k), "r3_%s_bench_%s.json" % (new, k)) for k in ("library_fast", "veach_fast", "veach_exact", "library_exact", "cornell64spp_fast", "bathroom2048spp_fast")]
pairs += [("%s_prof_bench_library_fast.txt" % tag, "r3_%s_rocprof_bench_library_fast.txt" % new),
          ("%s_prof_library_fast.txt" % tag, "r3_%s_rocprof_pmc_library512spp_fast.txt" % new),
          ("%s_prof_veach_fast.txt" % tag, "r3_%s_rocprof_pmc_veach512spp_fast.txt" % new)]
for a, _ in pairs:
    if a.endswith(".json") and json.load(open(os.path.join(G, a))).get("rehearsal"):
        raise SystemExit("%s is the line of a rehearsal run (ranks sharing GPUs): not a measurement, not installed" % a)
if old != new:
    for f in glob.glob(os.path.join(P, "r3_%s_bench_*.json" % old)) + glob.glob(os.path.join(P, "r3_%s_rocprof_*.txt" % old)):
        os.remove(f)
for a, b in pairs:
    shutil.copy(os.path.join(G, a), os.path.join(P, b))


def row(path, section, kern):
    s = open(path).read()
    s = s[s.index("== " + section):]
    rows = []
    for l in s.split("\n"):
        if l.startswith("== ") and rows:
            break                       # the next section
        if l.startswith(kern + " "):
            rows.append([float(x) for x in l.split()[1:]])
    return max(rows, key=lambda r: r[0]) if rows else None   # several instantiations: the one that ran longest


def calls(path, kern):
    for l in open(path):
        if l.startswith(kern):
            return int(l.split()[1])


sys.path.insert(0, R)
import bench  # noqa: E402  (kernel_source_hash)
stamp = {"git": subprocess.check_output(["git", "-C", R, "rev-parse", "--short", "HEAD"]).decode().strip(), "kernel_source_hash": bench.kernel_source_hash(),
         "spp_per_pass": 512}
t = {}
for key, name, note in (("library-synth/fast", "library512spp", "gathers of 64-B lines are outside the guide's calibration (may read up to 2x low)"),
                        ("veach-mis/fast", "veach512spp", "the whole scene (0.3 MB of nodes and triangles) lives in L2, what reaches the fabric is queue traffic")):
    rel = "profiles/r3_%s_rocprof_pmc_%s_fast.txt" % (new, name)
    path = os.path.join(R, rel)
    t[key] = {}
    for kern in ("k_extend", "k_shadow", "k_generate", "k_shade"):
        sym = kern if kern == "k_shade" else kern + "_wave"
        v, n = row(path, "pmc3", sym), calls(path, sym)
        p1, p2, p4 = row(path, "pmc1", sym), row(path, "pmc2", sym), row(path, "pmc4", sym)
        # cycles of the launches: GRBM_GUI_ACTIVE over the 8 XCDs, scaled from its pass to this one; when that pass was
        # disturbed (the counter then runs far ahead of the kernel's own busy cycles) SQ_BUSY_CYCLES over the 32 shader
        # engines of the same pass as the instruction count
        cyc, cyc_sq = v[1] / 8 * p1[0] / v[0], p1[1] / 32
        cycles_from = "GRBM_GUI_ACTIVE / 8"
        if cyc > 1.3 * cyc_sq:
            cyc, cycles_from = cyc_sq, "SQ_BUSY_CYCLES / 32"
        t[key][kern] = {
            "bytes_per_launch": int((v[2] + v[3]) * 64 / n),
            "how": "(TCC_EA0_RDREQ_sum + TCC_EA0_WRREQ_sum) x 64 B over the %d %s launches of one 512-spp pass (the size of the bench's passes); %s"
                   % (n, sym, note),
            "source": rel,
            # SQ_INSTS_VALU over the VALU issue slots of the launches: a CDNA4 SIMD is 32 lanes wide and issues one wave64
            # instruction per 2 cycles when several waves are ready (one wave alone sustains one per 4 cycles: the second
            # figure, reported next to it); 1024 SIMDs
            "valu_issue_frac": round(p1[4] * 2 / 1024 / cyc, 3),
            "valu_issue_frac_at_one_wave_rate": round(p1[4] * 4 / 1024 / cyc, 3),
            "cycles_from": cycles_from,
            "lanes_per_valu_inst_frac": round(p2[6] / (p2[2] * 64), 3),
            # share of the resident waves' time spent waiting (memory, LDS, dependencies): SQ_WAIT_ANY / SQ_WAVE_CYCLES
            "wave_wait_frac": round(p2[7] / p1[8] * (p1[0] / p2[0]), 3),
            "l2_hit_rate": round(v[4] / (v[4] + v[5]), 3),
            # busy share of the 256 texture-address and texture-data units over the launches (cycles of their own pass:
            # GRBM_GUI_ACTIVE over the 8 XCDs); "busy" includes the time a unit waits for the cache behind it
            "ta_busy_frac": round(p4[3] / 256 / (p4[1] / 8), 3), "td_busy_frac": round(p4[5] / 256 / (p4[1] / 8), 3),
            "td_stalled_by_cache_frac": round(p4[4] / 256 / (p4[1] / 8), 3),
            "avg_launch_ms": round(v[0] / n * 1e-3, 4), **stamp}
        print(key, kern, t[key][kern]["bytes_per_launch"], t[key][kern]["valu_issue_frac"], t[key][kern]["lanes_per_valu_inst_frac"], t[key][kern]["l2_hit_rate"])
json.dump(t, open(os.path.join(P, "traffic.json"), "w"), indent=1)
for k in ("library_fast", "veach_fast", "veach_exact", "library_exact"):
    d = json.load(open(os.path.join(P, "r3_%s_bench_%s.json" % (new, k))))
    r = d["roofline"]
    print(k, "Mrays/s %.0f" % d["value"], "samples/s %.3g" % d["samples_per_s"], "ms/step %.0f" % d["ms_per_step"], "| roofline", r["kernel"],
          "frac", r["frac"], "stale", r["stale"], "avg launch %.3f ms" % r["avg_launch_ms"], "lane accesses/unit %.1f" % r["gather"]["lane_accesses_per_unit"], "bytes/unit %.0f" % r["hbm"]["bytes_per_unit"], "units/launch %.3g" % r["units_per_launch"],
          "| cpu", d.get("cpu_baseline", {}).get("value"))
print(open(os.path.join(P, "r3_%s_rocprof_bench_library_fast.txt" % new)).read().split("\n")[2])
