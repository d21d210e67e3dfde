#!/usr/bin/env python3
"""Copies the results of `tools/gpu_round.sh <tag>` from gpurun_out/ into profiles/ as r1_<letter>_*, replacing the files of
the previous letter, refreshes profiles/traffic.json from the counter passes and prints the figures the design notes quote.
Usage: tools/install_round.py <tag> <new letter> <old letter>"""
import glob, json, os, re, shutil, sys
tag, new, old = sys.argv[1:4]
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
pairs = [("%s_bench_%s.json" % (tag, k), "r1_%s_bench_%s.json" % (new, k)) for k in ("library_fast", "veach_fast", "veach_exact", "library_exact")]
pairs += [("%s_prof_bench_library_fast.txt" % tag, "r1_%s_rocprof_bench_library_fast.txt" % new),
          ("%s_prof_library_fast.txt" % tag, "r1_%s_rocprof_pmc_library128spp_fast.txt" % new),
          ("%s_prof_veach_fast.txt" % tag, "r1_%s_rocprof_pmc_veach128spp_fast.txt" % new)]
for f in glob.glob(os.path.join(P, "r1_%s_bench_*.json" % old)) + glob.glob(os.path.join(P, "r1_%s_rocprof_*.txt" % old)):
    os.remove(f)
for a, b in pairs:
    shutil.copy(os.path.join(G, a), os.path.join(P, b))


def row(path, section, kern):
    s = open(path).read()
    s = s[s.index("== " + section):]
    for l in s.split("\n"):
        if l.startswith(kern):
            return [float(x) for x in l.split()[1:]]


def calls(path, kern):
    for l in open(path):
        if l.startswith(kern):
            return int(l.split()[1])


t = json.load(open(os.path.join(P, "traffic.json")))
for key, kern, name in (("library-synth/fast", "k_extend_wave", "library128spp"), ("veach-mis/fast", "k_shadow_wave", "veach128spp")):
    rel = "profiles/r1_%s_rocprof_pmc_%s_fast.txt" % (new, name)
    path = os.path.join(R, rel)
    v, n = row(path, "pmc3", kern), calls(path, kern)
    t[key]["bytes_per_launch"] = int((v[2] + v[3]) * 64 / n)
    t[key]["source"] = rel
    t[key]["how"] = re.sub(r"over the \d+ ", "over the %d " % n, t[key]["how"])
    p1, p2 = row(path, "pmc1", kern), row(path, "pmc2", kern)
    # SQ_INSTS_VALU x 4 cycles over the SIMD cycles of the launches (1024 SIMDs at 2.4 GHz); active lanes per VALU instruction
    t[key]["valu_issue_frac"] = round(p1[4] * 4 / (p1[0] * 1e-6 * 2.4e9 * 1024), 3)
    t[key]["lanes_per_valu_inst_frac"] = round(p2[6] / (p2[2] * 64), 3)
    t[key]["l2_hit_rate"] = round(v[4] / (v[4] + v[5]), 3)
    print(key, "launches", n, "fabric bytes/launch %.3g" % t[key]["bytes_per_launch"], "L2 hit %.3f" % (v[4] / (v[4] + v[5])),
          "VALU issue %.3f" % (p1[4] * 4 / (p1[0] * 1e-6 * 2.4e9 * 1024)), "lanes per VALU instruction %.3f" % (p2[6] / (p2[2] * 64)))
json.dump(t, open(os.path.join(P, "traffic.json"), "w"), indent=1)
for k in ("library_fast", "veach_fast", "veach_exact", "library_exact"):
    d = json.load(open(os.path.join(P, "r1_%s_bench_%s.json" % (new, k))))
    r = d["roofline"]
    print(k, "Mrays/s %.0f" % d["value"], "samples/s %.3g" % d["samples_per_s"], "ms/step %.0f" % d["ms_per_step"], "| roofline", r["kernel"],
          "frac %.3f" % r["frac"], "avg launch %.3f ms" % r["avg_launch_ms"], "bytes/unit %.0f" % r["bytes_per_unit"], "units/launch %.3g" % r["units_per_launch"],
          "| cpu", d.get("cpu_baseline", {}).get("value"))
print(open(os.path.join(P, "r1_%s_rocprof_bench_library_fast.txt" % new)).read().split("\n")[2])
