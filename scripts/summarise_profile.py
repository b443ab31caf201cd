"""Turn the rocprofv3 output of scripts/profile_bench.sh (+ profile_pmc2.sh) under gpurun_out/prof_<tag>/ into the
tracked files profiles/<name>_kernel_stats.csv, profiles/<name>_bench_rocprofv3_summary.json and
profiles/<name>_hbm_traffic.json.

    python scripts/summarise_profile.py <tag> [<name>]        (name defaults to the tag)

Counters are averaged per dispatch of the render kernel ("render_") and, prefixed "resolve:", of the resolve kernel.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB; FETCH_SIZE is doubled per the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (HBM section)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else tag
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.environ.get("PROF_DST", os.path.join(ROOT, "profiles"))


def one(pattern):
    # gpurun MERGES a call's output into gpurun_out/, it does not clear what an earlier call left there: take the NEWEST file of a kind
    f = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


stats_csv = one("trace/**/*_kernel_stats.csv")
assert stats_csv, "no kernel stats under " + src
shutil.copy(stats_csv, os.path.join(dst, name + "_kernel_stats.csv"))
kstats = [r for r in csv.DictReader(open(stats_csv)) if "rtw::" in r["Name"]]

dispatch = {}
trace = one("trace/**/*_kernel_trace.csv")
if trace:
    for r in csv.DictReader(open(trace)):
        if "render_" in r["Kernel_Name"]:
            dispatch = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                          "Accum_VGPR_Count", "SGPR_Count") if k in r}
            break

pmc = {}
_newest = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
    d = os.path.relpath(f, src).split(os.sep)[0]                      # one file per pass directory: the newest (see one())
    if d not in _newest or os.path.getmtime(f) > os.path.getmtime(_newest[d]):
        _newest[d] = f
for f in _newest.values():
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        key = r["Counter_Name"] if "render_" in kn else ("resolve:" + r["Counter_Name"] if "resolve_kernel" in kn else None)
        if key:
            agg[key].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k] = {"dispatches": len(v), "mean_per_dispatch": sum(v) / len(v)}


def m(k):
    return pmc[k]["mean_per_dispatch"] if k in pmc else None


derived = {}
for r in kstats:
    if "render_" in r["Name"]:
        derived["render_kernel_avg_ms"] = float(r["AverageNs"]) * 1e-6
    if "resolve_kernel" in r["Name"]:
        derived["resolve_kernel_avg_ms"] = float(r["AverageNs"]) * 1e-6
if m("SQ_THREAD_CYCLES_VALU") and m("SQ_ACTIVE_INST_VALU"):
    derived["valu_lane_utilisation"] = m("SQ_THREAD_CYCLES_VALU") / (64.0 * m("SQ_ACTIVE_INST_VALU"))
if m("GRBM_GUI_ACTIVE") and "render_kernel_avg_ms" in derived:
    cycles = m("GRBM_GUI_ACTIVE") / 8.0                       # summed over the 8 XCDs
    derived["shader_clock_GHz"] = cycles / (derived["render_kernel_avg_ms"] * 1e6)
    if m("SQ_INSTS_VALU"):
        derived["valu_issues_per_cycle_per_simd"] = m("SQ_INSTS_VALU") / (cycles * 1024.0)
    tot = sum(x for x in (m("SQ_INSTS_VALU"), m("SQ_INSTS_SALU"), m("SQ_INSTS_BRANCH"), m("SQ_INSTS_LDS"), m("SQ_INSTS_SMEM"),
                          m("SQ_INSTS_VMEM_RD"), m("SQ_INSTS_VMEM_WR")) if x)
    derived["all_instructions_per_cycle_per_simd"] = tot / (cycles * 1024.0)
if m("SQ_WAVE_CYCLES"):
    for k, out in (("SQ_ACTIVE_INST_ANY", "wave_cycles_active_frac"), ("SQ_WAIT_ANY", "wave_cycles_wait_any_frac"),
                   ("SQ_WAIT_INST_ANY", "wave_cycles_wait_inst_frac")):
        if m(k):
            derived[out] = m(k) / m("SQ_WAVE_CYCLES")
hbm = None
if m("FETCH_SIZE") is not None and m("WRITE_SIZE") is not None:
    rf, rw = m("FETCH_SIZE") * 1024.0 * 2.0, m("WRITE_SIZE") * 1024.0
    sf, sw = (m("resolve:FETCH_SIZE") or 0.0) * 1024.0 * 2.0, (m("resolve:WRITE_SIZE") or 0.0) * 1024.0
    hbm = rf + rw + sf + sw
    derived.update(hbm_bytes_per_step=hbm, render_fetch_bytes_x2=rf, render_write_bytes=rw, resolve_fetch_bytes_x2=sf,
                   resolve_write_bytes=sw,
                   hbm_bytes_per_step_note="render + resolve; FETCH_SIZE/WRITE_SIZE are in KB; FETCH_SIZE doubled per the gfx950 "
                                           "correction (MI355X_MICROARCH.md HBM)")

_args_file = os.path.join(src, "bench_args.txt")
_extra = open(_args_file).read().strip() if os.path.exists(_args_file) else ""
summary = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline" + (" " + _extra if _extra else "") +
                      "   (then one --pmc pass per counter group, same command; scripts/profile_bench.sh, scripts/profile_pmc2.sh; "
                      "summarised by scripts/summarise_profile.py)",
           "kernel_stats": kstats, "dispatch": dispatch, "pmc": dict(sorted(pmc.items())), "derived": derived}
json.dump(summary, open(os.path.join(dst, name + "_bench_rocprofv3_summary.json"), "w"), indent=1)
if hbm is not None:
    json.dump({"hbm_bytes_per_launch": hbm,
               "source": f"profiles/{name}_bench_rocprofv3_summary.json (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes; FETCH_SIZE x2 "
                         "gfx950 correction; render + resolve kernels of one step)",
               "algorithmic_bytes_per_launch": 24908083200}, open(os.path.join(dst, name + "_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(derived, indent=1))
