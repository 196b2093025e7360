"""Developer tool: turn a gpurun_out/<dir> written by tools/collect_profiles.sh into the tracked files of profiles/.
    python tools/assemble_profiles.py gpurun_out/r01_g r01_g"""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def pmc(dirn, kern):
    vals = {}
    for f in glob.glob(dirn + "/**/*counter_collection.csv", recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            vals[k] = sum(v) / len(v)
    return vals


for w in ("mesh_fwd", "mesh_fwdbwd", "points_fwd", "c5_per_gpu", "c5", "soft_fwdbwd", "geo_cad", "geo_ragged"):
    fs = glob.glob("%s/stats_%s/**/*kernel_stats.csv" % (src, w), recursive=True)
    if not fs:
        print(w, "(no kernel stats)")
        continue
    f = fs[0]
    shutil.copy(f, "%s/%s_kernel_stats_%s.csv" % (dst, tag, w))
    shutil.copy("%s/bench_%s.json" % (src, w), "%s/%s_bench_%s.json" % (dst, tag, w))
    print(w)
    for r in csv.DictReader(open(f)):
        print("   %-46s calls %4s avg %8.1f us" % (r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3))
note = ("separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `rocprofv3 --pmc X --output-format csv -- python3 bench.py --steps 5 --warmup 2 "
        "--no-cpu-baseline [--workload W]` (tools/collect_profiles.sh), averaged per launch; gfx950 correction: FETCH_SIZE doubled "
        "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE calibrated on the points kernel (235200 KiB == its 240 844 800 output bytes)")
for w, kern, out in (("mesh_fwd", "k_mesh_raster_fast", tag + "_traffic_mesh_fwd.json"), ("points_fwd", "k_points_raster_shade", tag + "_traffic_points_fwd.json"),
                     ("mesh_fwdbwd", "k_mesh_raster_fast", tag + "_traffic_mesh_fwdbwd.json"), ("mesh_fwdbwd", "k_mesh_backward_tab", tag + "_traffic_mesh_bwd_kernel.json")):
    fe = pmc("%s/pmc_FETCH_SIZE_%s" % (src, w), kern)["FETCH_SIZE"]
    wr = pmc("%s/pmc_WRITE_SIZE_%s" % (src, w), kern)["WRITE_SIZE"]
    d = {"kernel": kern.rstrip("("), "FETCH_SIZE_KiB": fe, "WRITE_SIZE_KiB": wr, "hbm_bytes_per_launch": int((2 * fe + wr) * 1024), "note": note}
    json.dump(d, open(os.path.join(dst, out), "w"), indent=1)
    print(out, d["hbm_bytes_per_launch"])
fe = pmc("%s/pmc_FETCH_SIZE_mesh_fwd" % src, "k_mesh_prepass")["FETCH_SIZE"]
wr = pmc("%s/pmc_WRITE_SIZE_mesh_fwd" % src, "k_mesh_prepass")["WRITE_SIZE"]
sq = {}
for kern in ("k_mesh_raster_fast", "k_mesh_backward_tab", "k_mesh_prepass"):
    v = {}
    v.update(pmc("%s/pmc_sq1_mesh_fwdbwd" % src, kern))
    v.update(pmc("%s/pmc_sq2_mesh_fwdbwd" % src, kern))
    v.update(pmc("%s/pmc_sq3_mesh_fwdbwd" % src, kern))
    sq[kern.rstrip("(")] = v
# the forward rasteriser as the inference step (C2) runs it; the entry above is the training step's (it also writes the pixel lists)
if glob.glob("%s/pmc_sq1_mesh_fwd/**/*counter_collection.csv" % src, recursive=True):
    sq["k_mesh_raster_fast (training step)"] = sq["k_mesh_raster_fast"]
    sq["k_mesh_raster_fast"] = pmc("%s/pmc_sq1_mesh_fwd" % src, "k_mesh_raster_fast")
sq["k_points_raster_shade"] = pmc("%s/pmc_sq1_points_fwd" % src, "k_points_raster_shade")
if glob.glob("%s/pmc_sq1_soft_fwdbwd/**/*counter_collection.csv" % src, recursive=True):
    for kern in ("k_mesh_soft_forward", "k_mesh_soft_backward("):
        sq[kern.rstrip("(")] = pmc("%s/pmc_sq1_soft_fwdbwd" % src, kern)
if glob.glob("%s/pmc_sq1_geo_cad/**/*counter_collection.csv" % src, recursive=True):   # round 6: the CAD-like batch's forward
    sq["k_mesh_raster_fast (CAD-like batch)"] = pmc("%s/pmc_sq1_geo_cad" % src, "k_mesh_raster_fast")
if os.path.exists("%s/bench_driver_like.json" % src):
    shutil.copy("%s/bench_driver_like.json" % src, "%s/%s_bench_driver_like.json" % (dst, tag))
sq["_note"] = ("rocprofv3 --pmc passes of tools/collect_profiles.sh (C3 step for the mesh kernels, C4 for the points kernel); SQ_*_CYCLES / "
               "SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles summed over waves, SQ_BUSY_CYCLES = 32 x kernel cycles")
sq["_prepass_hbm_bytes_per_launch"] = int((2 * fe + wr) * 1024)
json.dump(sq, open(os.path.join(dst, "%s_pmc_sq_counters.json" % tag), "w"), indent=1)
for k, v in sq.items():
    if isinstance(v, dict) and "SQ_ACTIVE_INST_VALU" in v:
        print(k, "VALU instrs %.1fM, VALU busy %.0f%% of kernel" % (v["SQ_INSTS_VALU"] / 1e6, 100 * v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v["SQ_BUSY_CYCLES"] / 32)))
print("pre-pass HBM bytes per launch", sq["_prepass_hbm_bytes_per_launch"])
