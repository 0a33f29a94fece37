#!/usr/bin/env python3
"""Condense gpurun_out/profiles_<tag>/ (rocprofv3 CSVs) into the tracked summaries under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"profiles_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 110 else name[:107] + "..."


def stats_table(d, out, top=12):
    f = sorted(glob.glob(os.path.join(src, d, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if not f:
        return None
    rows = list(csv.DictReader(open(f[-1])))  # newest run
    with open(os.path.join(dst, out), "w") as fh:
        fh.write(f"# rocprofv3 --kernel-trace --stats ({d})\n\n| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
        for r in rows[:top]:
            fh.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | "
                     f"{float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |\n")
    return rows


def pmc(d, kernel_filter=("gl_pair_kernel<3", "gl_static_kernel<3", "gl_main_kernel<3", "gl_cluster_kernel<3", "gl_clusterw_kernel<3", "gl_shp_kernel<3")):
    f = sorted(glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    if not f:
        return {}
    agg, dur = collections.defaultdict(list), []
    for r in csv.DictReader(open(f[-1])):  # newest run
        if any(k in r["Kernel_Name"] for k in kernel_filter):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {k: sum(v[1:]) / max(len(v[1:]), 1) for k, v in agg.items()}
    if dur:
        out["_kernel_us"] = sum(dur[1:]) / max(len(dur[1:]), 1) / 1e3
    return out


summary = {}
rows = stats_table("bench_stats", f"{tag}_bench_kernel_stats.md")
for w in ("C2", "C3", "C3direct", "C3D", "C4", "C5", "C6", "C3L", "simpair", "demo"):
    stats_table(f"kernel_stats_{w}", f"{tag}_{w}_kernel_stats.md", top=6)
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_mix", "pmc_clk"):
    summary[d] = pmc(d)
summary["pmc_C6"] = pmc("pmc_C6", kernel_filter=("gl_main_kernel<3",))
summary["pmc_C4"] = pmc("pmc_C4", kernel_filter=("gl_clusterw_kernel<3", "gl_cluster_kernel<3", "gl_main_kernel<3"))
c4 = summary["pmc_C4"]
if c4.get("SQ_ACTIVE_INST_VALU") and c4.get("GRBM_GUI_ACTIVE"):
    c4["valu_busy_frac"] = c4["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * c4["GRBM_GUI_ACTIVE"] / 8.0)
    c4["valu_insts_per_pixel"] = c4["SQ_INSTS_VALU"] * 64 / (512 * 65536)
# the simulate() pair at C2: the image-materialising kernels, where the HBM roofline is the bound
sp = {}
for mode, filt in (("fwd", ("gl_pair_kernel<0",)), ("bwd", ("gl_pair_kernel<1",))):
    f, w = pmc("pmc_simpair_fetch", kernel_filter=filt), pmc("pmc_simpair_write", kernel_filter=filt)
    if f.get("FETCH_SIZE") is not None and w.get("WRITE_SIZE") is not None:
        hbm = (2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024  # same units / gfx950 correction as below
        us = 0.5 * (f["_kernel_us"] + w["_kernel_us"])
        sp[mode] = {"kernel_us_under_pmc": us, "hbm_bytes_per_launch": hbm, "hbm_GBps": hbm / (us * 1e-6) / 1e9}
rows_sp = stats_table("kernel_stats_simpair", f"{tag}_simpair_kernel_stats.md", top=6)
if rows_sp:
    for r in rows_sp:
        for mode, key in (("fwd", "gl_pair_kernel<0"), ("bwd", "gl_pair_kernel<1")):
            if key in r["Name"] and mode in sp:
                sp[mode]["kernel_us"] = float(r["AverageNs"]) / 1e3
    if "fwd" in sp and "bwd" in sp and "kernel_us" in sp["fwd"] and "kernel_us" in sp["bwd"]:
        b1 = (8 * 16384 + 8 * 13) * 1024  # B1 = 8N + 8P per sample, C2, 1024 samples
        t = (sp["fwd"]["kernel_us"] + sp["bwd"]["kernel_us"]) * 1e-6
        sp["pair"] = {"algorithmic_bytes_B1": b1, "achieved_GBps": b1 / t / 1e9, "frac_of_8TBps": b1 / t / 8e12,
                      "hbm_bytes_measured": sp["fwd"]["hbm_bytes_per_launch"] + sp["bwd"]["hbm_bytes_per_launch"]}
summary["simulate_pair_C2"] = sp
c3l = summary["pmc_C3L"] = pmc("pmc_C3L", kernel_filter=("gl_shp_normal_kernel", "gl_normal_mfma_kernel"))  # stack-free kernel (round 3) / SYRK
if c3l.get("SQ_VALU_MFMA_BUSY_CYCLES") and c3l.get("GRBM_GUI_ACTIVE"):
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles over all SIMDs (MI355X_MICROARCH.md, PMC units); 1024 SIMDs
    c3l["mfma_busy_frac"] = c3l["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c3l["GRBM_GUI_ACTIVE"] / 8.0)
# round 3: the shapelet kernel of C3 (table mode; per launch of 1024 samples x 128^2)
c3 = {}
for n in ("sq", "mix", "clk", "mem", "fetch", "write"):
    c3.update({k: v for k, v in pmc(f"pmc_C3_{n}", kernel_filter=("gl_shp_kernel<3",)).items() if not k.startswith("_") or n == "clk"})
if c3.get("SQ_INSTS_VALU"):
    px = 1024 * 16384
    cyc = c3.get("GRBM_GUI_ACTIVE", 0) / 8.0
    c3["valu_insts_per_pixel"] = c3["SQ_INSTS_VALU"] * 64 / px
    if cyc:
        c3["kernel_cycles"] = cyc
        c3["valu_busy_frac"] = c3.get("SQ_ACTIVE_INST_VALU", 0) * 4.0 / (1024 * cyc)
        c3["mfma_busy_frac"] = c3.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc)
        c3["lds_active_frac"] = c3.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc)
        c3["ta_busy_frac"] = c3.get("TA_BUSY_avr", 0) / cyc
        c3["l1_accesses_per_cu_cycle"] = c3.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / (256 * cyc)
    if c3.get("FETCH_SIZE") is not None and c3.get("WRITE_SIZE") is not None:
        c3["hbm_bytes_per_launch"] = (2 * c3["FETCH_SIZE"] + c3["WRITE_SIZE"]) * 1024
summary["pmc_C3_shapelet_kernel"] = c3
c3d = pmc("pmc_C3direct", kernel_filter=("gl_shp_kernel<3",))
if c3d.get("SQ_INSTS_VALU") and c3d.get("GRBM_GUI_ACTIVE"):
    c3d["valu_insts_per_pixel"] = c3d["SQ_INSTS_VALU"] * 64 / (1024 * 16384)
    c3d["valu_busy_frac"] = c3d["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * c3d["GRBM_GUI_ACTIVE"] / 8.0)
summary["pmc_C3_direct"] = c3d
for gname in ("grad_accuracy.jsonl", "grad_accuracy_cases.jsonl"):
    gp = os.path.join(src, gname)
    if os.path.exists(gp):
        rows_g = [json.loads(l) for l in open(gp) if l.startswith("{")]
        summary[gname.replace(".jsonl", "")] = [{k: v for k, v in r.items() if k != "columns"} for r in rows_g]
        with open(os.path.join(dst, f"{tag}_{gname}"), "w") as fh:
            for r in rows_g:
                fh.write(json.dumps(r) + "\n")
tr = os.path.join(src, "two_rank_test.log")
if os.path.exists(tr):
    summary["two_rank_real_kernel_test"] = [l.strip() for l in open(tr) if "passed" in l or "failed" in l]
for name in ("map_step_time", "svi_hmc_step_time"):
    f = os.path.join(src, name + ".log")
    if os.path.exists(f):
        summary[name] = [l.strip() for l in open(f) if "ms per" in l]
fetch_kb = summary.get("pmc_fetch", {}).get("FETCH_SIZE")
write_kb = summary.get("pmc_write", {}).get("WRITE_SIZE")
if fetch_kb is not None and write_kb is not None:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide
    # coalesced reads -> double it; WRITE_SIZE is exact.
    summary["traffic_bytes_per_launch"] = (2 * fetch_kb + write_kb) * 1024
sq, clk = summary.get("pmc_sq", {}), summary.get("pmc_clk", {})
if sq.get("SQ_INSTS_VALU") and clk.get("GRBM_GUI_ACTIVE"):
    cycles = clk["GRBM_GUI_ACTIVE"] / 8.0                   # summed over the 8 XCDs
    summary["kernel_cycles"] = cycles
    summary["valu_busy_frac"] = sq["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * cycles)   # quad-cycles -> cycles, 1024 SIMDs
    summary["cycles_per_valu_inst"] = sq["SQ_ACTIVE_INST_VALU"] * 4.0 / sq["SQ_INSTS_VALU"]
    summary["eff_clock_ghz"] = cycles / (clk["_kernel_us"] * 1e3)
for key, name in (("bench", "bench.json"), ("bench_svi", "bench_svi.json"), ("bench_C5", "bench_C5.json"),
                  ("bench_C5_svi", "bench_C5_svi.json")):
    bj = os.path.join(src, name)
    if os.path.exists(bj):
        line = [l for l in open(bj) if l.startswith("{")]
        if line:
            summary[key] = json.loads(line[-1])
# the ISA execution model of bench.py against the hardware count of the same kernel
mix, b = summary.get("pmc_sq", {}), summary.get("bench", {})
isa = (b.get("roofline") or {}).get("isa") or {}
if mix.get("SQ_INSTS_VALU") and isa.get("valu_insts_per_pixel"):
    measured = mix["SQ_INSTS_VALU"] * 64 / (1024 * 16384)
    summary["isa_model_check"] = {"valu_insts_per_pixel_model": isa["valu_insts_per_pixel"], "valu_insts_per_pixel_pmc": measured,
                                  "ratio": isa["valu_insts_per_pixel"] / measured}
cfg = os.path.join(src, "bench_configs.jsonl")
if os.path.exists(cfg):
    summary["configs"] = [json.loads(l) for l in open(cfg) if l.startswith("{")]
json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if not k.startswith("bench") and k != "configs"}, indent=1))
