#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/prof_*) into the small, committed files under profiles/:

  profiles/<tag>_kernel_stats.csv     the --stats per-kernel table (kernel names shortened)
  profiles/<tag>_pmc.json             per-kernel means of every collected counter, with the gfx950 corrections of
                                      MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB,
                                      FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane streaming reads (x2)
  profiles/pmc_traffic.json           HBM bytes per launch of the dominant kernel, keyed "<shape>:<dims>:<gpus>:<kernel>",
                                      which bench.py reports as roofline.traffic
"""
import argparse
import collections
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:80]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats", default=None, help="directory of a --kernel-trace --stats run")
    ap.add_argument("--pmc", nargs="*", default=[], help="directories of --pmc runs")
    ap.add_argument("--kernel", default="stencil", help="substring of the kernels of interest")
    ap.add_argument("--traffic-key", default=None, help='e.g. "star2d1r:16384x16384:1:stencil2d_direct_kernel"')
    args = ap.parse_args()
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)

    if args.stats:
        f = glob.glob(os.path.join(args.stats, "**", "*_kernel_stats.csv"), recursive=True)[0]
        rows = list(csv.DictReader(open(f)))
        out = os.path.join(prof, f"{args.tag}_kernel_stats.csv")
        with open(out, "w", newline="") as fo:
            w = csv.writer(fo)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
        print("wrote", out)

    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    durations = collections.defaultdict(list)
    for d in args.pmc:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if args.kernel in r["Kernel_Name"]:
                    k = short(r["Kernel_Name"])
                    counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    durations[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if counters:
        summary = {}
        for k, cs in counters.items():
            e = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}
            e["_mean_duration_ns_under_pmc"] = sum(durations[k]) / len(durations[k])
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                fetch = e["FETCH_SIZE"]["mean"] * 1024 * 2  # KiB, x2: gfx950 wide-read correction
                write = e["WRITE_SIZE"]["mean"] * 1024
                e["hbm_read_bytes_per_launch"] = fetch
                e["hbm_write_bytes_per_launch"] = write
                e["hbm_bytes_per_launch"] = fetch + write
            # pipe utilisation (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles summed over waves,
            # SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs)
            if "GRBM_GUI_ACTIVE" in cs and "SQ_WAVE_CYCLES" in cs:
                cyc = e["GRBM_GUI_ACTIVE"]["mean"] / 8.0             # shader cycles of the launch
                simd_quads = cyc / 4.0 * 1024.0                       # quad-cycles available on 256 CUs x 4 SIMDs
                e["clock_ghz"] = cyc / e["_mean_duration_ns_under_pmc"]
                e["waves_per_simd"] = e["SQ_WAVE_CYCLES"]["mean"] / simd_quads
                if "SQ_ACTIVE_INST_VALU" in cs:
                    e["valu_busy"] = e["SQ_ACTIVE_INST_VALU"]["mean"] / simd_quads
                if "SQ_INSTS_VALU_FMA_F64" in cs:
                    # fp64 arithmetic instructions x 4 cycles each (16 lanes per cycle) of the SIMD cycles available
                    # (SQ_INSTS_VALU_FMA_F64 counts v_fma / v_mul / v_add _f64)
                    e["fp64_issue_frac"] = e["SQ_INSTS_VALU_FMA_F64"]["mean"] * 4.0 / (cyc * 1024.0)
                if "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
                    e["mfma_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (cyc * 1024.0)
                if "SQ_LDS_IDX_ACTIVE" in cs:
                    # LDS-array cycles summed over the CUs, of the cycles 256 CUs had
                    e["lds_active"] = e["SQ_LDS_IDX_ACTIVE"]["mean"] / (cyc * 256.0)
                if "SQ_WAIT_INST_ANY" in cs:
                    e["wait_inst_frac"] = e["SQ_WAIT_INST_ANY"]["mean"] / e["SQ_WAVE_CYCLES"]["mean"]
                if "SQ_WAIT_ANY" in cs:
                    e["wait_any_frac"] = e["SQ_WAIT_ANY"]["mean"] / e["SQ_WAVE_CYCLES"]["mean"]
            # EXECUTED arithmetic, wave-instructions per launch by kind (SQ_INSTS_VALU_FMA_F64 counts v_fma_f64 only: the
            # nested-profile 2D kernel reads 11.2 per point = its 10 FMAs x the halo lanes; adds and multiplies are their own
            # counters).  Flops = 64 lanes x (2 per FMA, 1 per add / multiply).
            kinds = {}
            for prec in ("F64", "F32"):
                for op in ("FMA", "ADD", "MUL"):
                    c = f"SQ_INSTS_VALU_{op}_{prec}"
                    if c in cs:
                        kinds[f"{op.lower()}_{prec.lower()}"] = e[c]["mean"]
            if kinds:
                e["valu_insts_by_kind"] = kinds
            if "SQ_INSTS_VALU" in cs:
                e["valu_insts"] = e["SQ_INSTS_VALU"]["mean"]
            summary[k] = e
        out = os.path.join(prof, f"{args.tag}_pmc.json")
        json.dump(summary, open(out, "w"), indent=1, sort_keys=True)
        print("wrote", out)
        if args.traffic_key:
            tpath = os.path.join(prof, "pmc_traffic.json")
            t = json.load(open(tpath)) if os.path.exists(tpath) else {}
            best = max(summary.values(), key=lambda e: e.get("hbm_bytes_per_launch", 0))
            if "hbm_bytes_per_launch" in best:
                t[args.traffic_key] = {"hbm_bytes": round(best["hbm_bytes_per_launch"]),
                                       "valu_busy": round(best["valu_busy"], 3) if "valu_busy" in best else None,
                                       "mfma_busy": round(best["mfma_busy"], 3) if "mfma_busy" in best else None,
                                       "waves_per_simd": round(best["waves_per_simd"], 2) if "waves_per_simd" in best else None,
                                       "fp64_issue_frac": round(best["fp64_issue_frac"], 3) if "fp64_issue_frac" in best else None,
                                       "clock_ghz": round(best["clock_ghz"], 3) if "clock_ghz" in best else None,
                                       "lds_active": round(best["lds_active"], 3) if "lds_active" in best else None,
                                       "wait_inst_frac": round(best["wait_inst_frac"], 3) if "wait_inst_frac" in best else None,
                                       "valu_insts": round(best["valu_insts"]) if "valu_insts" in best else None,
                                       "valu_insts_by_kind": ({k: round(v) for k, v in best["valu_insts_by_kind"].items()}
                                                              if "valu_insts_by_kind" in best else None),
                                       "source": f"profiles/{args.tag}_pmc.json"}
                json.dump(t, open(tpath, "w"), indent=1, sort_keys=True)
                print("traffic", args.traffic_key, t[args.traffic_key])


if __name__ == "__main__":
    main()
