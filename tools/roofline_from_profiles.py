#!/usr/bin/env python3
"""Regenerates the roofline inputs of bench.py from rocprofv3 CSVs.

    python3 tools/roofline_from_profiles.py <profile dir> --paths-per-launch N --workload "..." [--out profiles/rNN_roofline.json]

<profile dir> is what tools/profile_bench.sh wrote: one sub-directory per rocprofv3 pass over the SAME bench command
(trace/: --kernel-trace --stats; pmc_*/: one --pmc pass each, counters never mixed with tracing). For the two kernels
of the path it folds the per-dispatch rows into PER-PATH figures (the workload is deterministic, so per-path counts
carry over to any launch size of the same workload):

  valu_busy_simd_cycles   SQ_ACTIVE_INST_VALU x 4   (the counter is in quad-cycles, summed over waves; one SIMD issues
                                                      one vector instruction at a time, so this is vector-pipe busy time)
  simd_cycles_available   SQ_BUSY_CYCLES x 32        (the counter sums the 32 shader engines' busy cycles; 1024 SIMDs)
  valu_busy_frac          the quotient of the two    -- the bound that holds for this path: f64 VALU issue
  lane_efficiency         SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
  f64_flops               64 x lane_efficiency x (ADD_F64 + MUL_F64 + TRANS_F64 + 2 FMA_F64) wave-instructions
  hbm_bytes               FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024   (KiB units; FETCH x2 on gfx950: MI355X_MICROARCH.md, HBM)
  clock_ghz               GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the same pass

bench.py divides the per-path figures by the kernel time it measures live (HIP events).
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys

KERNELS = {"shade": "drt_shade_kernel", "trace": "drt_trace_kernel", "primary": "drt_primary_kernel", "bounce": "drt_bounce_kernel"}
N_SIMD = 1024          # 256 CUs x 4 SIMDs
N_SE = 32              # SQ_BUSY_CYCLES is summed over the shader engines
N_XCD = 8              # GRBM_GUI_ACTIVE is summed over the XCDs


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def kernel_of(name):
    for k, needle in KERNELS.items():
        if needle in name:
            return k
    return None


def read_trace(root):
    """per kernel: calls, average ns (from --kernel-trace --stats)"""
    out = {}
    for f in find(root, "*kernel_stats.csv"):
        for row in csv.DictReader(open(f)):
            k = kernel_of(row.get("Name", ""))
            # two instantiations of one kernel may show up (drt_create's sizing launch is one call of a fraction of a millisecond): the
            # one the time goes to is the one the workload runs
            if k and (k not in out or float(row["TotalDurationNs"]) > out[k]["total_ns"]):
                out[k] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "total_ns": float(row["TotalDurationNs"]),
                          "name": row["Name"].split("(")[0]}
    return out


def read_counters(root):
    """per kernel and counter: sum over dispatches; plus dispatch count, register counts and in-pass kernel time per counter"""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    span = collections.defaultdict(lambda: collections.defaultdict(dict))
    regs = {}
    for f in find(root, "*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = kernel_of(row.get("Kernel_Name", ""))
            if not k:
                continue
            c = row["Counter_Name"]
            agg[k][c] += float(row["Counter_Value"] or 0)
            disp[k][c].add((f, row["Dispatch_Id"]))
            span[k][c][(f, row["Dispatch_Id"])] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            # (rocprofv3's VGPR_Count column reads half the allocated registers on gfx950 -- 64 for a 128-register kernel -- so
            #  it is left out; the ISA's .vgpr_count is quoted in DESIGN.md instead)
            regs[k] = {"sgpr": int(row["SGPR_Count"]), "scratch_bytes_per_lane": int(row["Scratch_Size"]),
                       "lds_bytes_per_workgroup": int(row["LDS_Block_Size"])}
    return agg, disp, span, regs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--paths-per-launch", type=int, default=0)
    ap.add_argument("--workload", default="")
    ap.add_argument("--bench-log", default="", help="a log holding bench.py's JSON line of the profiled command: paths per launch and workload key are taken from it")
    ap.add_argument("--source", default="")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    if a.bench_log:
        line = [l for l in open(a.bench_log) if l.startswith("{")][-1]
        roof = json.loads(line)["roofline"]
        a.paths_per_launch = a.paths_per_launch or int(roof["launch"]["paths"])
        a.workload = a.workload or roof["workload_key"]
    if not a.paths_per_launch or not a.workload:
        sys.exit("give --bench-log, or --paths-per-launch and --workload")
    trace = read_trace(a.root)
    agg, disp, span, regs = read_counters(a.root)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import csrc_sha  # the kernel sources these counters were taken from: bench.py drops `frac` when its build is another one
    out = {"workload": a.workload, "paths_per_launch": a.paths_per_launch, "source": a.source or os.path.relpath(a.root), "csrc_sha": csrc_sha(),
           "units": "per path unless a key says otherwise; cycles are shader-clock cycles", "kernels": {}}
    for k in KERNELS:
        if k not in agg:
            continue
        c = agg[k]

        def per_path(counter, scale=1.0):
            n = len(disp[k][counter])
            return c[counter] * scale / (n * a.paths_per_launch) if n else None

        e = {"registers": regs.get(k)}
        if k in trace:
            e["launches_traced"] = trace[k]["calls"]
            e["avg_launch_ms_traced"] = round(trace[k]["avg_ns"] / 1e6, 4)
            e["kernel"] = trace[k]["name"]
        busy = per_path("SQ_ACTIVE_INST_VALU", 4.0)
        avail = per_path("SQ_BUSY_CYCLES", float(N_SIMD) / N_SE)
        if busy is not None and avail:
            e["valu_busy_simd_cycles"] = round(busy, 3)
            e["simd_cycles_available"] = round(avail, 3)
            e["valu_busy_frac"] = round(busy / avail, 4)
        if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
            # both counters must come from the same pass for the quotient to mean anything
            e["lane_efficiency"] = round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4)
        for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                     "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"):
            v = per_path(name)
            if v is not None:
                e.setdefault("wave_instructions", {})[name[len("SQ_INSTS_"):]] = round(v, 4)
        wi = e.get("wave_instructions", {})
        if "VALU_ADD_F64" in wi and "lane_efficiency" in e:
            ops = wi["VALU_ADD_F64"] + wi["VALU_MUL_F64"] + wi.get("VALU_TRANS_F64", 0.0) + 2.0 * wi["VALU_FMA_F64"]
            e["f64_flops"] = round(64.0 * e["lane_efficiency"] * ops, 1)
            e["f64_flops_note"] = "wave-instructions x 64 lanes x lane_efficiency (the kernel-wide active-lane share); FMA counts 2"
        for name, key in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_ACTIVE_INST_ANY", "active_inst_any"),
                          ("SQ_WAVE_CYCLES", "wave_cycles")):
            v = per_path(name, 4.0)
            if v is not None:
                e.setdefault("wave_quad_counters_x4", {})[key] = round(v, 2)
        fetch, write = per_path("FETCH_SIZE", 1024.0 * 2.0), per_path("WRITE_SIZE", 1024.0)
        if fetch is not None and write is not None:
            e["hbm_bytes"] = round(fetch + write, 2)
            e["hbm_bytes_read"] = round(fetch, 2)
            e["hbm_bytes_written"] = round(write, 2)
        if c.get("GRBM_GUI_ACTIVE"):
            t = sum(span[k]["GRBM_GUI_ACTIVE"].values())
            if t > 0:
                e["clock_ghz_in_pmc_pass"] = round(c["GRBM_GUI_ACTIVE"] / N_XCD / t, 3)
        out["kernels"][k] = e
    text = json.dumps(out, indent=1)
    if a.out:
        open(a.out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    sys.exit(main())
