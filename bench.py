#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X render path.

Metric (BASELINE.json): Mpaths/s (pixels*spp/s), Cornell 1024^2, depth 8 -- cornell_plane_light.scn at
1024x1024, 256 spp, max depth 8 (BASELINE configs[1]); plus the roofline fraction of the dominant kernel.
`--workload config3` (cornell_large_box 2048^2, 1024 spp, depth 16: BASELINE configs[2], the one it asks to tile over
8 GPUs), `config4` and `config5` run the other BASELINE configs through the same code; the default is the headline.

A "step" is one full render of that frame: every sample of every pixel through the trace and the shade+film
kernels, with scene, SPD tables and film resident in HBM when the clock starts. With N > 1 GPUs the frame is
tiled row-cyclically over the ranks (one process per GPU); a rank renders its rows in row blocks and each block's
film (three buffers, one contiguous allocation) is gathered to rank 0 over RCCL while the next block renders; the
step ends when the whole frame is assembled on rank 0. Total work is fixed, so scaling is "strong".

  python bench.py --gpus 1 --steps 3 --warmup 1
  python bench.py --gpus N --steps K --warmup W            (starts the N ranks itself, one child process per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line. `roofline` is the dominant kernel against the bound that holds for it -- f64 vector issue
(see roofline() below and DESIGN.md "Measurement"), with the measured HBM traffic beside it; `cpu_baseline` is the reference's own CPU path (oracle/_ref, when that
library travelled with the repo) or the CPU oracle port, timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

# BASELINE.json configs[1..4] as (scene, image size, samples per pixel, max depth); "@spheres:N" = the synthetic generator
WORKLOADS = {
    "config2": ("cornell_plane_light.scn", 1024, 256, 8, "BASELINE configs[1]"),
    "config3": ("cornell_large_box.scn", 2048, 1024, 16, "BASELINE configs[2]"),
    "config4": ("cornell_gold_mirror.scn", 1024, 512, 8, "BASELINE configs[3]"),
    "config5": ("@spheres:10000", 4096, 64, 8, "BASELINE configs[4]"),
}


def load_workload_scene(scene, W, H):
    import pydrt
    if scene.startswith("@spheres:"):
        return pydrt.synthetic_sphere_scene(int(scene.split(":")[1]), W, H)
    return pydrt.load_scene(os.path.join(REPO, "scenes", scene), W, H)


def _code_only(text):
    """C / C++ source without its comments and with runs of white space collapsed: what the compiler sees. String and character
    literals are kept as they are (a "//" inside one is not a comment)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            while j > 0 and text[j - 1] == "\\":  # a line comment continued by a backslash
                j = text.find("\n", j + 1)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            if out and out[-1] != " ":
                out.append(" ")
            i = n if j < 0 else j + 2
        elif c.isspace():
            if out and out[-1] != " ":
                out.append(" ")
            i += 1
        else:
            out.append(c)
            i += 1
    return "".join(out).strip()


def csrc_sha():
    """sha256 over the kernel sources AS CODE (comments and white space taken out: rewording a comment is not another build): stamps
    profiles/roofline*.json (tools/roofline_from_profiles.py), so that per-path counters taken from an OLDER build of the kernels are
    not multiplied by this build's timings."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(REPO, "daily-ray-trace_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode())
            h.update(_code_only(open(os.path.join(d, name), encoding="utf-8", errors="replace").read()).encode())
    return h.hexdigest()[:16]


def algorithmic_bytes(S, v_int, v_shade, xyz=False):
    """SURVEY 8d byte model per path, split by kernel.
    B_film = 2*8*(S+1) + 2*8*S + 2*8*S           sum(+filter), mean, variance read-modify-write
    B_state = 2*(48 + 8*S + 4 + 8 + 4)            ray, throughput spectrum, pixel id, rng, depth: per closest-hit iteration
    B_rad = 2*8*S                                 radiance spectrum RMW per shaded vertex
    The spectral part (film, radiance, throughput spectrum) is the shade kernel's; the ray/rng/id state is the
    trace kernel's."""
    b_film = 48 if xyz else 2 * 8 * (S + 1) + 2 * 8 * S + 2 * 8 * S  # XYZ-only film: 48 B
    b_state = 2 * (48 + 8 * S + 4 + 8 + 4)
    b_rad = 2 * 8 * S
    shade = b_film + v_shade * b_rad + v_int * (2 * 8 * S)
    trace = v_int * (b_state - 2 * 8 * S)
    return {"path": b_film + v_int * b_state + v_shade * b_rad, "shade": shade, "trace": trace}


def cpu_baseline(bundle_loader, width, height, depth, seconds_target=12.0):
    """Times the CPU path on a bounded sample of the same workload: whole rows of the 1024^2 frame, 1 thread.
    Mpaths/s does not depend on spp, so the sample is 1 spp over as many rows as fit the time budget."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_py as O
    import pydrt
    bundle = bundle_loader()
    use_ref = O.ref_available()
    # calibrate on 8 rows in the middle of the frame, then size the sample
    def run(rows, threads=1):
        p = pydrt.make_params(width, height, spp=1, max_depth=depth, seed=1, y0=(height - rows) // 2, tile_h=rows)
        t0 = time.perf_counter()
        if use_ref and threads == 1:
            O.ref_render_tile(bundle, p)
        else:
            O.oracle_render_tile(bundle, p, math_mode=O.MATH_REFERENCE, num_threads=threads)
        return rows * width / (time.perf_counter() - t0)
    rate = run(8)
    rows = int(max(8, min(height, rate * seconds_target / width)))
    rate = run(rows)
    out = {"value": round(rate / 1e6, 4), "unit": "Mpaths/s", "cores": 1, "kind": "reference" if use_ref else "port",
           "sample": "%d centre rows x %d px, 1 spp, depth %d of the same frame (%d paths), single thread%s" % (
               rows, width, depth, rows * width, ", compiled reference path (oracle/_ref)" if use_ref else ", CPU oracle port")}
    ncpu = os.cpu_count() or 1
    if ncpu > 1:
        rows_mt = int(min(height, rows * min(ncpu, 16)))
        t_rate = run(rows_mt, threads=ncpu)
        out["all_cores"] = {"value": round(t_rate / 1e6, 4), "cores": ncpu, "kind": "port",
                            "sample": "%d rows, %d threads (row-parallel CPU oracle)" % (rows_mt, ncpu)}
    return out


VALU_PEAK_GCYCLES = 1024 * 2.4  # vector-issue cycles available per ns: 256 CUs x 4 SIMDs at the 2.4 GHz peak shader clock
FP64_PEAK_TFLOPS = 78.6         # 1024 SIMDs x 16 f64 lanes per clock x 2 (FMA) x 2.4 GHz (MI355X_MICROARCH.md: half the 157.3 TF f32 vector rate)
FP64_PEAK_NO_FMA_TFLOPS = 39.3  # the same with one flop per lane and clock: the path is compiled -ffp-contract=off (the reference's
                                # operation order is the contract), so a*b+c is two instructions and this is the peak it can reach


def roofline(dominant, kernel_ms, launches, paths_per_launch, model, workload, xyz, profile_name="roofline.json"):
    """The `roofline` object of the JSON line, for the dominant kernel.

    This path has no dense contraction (no MFMA) and keeps its spectra in registers, so neither the matrix peak nor the HBM
    peak bounds it: what bounds it is the rate at which a SIMD issues f64 vector instructions. `achieved` is the kernel's
    vector-pipe busy time per launch (SQ_ACTIVE_INST_VALU x 4 cycles, per path, from the committed rocprofv3 PMC profile of
    this same workload: profiles/roofline.json, regenerated by tools/roofline_from_profiles.py) divided by the launch
    duration measured HERE with HIP events on the launch stream; `peak` is every SIMD issuing every cycle at the peak clock.
    `traffic` is the measured HBM bytes per launch from the same profile and `hbm_measured_frac` what that is of 8 TB/s.
    SURVEY 8d's byte model (a design that streams path state through HBM) is reported under `algorithmic_model`, as a model:
    this design does not move those bytes, so they are never divided by time into a bandwidth."""
    prof = None
    ppath = os.path.join(REPO, "profiles", profile_name)
    stale = None
    if os.path.exists(ppath) and not xyz:
        try:
            pj = json.load(open(ppath))
            if pj.get("workload") == workload:
                prof = pj
                if pj.get("csrc_sha") != csrc_sha():
                    stale = "profiles/%s was taken from kernel sources %s, this build is %s" % (profile_name, pj.get("csrc_sha"), csrc_sha())
        except Exception:
            prof = None
    kernel_ms = dict(kernel_ms)
    if prof and "bounce" in prof["kernels"] and "primary" in prof["kernels"]:
        # scenes behind the hierarchy: the library times the trace STAGE (drt_primary_kernel + drt_bounce_kernel) with one pair of HIP
        # events; the stage's live time is split between the two kernels in the proportion of their traced durations
        tb = prof["kernels"]["bounce"].get("avg_launch_ms_traced", 0.0) * prof["kernels"]["bounce"].get("launches_traced", 0)
        tp_ = prof["kernels"]["primary"].get("avg_launch_ms_traced", 0.0) * prof["kernels"]["primary"].get("launches_traced", 0)
        if tb + tp_ > 0:
            kernel_ms["bounce"] = kernel_ms["trace"] * tb / (tb + tp_)
            kernel_ms["primary"] = kernel_ms["trace"] * tp_ / (tb + tp_)
            del kernel_ms["trace"]
            dominant = max(kernel_ms, key=lambda k: kernel_ms[k])
    avg_ms = {k: (kernel_ms[k] / launches if launches else 0.0) for k in kernel_ms}
    out = {"bound": "fp64_valu", "kernel": "drt_%s_kernel" % dominant, "workload_key": workload, "achieved": None, "peak": round(VALU_PEAK_GCYCLES, 1),
           "unit": "G SIMD-cycles/s of vector issue", "frac": None,
           "frac_is": "vector-issue occupancy: the share of SIMD cycles in which the kernel issues a vector instruction "
                      "(SQ_ACTIVE_INST_VALU x 4 / cycles available) -- how busy the pipe is, NOT useful work / peak; "
                      "useful f64 work against the peak is `fp64` beside it",
           "traffic": None, "hbm_measured_frac": None,
           "launch": {"paths": paths_per_launch, "avg_ms": round(avg_ms[dominant], 4), "count": launches}}
    if prof:
        per_kernel = {}
        for k, e in prof["kernels"].items():
            sec = avg_ms.get(k, 0.0) * 1e-3
            if sec <= 0:
                continue
            d = {"avg_launch_ms": round(avg_ms[k], 4)}
            if "valu_busy_simd_cycles" in e:
                d["valu_busy_Gcycles_per_s"] = round(e["valu_busy_simd_cycles"] * paths_per_launch / sec / 1e9, 1)
                d["valu_busy_frac"] = round(d["valu_busy_Gcycles_per_s"] / VALU_PEAK_GCYCLES, 4)
                d["valu_busy_frac_in_profile"] = e.get("valu_busy_frac")  # against the clock the chip actually held in the PMC pass
            if "lane_efficiency" in e:
                d["lane_efficiency"] = e["lane_efficiency"]
            if "f64_flops" in e:
                tf = e["f64_flops"] * paths_per_launch / sec / 1e12
                d["fp64"] = {"achieved": round(tf, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP64_PEAK_TFLOPS, 4),
                             "peak_no_fma": FP64_PEAK_NO_FMA_TFLOPS, "frac_no_fma": round(tf / FP64_PEAK_NO_FMA_TFLOPS, 4)}
            if "hbm_bytes" in e:
                d["hbm_bytes_per_launch"] = round(e["hbm_bytes"] * paths_per_launch)
                d["hbm_GBs"] = round(e["hbm_bytes"] * paths_per_launch / sec / 1e9, 1)
                d["hbm_measured_frac"] = round(d["hbm_GBs"] / HBM_PEAK_GBS, 4)
            per_kernel[k] = d
        dk = per_kernel.get(dominant, {})
        out.update({"achieved": dk.get("valu_busy_Gcycles_per_s"), "frac": dk.get("valu_busy_frac"),
                    "traffic": dk.get("hbm_bytes_per_launch"), "hbm_measured_frac": dk.get("hbm_measured_frac"),
                    "lane_efficiency": dk.get("lane_efficiency"), "fp64": dk.get("fp64"), "per_kernel": per_kernel,
                    "counters_from": "profiles/%s <- %s" % (profile_name, prof.get("source", ""))})
        if stale:
            # per-path counters of another build times this build's timings would be a number about neither
            out.update({"achieved": None, "frac": None, "stale_profile": stale})
    else:
        out["note"] = "no committed PMC profile matches this workload (profiles/%s): vector-issue and HBM fractions not reported" % profile_name
    out["algorithmic_model"] = {
        "what": "SURVEY 8d byte model of a design that streams path state through HBM; a MODEL, not traffic -- this design keeps "
                "throughput and radiance spectra in registers, so these bytes are not moved and are not a bandwidth",
        "bytes_per_path": {k: round(v, 1) for k, v in model.items()},
        "bytes_per_launch": round(model.get(dominant, model["trace"]) * paths_per_launch),
        "measured_traffic_over_model": (round(out["traffic"] / (model.get(dominant, model["trace"]) * paths_per_launch), 4) if out["traffic"] else None)}
    return out


def oneshot_child(size, spp, depth, scene="cornell_plane_light.scn"):
    """Fresh process: the drop-in call itself -- drt_render_tile() with caller-owned host buffers, zero-filled as the reference's
    alloc() leaves them (DRT_FLAG_FILM_ZERO), film copied back to the host -- timed wall-clock, PCIe and context set-up included."""
    import numpy as np
    import pydrt
    bundle = load_workload_scene(scene, size, size)
    t0 = time.perf_counter()
    pydrt.render_tile(bundle, pydrt.make_params(64, 64, spp=1, max_depth=depth, seed=1))  # HIP runtime and code object loaded
    init_ms = (time.perf_counter() - t0) * 1e3
    runs = []
    for _ in range(2):
        p = pydrt.make_params(size, size, spp=spp, max_depth=depth, seed=1, flags=pydrt.FLAG_FILM_ZERO)
        t0 = time.perf_counter()
        px, av, va, st = pydrt.render_tile(bundle, p)
        dt = time.perf_counter() - t0
        assert st.paths == size * size * spp and float(px[0, bundle.S]) == float(spp)
        runs.append({"wall_ms": round(dt * 1e3, 1), "Mpaths_per_s": round(size * size * spp / dt / 1e6, 1), "device_kernel_ms": round(st.total_ms, 1)})
        del px, av, va
    print(json.dumps({"oneshot": runs, "runtime_init_ms": round(init_ms, 1)}), flush=True)


def oneshot_leg(size, spp, depth, scene="cornell_plane_light.scn"):
    """The same workload through the one-shot C-ABI call, in a child process of its own (a fresh HIP context, as a caller of the
    drop-in would have): value = the better of two calls."""
    import subprocess

    def fresh_process():
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--oneshot-child", "--size", str(size), "--spp", str(spp), "--depth", str(depth),
                              "--scene", scene], capture_output=True, text=True, timeout=900)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not lines:
            return None, (out.stderr or out.stdout)[-300:]
        return json.loads(lines[-1]), None

    child, err = fresh_process()
    if child is None:
        return {"error": err}
    # A box whose GPU another process has just left (a test run before this bench, another tenant): the driver scrubs the memory that
    # process freed before it hands any of it out again, several seconds for a few hundred GB, and the first allocation of the next
    # process waits for it. That is the earlier process's bill, not this call's: when the first call of the first fresh process took
    # far longer than its second (which a clean first call never does: 1.15 x), the leg is measured again in ANOTHER fresh process,
    # and the first one is kept in the line as what it was.
    waited = None
    first = child["oneshot"]
    if first[0]["wall_ms"] > 1.5 * first[-1]["wall_ms"] + 50.0:
        again, err = fresh_process()
        if again is not None:
            waited = {"first_call_ms": first[0]["wall_ms"], "second_call_ms": first[-1]["wall_ms"],
                      "why": "this process's first allocation waited for the driver's scrub of memory an earlier process had freed; measured again in another fresh process"}
            child = again
    runs = child["oneshot"]
    # The reference's main() calls render_image() ONCE (src/win32_main.c:146), so the call a maintainer sees is the FIRST one of a
    # process: `cold` (device memory the process touches for the first time is cleared by the driver, pages are faulted in) is the
    # value; `warm` (the same call again in the same process) is beside it.
    cold, warm = runs[0], runs[-1]
    return {"value": cold["Mpaths_per_s"], "unit": "Mpaths/s", "wall_ms": cold["wall_ms"], "device_kernel_ms": cold["device_kernel_ms"],
            "cold": cold, "warm": warm, "runs": runs,
            "runtime_init_ms": child.get("runtime_init_ms"),  # a 64x64 call before the timed ones: HIP runtime start-up and code object load, which any GPU program pays once
            "cold_with_runtime_init_ms": round(cold["wall_ms"] + (child.get("runtime_init_ms") or 0.0), 1),
            "earlier_fresh_process": waited,
            "what": "drt_render_tile(): host film buffers (zero-filled, DRT_FLAG_FILM_ZERO), context creation, kernels, film download over PCIe; "
                    "a fresh process started BEFORE this one touches the GPU (memory another process has just freed is scrubbed by the driver "
                    "before it is handed out again, which is not the call's cost); value = that process's FIRST full call (cold); never the headline `value`"}


def self_launch(n):
    """`python bench.py --gpus N` from a plain shell: start the N ranks as fresh child processes (one per GPU, torch's own
    launcher, rendezvous on 127.0.0.1) and relay rank 0's JSON line. This parent never imports torch or touches the GPU, so
    nothing that has initialised HIP is ever exec'ed or forked; a rank that fails makes the whole call fail."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for line in child.stdout:
        if line.startswith("{"):
            lines.append(line.rstrip("\n"))
        else:
            sys.stderr.write(line)
    rc = child.wait()
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank launch failed (exit code %d)\n" % (n, rc))
        return rc
    if len(lines) != 1:
        sys.stderr.write("bench.py: expected one JSON line from rank 0, got %d\n" % len(lines))
        return 1
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS),
                    help="which BASELINE config: config2 = the headline (configs[1], cornell_plane_light 1024^2 x 256 spp, depth 8); "
                         "config3 = configs[2] (cornell_large_box 2048^2 x 1024 spp, depth 16, the one tiled over 8 GPUs); config4; config5")
    ap.add_argument("--size", type=int, default=0, help="image width = height (0: the workload's)")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (0: the workload's)")
    ap.add_argument("--depth", type=int, default=0, help="max depth (0: the workload's)")
    ap.add_argument("--scene", default="", help=argparse.SUPPRESS)
    ap.add_argument("--batch", type=int, default=0, help="samples per kernel pair (0 = library default)")
    ap.add_argument("--gather-blocks", type=int, default=0, help="row blocks per rank (0 = 1 at N=1, 4 at N>1)")
    ap.add_argument("--block-streams", type=int, default=1, help="HIP streams the row blocks are spread over (blocks on different streams overlap)")
    ap.add_argument("--film", default="spectral", choices=["spectral", "xyz"],
                    help="spectral = the reference's film (the headline metric); xyz = DRT_MODE_XYZ, a different mode, labelled as such")
    ap.add_argument("--checksum", action="store_true", help="add an order-independent checksum of the assembled frame to the JSON line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oneshot", action="store_true", help="skip the one-shot (PCIe-inclusive) leg")
    ap.add_argument("--oneshot-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal only: all ranks use GPU 0 (needs --backend gloo)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL across processes needs dmabuf IPC on this driver
    wl_scene, wl_size, wl_spp, wl_depth, wl_label = WORKLOADS[args.workload]
    args.size = args.size or wl_size
    args.spp = args.spp or wl_spp
    args.depth = args.depth or wl_depth
    if args.oneshot_child:
        return oneshot_child(args.size, args.spp, args.depth, args.scene or wl_scene)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    # the one-shot (PCIe-inclusive) leg first, in a child process of its own, while this process holds nothing on the GPU
    oneshot = None
    if not args.no_oneshot and args.gpus == 1 and args.film == "spectral":
        oneshot = oneshot_leg(args.size, args.spp, args.depth, wl_scene)
    import torch
    import pydrt
    import drt_dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one process per GPU (or leave WORLD_SIZE unset and bench.py starts them itself)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    def die(stage, err):
        """A failing rank fails the run, loudly: no fallback to another backend, no JSON line."""
        sys.stderr.write("bench.py: rank %d of %d (GPU %d): %s failed with backend %r: %s: %s\n" % (rank, world, local_rank, stage, args.backend, type(err).__name__, err))
        sys.stderr.flush()
        os._exit(3)

    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(args.backend, rank=rank, world_size=world)
        except Exception as err:  # noqa: BLE001 -- whatever the backend raises, the run is over
            die("init_process_group", err)

    W = H = args.size
    bundle = load_workload_scene(wl_scene, W, H)
    S = bundle.S
    dev = torch.device("cuda", local_rank)
    n_tile = drt_dist.rank_rows(H, rank, world)[1] * W
    # The rank's rows (cyclic over the ranks) in row blocks: each block is one contiguous film buffer (three regions)
    # with its own render context; block b's gather to rank 0 is in flight while block b+1 renders.
    # Blocks of >= 128k pixels keep the kernels' last rounds short; at least 2 so a gather can hide.
    # (from the LARGEST tile, so that every rank arrives at the same number of blocks = the same sequence of collectives)
    max_tile = drt_dist.max_tile_rows(H, world) * W
    n_blocks = args.gather_blocks if args.gather_blocks > 0 else (max(2, min(8, max_tile // (128 * 1024))) if world > 1 else 1)
    xyz = args.film == "xyz"
    blocks = drt_dist.film_blocks(H, W, S, rank, world, dev, n_blocks, channels=(8,) if xyz else None)
    # One explicit (non-default) stream carries the film zero-fill, the kernels, the collectives' dependencies and the
    # de-interleave copies. The default stream's handle is 0, which drt_set_stream() reads as "the context's own stream":
    # the gather would then not wait for the render.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    if world > 1:
        # the kernels are persistent and fill the chip: leave a few workgroup slots free so that the gather of the previous block
        # (RCCL's kernels) and rank 0's de-interleave copies start at once instead of at the next kernel boundary
        os.environ.setdefault("DRT_RESERVE_BLOCKS", "16")
    block_streams = [stream] + [torch.cuda.Stream(device=dev) for _ in range(max(1, args.block_streams) - 1)]
    renderers = []
    for b, fb in enumerate(blocks):
        by0, brows, bstride = fb.tile()
        if brows == 0:
            renderers.append(None)
            continue
        # contexts live across all the steps, so launches are sized for throughput (DRT_BATCH_RESIDENT: 64 M paths per kernel pair,
        # at least 16 samples per pixel), not for the one frame `spp` announces
        batch = args.batch if args.batch > 0 else pydrt.BATCH_RESIDENT
        params = pydrt.make_params(W, H, spp=args.spp, max_depth=args.depth, seed=1, y0=by0, tile_h=brows, row_stride=bstride,
                                   device=local_rank, batch_spp=batch, mode=pydrt.MODE_XYZ if xyz else pydrt.MODE_SPECTRAL)
        r = pydrt.Renderer(bundle, params)
        if xyz:
            r.bind_film(fb.region(0).data_ptr(), None, None)
        else:
            r.bind_film(fb.region(0).data_ptr(), fb.region(1).data_ptr(), fb.region(2).data_ptr())
        r.set_stream(block_streams[b % len(block_streams)].cuda_stream)
        renderers.append(r)
    live = [r for r in renderers if r is not None]

    staging = args.backend != "nccl"  # gloo rehearsal: collectives on host copies

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gather_ms = [0.0]

    side = torch.cuda.Stream(device=dev)  # rank 0 de-interleaves a gathered block here while the next block renders

    def finish_on_side(fb):
        with torch.cuda.stream(side):
            fb.finish()

    def step():
        for bs in block_streams[1:]:
            bs.wait_stream(stream)
        prev = None
        for b, (fb, r) in enumerate(zip(blocks, renderers)):
            with torch.cuda.stream(block_streams[b % len(block_streams)]):
                fb.zero_()
                if r is not None:
                    r.render(0, args.spp)
                if world > 1:
                    fb.gather_async(staging=staging)
            if world > 1:
                if prev is not None:
                    finish_on_side(prev)
                prev = fb
        for bs in block_streams[1:]:
            stream.wait_stream(bs)
        if world > 1:
            # what is left of the gathers once the last block has rendered = the part not hidden behind compute
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            finish_on_side(prev)
            stream.wait_stream(side)  # the next step's buffers, and the end of the timed region, wait for the frame
            t1.record()
            torch.cuda.synchronize()
            gather_ms[0] += t0.elapsed_time(t1)

    def all_stats():
        for r in live:
            r.synchronize()
        sts = [r.stats() for r in live]
        out = {k: sum(getattr(st, k) for st in sts) for k in ("paths", "trace_ms", "shade_ms", "closest_hit_scans", "shaded_vertices", "redone_launches", "launches")}
        out["pool_bytes"] = sum(st.record_pool_blocks * st.record_block_bytes for st in sts)
        out["pool_peak_bytes"] = max(st.record_pool_peak * st.record_block_bytes for st in sts)
        return out

    try:
        if world > 1:
            # set up the communicator and its point-to-point connections (made lazily on first use) outside the timed region,
            # whatever --warmup is: one gather of the same shape as a block's
            blocks[0].gather_async(staging=staging)
            finish_on_side(blocks[0])
            stream.wait_stream(side)
            torch.cuda.synchronize()
    except Exception as err:  # noqa: BLE001
        die("the first gather (communicator set-up)", err)
    try:
        for _ in range(args.warmup):
            step()
        barrier()
        st0 = all_stats()
        gather_ms[0] = 0.0
        t_start = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        local_elapsed = time.perf_counter() - t_start  # this rank's own steps, before it waits for the others
        barrier()
        elapsed = time.perf_counter() - t_start
        st1 = all_stats()
    except Exception as err:  # noqa: BLE001
        die("rendering / gathering", err)
    batch_spp = live[0].batch_spp()
    block_pixels = blocks[0].rows * W
    per_rank = None
    if world > 1:
        try:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if staging else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            # every rank's own figures, per step: kernel time on its GPU (HIP events), wall time of its steps before the barrier,
            # and how long it waited for gathers that rendering did not hide
            mine = torch.tensor([(st1["trace_ms"] - st0["trace_ms"] + st1["shade_ms"] - st0["shade_ms"]) / args.steps,
                                 local_elapsed * 1e3 / args.steps, gather_ms[0] / args.steps,
                                 float(drt_dist.rank_rows(H, rank, world)[1])], dtype=torch.float64, device="cpu" if staging else dev)
            everyone = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(everyone, mine)
            per_rank = [[float(x) for x in e.cpu()] for e in everyone]
        except Exception as err:  # noqa: BLE001
            die("the timing reductions", err)

    # kernel time from HIP events recorded by the library on the launch stream (timed region only)
    paths_rank = st1["paths"] - st0["paths"]
    trace_ms = st1["trace_ms"] - st0["trace_ms"]
    shade_ms = st1["shade_ms"] - st0["shade_ms"]
    v_int = (st1["closest_hit_scans"] - st0["closest_hit_scans"]) / max(paths_rank, 1)
    v_shade = (st1["shaded_vertices"] - st0["shaded_vertices"]) / max(paths_rank, 1)

    if rank == 0:
        total_paths = W * H * args.spp * args.steps
        value = total_paths / elapsed / 1e6
        model = algorithmic_bytes(S, v_int, v_shade, xyz)
        dominant = "shade" if shade_ms >= trace_ms else "trace"
        # per launch: paths per kernel launch and its average duration (launches = batches)
        # kernel pairs the library launched in the timed region (drt_stats.launches) and the paths of an average one: a long call on a
        # large frame goes out in row blocks whose pairs take other sample counts than batch_spp x the block's pixels
        launches = max(1, int(st1["launches"] - st0["launches"]))
        paths_per_launch = max(1, int(round(paths_rank / launches)))
        paths_per_launch_all = batch_spp * sum(fb.rows for fb in blocks) * W  # this rank's launches side by side (one context per row block)
        kernel_ms = {"trace": trace_ms, "shade": shade_ms}
        scene_stem = wl_scene.replace(".scn", "").replace("@", "").replace(":", "_")
        roof = roofline(dominant, kernel_ms, launches, paths_per_launch, model,
                        "%s %dx%d depth %d" % (scene_stem, W, H, args.depth), xyz,
                        "roofline.json" if args.workload == "config2" else "roofline_%s.json" % args.workload)
        roof.update({"kernel_ms_per_step": {"trace": round(trace_ms / args.steps, 3), "shade": round(shade_ms / args.steps, 3)},
                     "v_int": round(v_int, 4), "v_shade": round(v_shade, 4)})
        if st1["redone_launches"]:
            raise SystemExit("bench.py: %d launches ran out of record blocks and were rendered again: the timing is not the path's" % st1["redone_launches"])
        out = {
            "metric": "Mpaths/s (pixels*spp/s) Cornell 1024^2 depth 8; achieved HBM GB/s % of peak" + (" [XYZ-only film: NOT the headline mode]" if xyz else "") +
                      ("" if args.workload == "config2" else " [workload %s: NOT the headline config]" % args.workload),
            "value": round(value, 2), "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s %dx%d, %d spp, depth %d (%s%s)" % (wl_scene, W, H, args.spp, args.depth, wl_label,
                                                                             "" if (W, args.spp, args.depth) == (wl_size, wl_spp, wl_depth) else ", REDUCED from its stated size"),
                       "film": ("XYZ only (DRT_MODE_XYZ: 8 accumulators per pixel, no mean/variance)" if xyz else
                                "full spectral (sum+filter, mean, variance x %d wavelengths)" % S),
                       "partition": ("whole frame on one GPU" if world == 1 and len(blocks) == 1 else
                                     "rows cyclic over %d rank(s) in %d row block(s); each block's film gathered to rank 0 while the next renders" % (world, len(blocks))),
                       "paths_per_step": W * H * args.spp,
                       "record_pool_GB": round(st1["pool_bytes"] / 1e9, 2), "record_pool_peak_GB": round(st1["pool_peak_bytes"] / 1e9, 2),
                       "record_pool_worst_case_GB": round(paths_per_launch_all * args.depth * 128 / 1e9, 2)},
            "roofline": roof,
        }
        if world > 1:
            out["gather_ms_per_step"] = round(gather_ms[0] / args.steps, 3)  # the part not hidden behind rendering (rank 0)
            k_ms = [r[0] for r in per_rank]
            out["per_rank_ms"] = {"kernels": [round(x, 3) for x in k_ms], "wall": [round(r[1], 3) for r in per_rank],
                                  "gather_wait": [round(r[2], 3) for r in per_rank], "rows": [int(r[3]) for r in per_rank],
                                  "what": "per step and rank: trace + shade kernel time on the rank's GPU (HIP events), wall time of its own steps "
                                          "(gathers and rank 0's frame assembly included), the part of the gathers rendering did not hide, image rows owned"}
            out["load_imbalance"] = round(max(k_ms) / (sum(k_ms) / len(k_ms)), 4) if sum(k_ms) > 0 else None  # slowest rank's kernel time / the mean
            # rank 0 holds the assembled frame and one receive buffer per row block beside its own share of the work
            frame_b = sum(int(t.numel()) * 8 for t in blocks[0].image)
            recv_b = sum(int(fb.recv.numel()) * 8 for fb in blocks if fb.recv is not None)
            out["config"]["rank0_frame_GB"] = round(frame_b / 1e9, 2)
            out["config"]["rank0_recv_buffers_GB"] = round(recv_b / 1e9, 2)
            out["config"]["collective"] = "torch.distributed.gather, backend %s (nccl = RCCL over xGMI), one per row block, async behind the next block's kernels" % args.backend
        if args.checksum:
            # wrap-around int64 sum of the bit patterns: exact and independent of the order of pixels
            if world == 1:
                for fb in blocks:
                    fb.finish()
            torch.cuda.synchronize()
            out["frame_checksum"] = [int(t.view(torch.int64).sum().item()) for t in blocks[0].image]
        if not args.no_cpu_baseline:
            # rank 0 only, after the timed region (the other ranks wait at the final barrier)
            out["cpu_baseline"] = cpu_baseline(lambda: load_workload_scene(wl_scene, W, H), W, H, args.depth)
    for r in live:
        r.close()
    if rank == 0:
        if oneshot is not None:
            out["oneshot"] = oneshot
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
