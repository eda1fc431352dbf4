"""Summarise gpurun_out/prof_<tag>_{stats,fetch,write} into profiles/ (tracked):
  profiles/<tag>_kernel_stats.csv  - rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_pcg_launches.csv  - per (kernel, grid) launch durations of the PCG kernels from the trace
  profiles/pmc_traffic.json        - HBM bytes per PCG launch per bench workload from the PMC passes
    python tools/summarize_profile.py r01
"""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
# GATO_PROFILE_OUT: summarise ON the GPU box into a directory under gpurun_out/ (the raw traces are far beyond what a
# gpurun call copies back), then move the files into profiles/ here
out = os.environ.get("GATO_PROFILE_OUT") or os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def short(name):
    return re.sub(r"^void gato::\(anonymous namespace\)::", "", name).split("(")[0]


def one(pattern):
    g = glob.glob(os.path.join(ROOT, "gpurun_out", pattern))
    return max(g, key=os.path.getmtime) if g else None          # several runs may have been merged: newest wins


stats = one(f"prof_{tag}_stats/*/*_kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
trace = one(f"prof_{tag}_stats/*/*_kernel_trace.csv")
launch = defaultdict(list)
if trace:
    for r in csv.DictReader(open(trace)):
        if ("pcg_" in r["Kernel_Name"] or "stream_step" in r["Kernel_Name"]) and not re.search(r"pcg_resident_kernel<\w+, \d+, \d+, \d+, true,", r["Kernel_Name"]):
            key = (short(r["Kernel_Name"]),
                   int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]),
                   int(r["Workgroup_Size_X"] if "Workgroup_Size_X" in r else r["Workgroup_Size"]))
            launch[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(out, f"{tag}_pcg_launches.csv"), "w") as f:
        f.write("kernel,grid_threads,workgroup_threads,launches,avg_ns,min_ns,max_ns\n")
        for k, v in sorted(launch.items()):
            # one-XCD launches (8x oversubscribed grid of the plain resident / single-reduction kernels): creating a solver
            # (gato_solver_create -> gato_solver_tune) runs 16 short trial launches (16 iterations each) - listed apart
            trials = []
            packed = re.match(r"pcg_resident_kernel<\w+, \d+, \d+, 0, 0, 0, false, false(, -?\d+)?(, (true|false))?>|pcg_cg1_kernel<", k[0]) and k[1] >= 16 * k[2]
            if packed and len(v) > 16 and max(v) > 3 * min(v):
                cut = max(v) / 3
                short_ = [x for x in v if x < cut]
                if short_ and len(short_) % 16 == 0:
                    trials, v = short_, [x for x in v if x >= cut]
            if trials:
                f.write(f'"{k[0]} [XCD calibration trials, 16 iterations]",{k[1]},{k[2]},{len(trials)},{sum(trials) / len(trials):.0f},{min(trials)},{max(trials)}\n')
            groups = [v]
            if max(v) > 1.6 * min(v) and ", 16, false" in k[0]:      # two bench workloads on one (kernel, grid): split by duration
                cut = (max(v) + min(v)) / 2
                groups = [[x for x in v if x < cut], [x for x in v if x >= cut]]
            for g in groups:
                f.write(f'"{k[0]}",{k[1]},{k[2]},{len(g)},{sum(g) / len(g):.0f},{min(g)},{max(g)}\n')


DIAG = re.compile(r"pcg_resident_kernel<\w+, \d+, \d+, \d+, (true|1|2),|pcg_single_f64m_kernel<\d+, \d+, \d+, [1-9]\d*[,>]")      # the diagnostic (STAMP) build: bench.py's latency-floor launches


def counter(kind):
    p = one(f"prof_{tag}_{kind}/*/*_counter_collection.csv")
    acc = defaultdict(list)
    if p:
        for r in csv.DictReader(open(p)):
            if DIAG.search(r["Kernel_Name"]):
                continue
            if "pcg_" in r["Kernel_Name"] or "stream_step" in r["Kernel_Name"]:
                key = (short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]))
                acc[key].append(float(r["Counter_Value"]))
    return acc


fetch, write = counter("fetch"), counter("write")
# bench workloads -> (kernel template prefix, S, dtype) ; grid identifies K
WL = {"iiwa_14_7_k50_f64": ("pcg_single_f64m_kernel<14", 50), "iiwa_14_7_k50_f32": ("pcg_single_f32x2_kernel<14", 50),
      "iiwa_14_7_k512_f32": ("pcg_resident_kernel<float, 14", 512), "iiwa_14_7_k4096_f32": ("pcg_resident_kernel<float, 14", 4096),
      "iiwa_14_7_k4096_f64": ("pcg_resident_kernel<double, 14", 4096), "s32_c16_k1024_f32": ("pcg_resident_kernel<float, 32", 1024),
      "iiwa_14_7_k131072_f32": ("pcg_dma_kernel<float, 14", 131072), "iiwa_14_7_k131072_f32_semi": ("pcg_resident_kernel<float, 14", 131072),
      "s32_c16_k32768_f32": ("pcg_dma_kernel<float, 32", 32768), "s32_c16_k32768_f32_semi": ("pcg_resident_kernel<float, 32", 32768),
      "iiwa_14_7_k65536_f64": ("pcg_dma_kernel<double, 14", 65536), "iiwa_14_7_k65536_f64_semi": ("pcg_resident_kernel<double, 14", 65536)}
bench = one(f"prof_{tag}_bench.json")
geom = {}
if bench:
    try:
        lines = [json.loads(x) for x in open(bench).read().strip().splitlines() if x.startswith("{")]
        d = lines[-1]                           # the compact headline; the sweep entries are earlier lines
        geom[d["config"]["workload"]] = d["config"]["pcg_workgroups"] * d["config"]["pcg_threads"]
        for r in [x["sweep_entry"] for x in lines if "sweep_entry" in x]:
            if "pcg_groups" in r and r.get("pcg_groups"):
                geom[r["workload"]] = r["pcg_groups"] * r["pcg_threads"]
    except Exception as e:
        print("bench json unreadable:", e)
traffic = {}
# the semi-resident launches of K = 16384 and K = 131072 share kernel and grid (256 x 512): told apart by their traffic
WL["iiwa_14_7_k16384_f32"] = ("pcg_resident_kernel<float, 14", 16384)


SEMI = re.compile(r"pcg_resident_kernel<\w+, \d+, \d+, \d+, \w+, [1-9]")        # XR > 0: the semi-resident instantiations
for name, (prefix, K) in WL.items():
    for key in fetch:
        want = geom.get(name) or 0
        # a register-resident and a semi-resident launch can share prefix AND grid (14/7/4096 f64: 128 x 512, 14/7/65536 f64 semi:
        # 256 x 256): the semi-resident entries take the XR > 0 instantiations only, the others never
        if key[0].startswith("pcg_resident_kernel") and bool(SEMI.match(key[0])) != (name.endswith("_semi") or name == "iiwa_14_7_k16384_f32"):
            continue
        helpers = "pcg_single_" in key[0] and want > 0 and key[1] % want == 0 and (key[1] // want - 1) % 8 == 0 and key[1] // want < 200   # + 8 x h helper blocks
        if key[0].startswith(prefix) and (want in (key[1], key[1] // 8 if key[1] % 8 == 0 else -1) or helpers):   # xcd_pack launches an 8x grid
            fv, wv = fetch[key], write.get(key, [])
            if ", 16, false" in key[0] and ("iiwa_14_7_k16384_f32" in geom and "iiwa_14_7_k131072_f32_semi" in geom):
                big = name == "iiwa_14_7_k131072_f32_semi"
                fv = [v for v in fv if (v >= max(fetch[key]) / 2) == big]
                wv = [v for v in wv if (v >= max(write[key]) / 2) == big] if wv else wv
            if not fv:
                continue
            f_kb = sum(fv) / len(fv)
            w_kb = sum(wv) / len(wv) if wv else 0.0
            traffic[name] = dict(kernel=key[0], grid_threads=key[1], fetch_size_kb=f_kb, write_size_kb=w_kb,
                                 hbm_bytes_per_launch=(2 * f_kb + w_kb) * 1024,
                                 note="HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB: gfx950 FETCH_SIZE reports half the "
                                      "fetched bytes (MI355X_MICROARCH.md, HBM section); calibrated here on the K=50 "
                                      "launches, where 2 x FETCH_SIZE equals the S + Pinv bytes read once. launches=%d"
                                      % len(fv))
# streaming PCG: traffic of one whole gato_pcg call = sum over its launches (init + 2 per iteration).  Two bench
# entries stream: K = 131072 (20 iterations, 41 launches, the largest grid) and K = 512 (100 iterations, 201 launches)
def stream_entry(name, pick, launches_per_call, what, ctype="float"):
    keys = [k for k in fetch if k[0].startswith(f"stream_step_kernel<{ctype}, 14") and pick(k[1])]
    n = sum(len(fetch[k]) for k in keys)
    if not n:
        return
    f_kb = sum(sum(fetch[k]) for k in keys)
    nw = sum(len(write[k]) for k in keys if k in write)
    w_kb = sum(sum(write[k]) for k in keys if k in write)
    traffic[name] = dict(
        kernel=f"stream_step_kernel<{ctype}, 14, *> (all phases)", launches=n,
        fetch_size_kb_per_step_launch=f_kb / n, write_size_kb_per_step_launch=w_kb / max(nw, 1),
        hbm_bytes_per_launch=(2 * f_kb / n + w_kb / max(nw, 1)) * 1024 * launches_per_call,
        note="per gato_pcg call of %s = %d stream_step launches; (2 x FETCH_SIZE + WRITE_SIZE) averaged per launch x %d; "
             "16-B-per-lane LDS-DMA stream: the x2 FETCH_SIZE correction applies" % (what, launches_per_call, launches_per_call))


grids = sorted({k[1] for k in fetch if k[0].startswith("stream_step_kernel<float, 14")})
if grids:
    stream_entry("iiwa_14_7_k131072_f32_streaming", lambda g: g == grids[-1], 41, "20 iterations")
    if len(grids) > 1:
        stream_entry("iiwa_14_7_k512_f32_streaming", lambda g: g == grids[0], 201, "100 iterations")
gd = sorted({k[1] for k in fetch if k[0].startswith("stream_step_kernel<double, 14")})
if gd:                                        # fp64 beyond residency (K = 65536, 20 iterations per call)
    stream_entry("iiwa_14_7_k65536_f64_streaming", lambda g: g == gd[-1], 41, "20 iterations", "double")
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)

# matrix-core counters of the assembly kernels
mf = one(f"prof_{tag}_mfma/*/*_counter_collection.csv")
if mf:
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(mf)):
        n = short(r["Kernel_Name"])
        if any(x in n for x in ("assemble_kernel", "gather_kernel", "schur_kernel", "ss_kernel", "invert_G_kernel", "pcg_resident", "pcg_single", "stream_step")):
            acc[(n, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = ["SQ_INSTS_VALU_MFMA_F32", "SQ_INSTS_VALU_MFMA_F64", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"]
    with open(os.path.join(out, f"{tag}_mfma_counters.csv"), "w") as f:
        f.write("kernel,grid_threads,launches," + ",".join(n + "_avg" for n in names) + "\n")
        for k, v in sorted(acc.items()):
            row = [f"{sum(v[n]) / len(v[n]):.0f}" if v.get(n) else "" for n in names]
            f.write(f'"{k[0]}",{k[1]},{len(next(iter(v.values())))},' + ",".join(row) + "\n")
# issue / LDS activity of the PCG kernels of the default workload
rows = defaultdict(lambda: defaultdict(list))
for kind in ("issue1", "issue2", "issue1_iiwa_14_7_k512_f32", "issue2_iiwa_14_7_k512_f32", "issue1_iiwa_14_7_k4096_f32", "issue2_iiwa_14_7_k4096_f32"):
    p = one(f"prof_{tag}_{kind}/*/*_counter_collection.csv")
    if p:
        for r in csv.DictReader(open(p)):
            if "pcg_" in r["Kernel_Name"]:
                rows[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
if rows:
    names = ["SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE",
             "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"]
    with open(os.path.join(out, f"{tag}_pcg_issue.csv"), "w") as f:
        f.write("kernel,grid_threads,launches," + ",".join(n + "_avg" for n in names) + "\n")
        for k, v in sorted(rows.items()):
            f.write(f'"{k[0]}",{k[1]},{len(next(iter(v.values())))},' + ",".join(f"{sum(v[n]) / len(v[n]):.0f}" if v.get(n) else "" for n in names) + "\n")
print("wrote", sorted(os.listdir(out)))

# cache counters of the persistent K = 131072 launches: one pass per counter and kernel (tools/profile.sh)
cache = {}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_cache_*"))):
    if not os.path.isdir(d):
        continue
    g = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not g:
        continue
    for r in csv.DictReader(open(max(g, key=os.path.getmtime))):
        for kn, label in (("pcg_dma_kernel", "pcg_dma_kernel (LDS-DMA ring, 256 x 512)"), ("pcg_resident_kernel", "pcg_resident_kernel (semi-resident, 256 x 512)")):
            if kn in r["Kernel_Name"]:
                cache.setdefault((label, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
if cache:
    with open(os.path.join(out, f"{tag}_cache_counters.csv"), "w") as f:
        f.write("kernel,workload,counter,launches,avg_per_launch,note\n")
        for (label, k), v in sorted(cache.items()):
            v = v[1:] if len(v) > 1 else v            # the first launch also reads the matrices into the register-resident rows
            f.write(f'"{label}","iiwa 14/7/131072 f32, 10 iterations per launch",{k},{len(v)},{sum(v) / len(v):.0f},'
                    f'"one rocprofv3 --pmc pass per counter"\n')
    print("wrote cache counters:", sorted(cache))
