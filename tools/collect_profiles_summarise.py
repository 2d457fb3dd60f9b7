#!/usr/bin/env python3
"""Condenses what tools/collect_profiles.sh left under gpurun_out/<tag>_* into the files profiles/README.md lists (run by that script on
the GPU box; can be re-run here on the merged gpurun_out/):  python tools/collect_profiles_summarise.py r03"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1]
# on the GPU box only gpurun_out/ travels back: part B condenses its counter directories into gpurun_out/<tag>_condensed/ there
# (--out), and the same script run here afterwards (no --out) fills profiles/ from the merged gpurun_out/, taking the condensed
# counter files as they are
PROF = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "profiles")
os.makedirs(PROF, exist_ok=True)
COND = os.path.join(OUT, f"{tag}_condensed")
if "--out" not in sys.argv and os.path.isdir(COND):
    for name in os.listdir(COND):
        if name != "traffic.json":
            shutil.copyfile(os.path.join(COND, name), os.path.join(PROF, name))
            print("copied (condensed on the GPU box)", name)
PEAK = {"fp32x3": 2.5e15, "fp16": 2.5e15, "fp32": 157.3e12}


def commit():
    """tree the numbers were measured on: EOD_TREE (the GPU box has no .git: the build container passes `git rev-parse --short HEAD`
    on the gpurun command line, see collect_profiles.sh), else git here"""
    if os.environ.get("EOD_TREE"):
        return os.environ["EOD_TREE"]
    try:
        return subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or "?"
    except Exception:
        return "?"


def last_json(path):
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1])


def copy(src, dst):
    if os.path.exists(os.path.join(OUT, src)):
        shutil.copyfile(os.path.join(OUT, src), os.path.join(PROF, dst))
        print("copied", dst)
    else:
        print("MISSING", src)


def counters(dirname, match=""):
    """kernel -> counter -> [sum, n]; plus kernel -> [ns, n] from the rows of one counter"""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    dur_from = {}
    for path in sorted(glob.glob(os.path.join(OUT, dirname, "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"]
            if match and match not in k:
                continue
            c = acc[k][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"])
            c[1] += 1
            if row["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "FETCH_SIZE"):
                # durations from the rows of ONE counter per kernel (a two-pass input file would count every launch twice)
                if dur_from.setdefault(k, row["Counter_Name"]) == row["Counter_Name"]:
                    dur[k][0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    dur[k][1] += 1
    return acc, dur


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:118]


# ---- 1. bench lines, per-op tables, end-to-end calls
for src, dst in [(f"{tag}_bench_default.json", f"{tag}_bench_default.json"),
                 (f"{tag}_bench_fp32x3_per_op_hip_events.json", f"{tag}_bench_fp32x3_per_op_hip_events.json"),
                 (f"{tag}_bench_A1_256_bs8.json", f"{tag}_bench_A1_256_bs8.json"),
                 (f"{tag}_bench_A1_256_bs8_fp32x3_per_op_hip_events.json", f"{tag}_bench_A1_256_bs8_fp32x3_per_op_hip_events.json"),
                 (f"{tag}_bench_A0_64_bs16.json", f"{tag}_bench_A0_64_bs16.json"),
                 (f"{tag}_bench_A0_64_bs16_fp32x3_per_op_hip_events.json", f"{tag}_bench_A0_64_bs16_fp32x3_per_op_hip_events.json"),
                 (f"{tag}_bench_train_fp16.json", f"{tag}_bench_train_fp16.json"),
                 (f"{tag}_bench_train_config5_bs2_fp16.json", f"{tag}_bench_train_config5_bs2_fp16.json"),
                 (f"{tag}_full_1000step_sampling.txt", f"{tag}_full_1000step_sampling.txt"),
                 (f"{tag}_full_ddim250_repaint_config3.txt", f"{tag}_full_ddim250_repaint_config3.txt")]:
    copy(src, dst)

# ---- 2. rocprofv3 kernel statistics
for mode, dst in [("fp32x3", f"{tag}_bench_fp32x3_kernel_stats.csv"), ("fp16", f"{tag}_bench_fp16_kernel_stats.csv"),
                  ("fp32", f"{tag}_bench_fp32_exact_kernel_stats.csv"), ("A1", f"{tag}_bench_A1_256_bs8_fp32x3_kernel_stats.csv"),
                  ("train", f"{tag}_train_step_fp16_kernel_stats.csv")]:
    hits = glob.glob(os.path.join(OUT, f"{tag}_ks_{mode}", "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copyfile(hits[0], os.path.join(PROF, dst))
        print("copied", dst)
    else:
        print("MISSING kernel stats", mode)


# ---- 3. per-kernel algorithmic rate / fraction of the MFMA peak from (kernel stats CSV, per-op table of the same command)
def per_kernel_frac(stats_csv, ops_json, steps_total, out_name, prec="fp32x3"):
    if not (os.path.exists(stats_csv) and os.path.exists(ops_json)):
        print("MISSING inputs for", out_name)
        return
    ops = json.load(open(ops_json))
    flops = defaultdict(float)
    launches = defaultdict(int)
    for o in ops:
        k = o.get("kernel") or {"attention": "attn_fwd_nat", "gemm": "igemm_kernel<gemm>"}.get(o["kind"], o["kind"])
        flops[k] += o.get("flops", 0.0)
        launches[k] += 1
    rows = list(csv.DictReader(open(stats_csv)))
    t = defaultdict(float)
    n = defaultdict(int)
    for r in rows:
        name = r["Name"]
        for key in ("conv3x3_halo_kernel", "conv_up4_halo_kernel", "conv_head_kernel", "igemm_kernel", "attn_fwd_nat", "gn_finalize", "gn_apply",
                    "act_bound", "softmax"):
            if key in name:
                if key == "conv3x3_halo_kernel" and re.search(r"Li32E|<[^,]+, 32,", name):
                    key = "conv3x3_halo_kernel<BN=32>"
                t[key] += float(r["TotalDurationNs"])
                n[key] += int(r["Calls"])
                break
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    res = {"source": f"{os.path.basename(stats_csv)} + {os.path.basename(ops_json)} (same command), tree {commit()}", "peak_tflops": PEAK[prec] / 1e12,
           "note": "algorithmic FLOPs of one forward (per-op table) x steps / total kernel time of the rocprofv3 trace; igemm_kernel also runs "
                   "the GEMM ops of the table", "kernels": {}}
    fl_gen = flops.get("igemm_kernel", 0.0) + flops.get("igemm_kernel<gemm>", 0.0)
    for key in t:
        fl = fl_gen if key == "igemm_kernel" else flops.get(key, 0.0)
        sec = t[key] * 1e-9
        steps = steps_total if fl else 0  # (an op may be two launches: 384-column convs = 256 + 128 columns)
        res["kernels"][key] = {"calls": n[key], "total_ms": t[key] / 1e6, "avg_us": t[key] / max(1, n[key]) / 1e3, "share_of_gpu_time": t[key] / total,
                               "algorithmic_tflops": (fl * steps / sec / 1e12) if fl and sec else None,
                               "frac_of_mfma_peak": (fl * steps / sec / PEAK[prec]) if fl and sec else None}
    json.dump(res, open(os.path.join(PROF, out_name), "w"), indent=1)
    print("wrote", out_name)


per_kernel_frac(os.path.join(PROF, f"{tag}_bench_A1_256_bs8_fp32x3_kernel_stats.csv"),
                os.path.join(PROF, f"{tag}_bench_A1_256_bs8_fp32x3_per_op_hip_events.json"), 13, f"{tag}_A1_256_bs8_per_kernel_frac.json")
per_kernel_frac(os.path.join(PROF, f"{tag}_bench_fp32x3_kernel_stats.csv"),
                os.path.join(PROF, f"{tag}_bench_fp32x3_per_op_hip_events.json"), 13, f"{tag}_A0_256_bs16_per_kernel_frac.json")

# ---- 4. HBM traffic of the dominant kernel (FETCH_SIZE doubled: gfx950), both storage types
traffic_path = os.path.join(PROF, "traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
if "--out" not in sys.argv and os.path.exists(os.path.join(COND, "traffic.json")):  # entries measured on the GPU box
    traffic.update(json.load(open(os.path.join(COND, "traffic.json"))))
skip32 = re.compile(r"halo_kernelI[^L]*Li32E|halo_kernel<[^,]+, 32,")
for prec in ("fp32x3", "fp16"):
    acc, _ = counters(f"{tag}_pmc_traffic_{prec}", "conv3x3_halo_kernel")
    f = [0.0, 0]
    w = [0.0, 0]
    for k, cs in acc.items():
        if skip32.search(k):
            continue
        for name, dst_ in (("FETCH_SIZE", f), ("WRITE_SIZE", w)):
            if name in cs:
                dst_[0] += cs[name][0]
                dst_[1] += cs[name][1]
    if not f[1] or not w[1]:
        print("MISSING traffic counters", prec)
        continue
    fm, wm = f[0] / f[1], w[0] / w[1]
    entry = {"bytes_per_launch": (2.0 * fm + wm) * 1024.0, "fetch_kib_mean": fm, "write_kib_mean": wm, "launches": f[1], "kernel": "conv3x3_halo_kernel",
             "source": f"rocprofv3 -i tools/pmc_traffic.txt -- python3 bench.py --precision {prec} --steps 4 --warmup 2 (tools/collect_profiles.sh {tag}), tree {commit()}",
             "note": "mean over the dominant kernel's launches: (2*FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled (gfx950 correction, "
                     "MI355X_MICROARCH.md HBM section); separate passes per counter"}
    traffic[f"A0_256_16_{prec}"] = entry
    print("traffic", prec, json.dumps(entry)[:200])
ops_path = os.path.join(PROF, f"{tag}_bench_fp32x3_per_op_hip_events.json")
if os.path.exists(ops_path):  # algorithmic bytes (input + weights + output + residual) of the same launches; fp16 storage: half
    ops = [o for o in json.load(open(ops_path)) if o.get("kernel") == "conv3x3_halo_kernel"]
    for prec in ("fp32x3", "fp16"):
        e = traffic.get(f"A0_256_16_{prec}")
        if e and ops and str(e.get("source", "")).find(tag) >= 0:
            e["algorithmic_bytes_per_launch"] = sum(o["bytes"] for o in ops) / len(ops) * (0.5 if prec == "fp16" else 1.0)
            e["ratio_to_algorithmic"] = e["bytes_per_launch"] / e["algorithmic_bytes_per_launch"]
traffic.pop("_pmc_mfma", None)  # (round-1 block: superseded by <tag>_mfma_busy_clock_fp32x3.txt)
json.dump(traffic, open(traffic_path, "w"), indent=1)

# ---- 5. MFMA busy + effective clock per kernel
acc, dur = counters(f"{tag}_pmc_mfma")
lines = []
for k, cs in sorted(acc.items(), key=lambda kv: -dur[kv[0]][0]):
    if "GRBM_GUI_ACTIVE" not in cs or not dur[k][1]:
        continue
    cyc = cs["GRBM_GUI_ACTIVE"][0] / 8.0
    busy = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0, 0])[0] / (4 * 256 * cyc) if cyc else 0.0
    ns = dur[k][0]
    if ns / max(1, dur[k][1]) < 20e3:
        continue
    lines.append(f"{short(k):118s} n={dur[k][1]:4d} avg={ns / dur[k][1] / 1e3:9.1f} us  clock={cyc / ns:.3f} GHz  mfma_busy={busy:.3f}")
if lines:
    with open(os.path.join(PROF, f"{tag}_mfma_busy_clock_fp32x3.txt"), "w") as fh:
        fh.write(f"rocprofv3 -i tools/pmc_mfma.txt -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-op-timing (tree {commit()})\n"
                 "busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x cycles), cycles = GRBM_GUI_ACTIVE / 8 XCDs, clock = cycles / duration\n")
        fh.write("\n".join(lines) + "\n")
    print("wrote mfma busy")


# ---- 6. wave-state split (where the waves' cycles go), per kernel
def wave_state(dirname, out_name, header, min_us=20.0):
    acc, dur = counters(dirname)
    out = [header]
    for k, cs in sorted(acc.items(), key=lambda kv: -dur[kv[0]][0]):
        if "SQ_WAVE_CYCLES" not in cs or not dur[k][1] or dur[k][0] / dur[k][1] < min_us * 1e3:
            continue
        wc = cs["SQ_WAVE_CYCLES"][0] / cs["SQ_WAVE_CYCLES"][1]
        parts = []
        for name, label in (("SQ_WAIT_ANY", "parked at s_waitcnt / s_barrier"), ("SQ_WAIT_INST_ANY", "issue-stalled"), ("SQ_ACTIVE_INST_ANY", "issuing"),
                            ("SQ_ACTIVE_INST_VALU", "  of which VALU + MFMA issue"), ("SQ_ACTIVE_INST_LDS", "  LDS"), ("SQ_ACTIVE_INST_VMEM", "  VMEM (incl. LDS-DMA)"),
                            ("SQ_ACTIVE_INST_SCA", "  scalar"), ("SQ_WAIT_INST_LDS", "  (issue-stalled on LDS)")):
            if name in cs:
                parts.append(f"      {label:34s} {cs[name][0] / cs[name][1] / wc:6.3f}")
        extra = ""
        if "SQ_LDS_BANK_CONFLICT" in cs and "SQ_LDS_IDX_ACTIVE" in cs and cs["SQ_LDS_IDX_ACTIVE"][0]:
            extra = f"      LDS bank-conflict cycles / LDS active cycles {cs['SQ_LDS_BANK_CONFLICT'][0] / cs['SQ_LDS_IDX_ACTIVE'][0]:.3f}"
        out.append(f"{short(k)}  (n={dur[k][1]}, avg {dur[k][0] / dur[k][1] / 1e3:.1f} us): fractions of SQ_WAVE_CYCLES")
        out += parts + ([extra] if extra else [])
    if len(out) > 1:
        open(os.path.join(PROF, out_name), "w").write("\n".join(out) + "\n")
        print("wrote", out_name)
    else:
        print("MISSING wave-state counters", dirname)


wave_state(f"{tag}_pmc_stall", f"{tag}_pmc_wave_state_fp32x3.txt",
           f"rocprofv3 -i tools/pmc_stall.txt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-op-timing (tree {commit()}); "
           "SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES (MI355X_MICROARCH.md)")

# ---- 7. attention kernels: MFMA busy, VALU share
acc, dur = counters(f"{tag}_pmc_attn", "attn")
out = [f"rocprofv3 -i tools/pmc_attn.txt -- python3 bench.py --arch A1 --batch 8 --steps 3 --warmup 1 ... (tree {commit()})"]
for k, cs in sorted(acc.items(), key=lambda kv: -dur[kv[0]][0]):
    if "GRBM_GUI_ACTIVE" not in cs or not dur[k][1]:
        continue
    cyc = cs["GRBM_GUI_ACTIVE"][0] / 8.0
    busy = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0, 0])[0] / (4 * 256 * cyc)
    line = f"{short(k)}  n={dur[k][1]} avg={dur[k][0] / dur[k][1] / 1e3:.1f} us clock={cyc / dur[k][0]:.3f} GHz mfma_busy={busy:.3f}"
    if "SQ_ACTIVE_INST_VALU" in cs and "SQ_BUSY_CYCLES" in cs and cs["SQ_BUSY_CYCLES"][0]:
        line += f" valu_issue/wave_cycles={cs['SQ_ACTIVE_INST_VALU'][0] / max(1.0, cs.get('SQ_WAVE_CYCLES', [1.0, 1])[0]):.3f}"
    if "SQ_INSTS_VALU" in cs:
        line += f" valu_insts_per_launch={cs['SQ_INSTS_VALU'][0] / cs['SQ_INSTS_VALU'][1]:.3e}"
    out.append(line)
if len(out) > 1:
    open(os.path.join(PROF, f"{tag}_attention_pmc.txt"), "w").write("\n".join(out) + "\n")
    print("wrote attention pmc")
