#!/usr/bin/env python3
"""HBM traffic per launch of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), collected and corrected
as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:

    rocprofv3 -i tools/pmc_traffic.txt -d gpurun_out/pmc -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-op-timing
    python tools/pmc_traffic.py gpurun_out/pmc <key> [profiles/traffic.json] [ops.json of bench.py --dump-ops]

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled.  Writes
`key` = {bytes_per_launch, fetch_kib_mean, write_kib_mean, launches, kernel, source} into the JSON (the number bench.py attaches to
its line as `roofline.traffic_from_profile`)."""
import csv
import glob
import json
import os
import subprocess
import sys


def main():
    root, key = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    match = os.environ.get("EOD_PMC_KERNEL", "conv3x3_halo_kernel")
    # the 32-column head-conv instance is not the dominant kernel (mangled: ...Li32E..., demangled: conv3x3_halo_kernel<float, 32, ...)
    import re
    skip = re.compile(os.environ.get("EOD_PMC_SKIP", r"halo_kernelI[^L]*Li32E|halo_kernel<[^,]+, 32,"))
    sums = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                if match not in name or skip.search(name):
                    continue
                c = sums.setdefault(row["Counter_Name"], [0.0, 0])
                c[0] += float(row["Counter_Value"])
                c[1] += 1
    if "FETCH_SIZE" not in sums or "WRITE_SIZE" not in sums:
        raise SystemExit(f"no FETCH_SIZE / WRITE_SIZE rows for '{match}' under {root}: {list(sums)}")
    f_mean = sums["FETCH_SIZE"][0] / sums["FETCH_SIZE"][1]
    w_mean = sums["WRITE_SIZE"][0] / sums["WRITE_SIZE"][1]
    import os
    commit = os.environ.get("EOD_TREE", "")
    if not commit:
        try:
            commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        except Exception:
            commit = "?"
    data = json.load(open(out)) if os.path.exists(out) else {}
    alg = None
    if len(sys.argv) > 4:  # per-op table of `bench.py --dump-ops`: algorithmic bytes (input + weights + output + residual) of the same launches
        ops = [o for o in json.load(open(sys.argv[4])) if o.get("kernel") == match]
        alg = sum(o["bytes"] for o in ops) / max(1, len(ops))
    data[key] = {"bytes_per_launch": (2.0 * f_mean + w_mean) * 1024.0, "fetch_kib_mean": f_mean, "write_kib_mean": w_mean,
                 "launches": sums["FETCH_SIZE"][1], "kernel": match, "source": f"rocprofv3 -i tools/pmc_traffic.txt ({root}), tree {commit}",
                 "note": "mean over the dominant kernel's launches: (2*FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled (gfx950 correction)"}
    if alg is not None:
        data[key]["algorithmic_bytes_per_launch"] = alg
    with open(out, "w") as f:
        json.dump(data, f, indent=1)
    print(json.dumps(data[key]))


if __name__ == "__main__":
    main()
