#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).
Usage: python tools/kernel_regs.py eo_diffusion_amd/csrc/igemm.hip [substring ...]   (CPU only: cross-compiles, no GPU needed)"""
import re
import subprocess
import sys
import tempfile


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def table(src, extra=()):
    with tempfile.NamedTemporaryFile(suffix=".o") as o:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                            *extra, "-c", src, "-o", o.name], capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr)
        raise SystemExit(r.returncode)
    rows, cur = [], None
    for line in r.stderr.split("\n"):
        m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill|SGPRs Spill): (.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).split(" [")[0].strip()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


if __name__ == "__main__":
    rows = table(sys.argv[1])
    names = demangle([r["name"] for r in rows])
    pats = sys.argv[2:]
    print(f"{'VGPR':>5} {'AGPR':>5} {'vspill':>6} {'scratch':>7} {'occ':>4}  kernel")
    for r in rows:
        n = names[r["name"]]
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*$", "", n)
        if pats and not any(p in n for p in pats):
            continue
        print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('VGPRs Spill', '?'):>6} {r.get('ScratchSize [bytes/lane]', '?'):>7} "
              f"{r.get('Occupancy [waves/SIMD]', '?'):>4}  {n}")
