#!/usr/bin/env python3
"""Kernel resource use as the compiler reports it: registers, spills, LDS and the waves per SIMD they allow.

  python tools/kernel_resources.py > profiles/r03_kernel_resources.txt

Compiles the device code of both .hip files for gfx950 with the flags of the Makefile plus
-Rpass-analysis=kernel-resource-usage (device side only, nothing is linked) and tabulates the remarks.
`waves/SIMD (registers)` is the compiler's occupancy figure; `workgroups/CU (LDS)` is what the 160 KiB of LDS
of a compute unit hold of this kernel's workgroups (256 threads = 4 waves = one wave per SIMD each), so the
waves per SIMD actually resident are the smaller of the two.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt "
         "-fno-slp-vectorize -Wno-pass-failed -Wno-unused-function --cuda-device-only "
         "-Rpass-analysis=kernel-resource-usage -c -o /dev/null").split()
LDS_PER_CU = 160 * 1024


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines() if out.returncode == 0 else names


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^(]*>)?)\(", name)
    return m.group(1) if m else name[:60]


def main():
    rows = []
    for src in ("logipathtracer_amd/csrc/lpt_hip.hip", "logipathtracer_amd/csrc/lpt_lbvh.hip"):
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + sys.argv[1:] + [os.path.join(ROOT, src)], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            return 1
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", line)
            if not m:
                continue
            text = m.group(1).strip()
            if text.startswith("Function Name:"):
                cur = {"name": text.split(":", 1)[1].strip(), "file": os.path.basename(src)}
                rows.append(cur)
            elif cur is not None and ":" in text:
                k, v = text.rsplit(":", 1)
                cur[k.strip()] = v.strip()
    names = demangle([r["name"] for r in rows])
    print("Kernel resource use, gfx950, flags of the Makefile (hipcc -Rpass-analysis=kernel-resource-usage); "
          "extra flags: " + (" ".join(sys.argv[1:]) or "none"))
    print("%-44s %5s %5s %6s %7s %9s %11s %13s %9s" % ("kernel", "VGPR", "SGPR", "spills", "scratch", "LDS bytes",
                                                     "waves/SIMD", "workgroups/CU", "resident"))
    print("%-44s %5s %5s %6s %7s %9s %11s %13s %9s" % ("", "", "", "(VGPR)", "B/lane", "/block", "(registers)", "(LDS)", "waves/SIMD"))
    for r, n in zip(rows, names):
        if r["name"].startswith("__") or "rocprim" in n:
            continue
        lds = int(r.get("LDS Size [bytes/block]", "0"))
        occ = int(r.get("Occupancy [waves/SIMD]", "0"))
        by_lds = LDS_PER_CU // lds if lds else 99
        # workgroups of 256 threads put one wave on each SIMD; small helper kernels use other sizes: registers only
        resident = min(occ, by_lds) if lds else occ
        print("%-44s %5s %5s %6s %7s %9d %11d %13s %9d" % (short(n), r.get("VGPRs", "?"), r.get("TotalSGPRs", "?"),
                                                         r.get("VGPRs Spill", "?"), r.get("ScratchSize [bytes/lane]", "?"), lds, occ,
                                                         by_lds if lds else "-", resident))
    return 0


if __name__ == "__main__":
    sys.exit(main())
