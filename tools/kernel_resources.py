#!/usr/bin/env python3
"""Registers, scratch, LDS and instruction counts of the library's kernels (runs anywhere hipcc is: no GPU needed).

Compiles each source of the library for the device only (gfx950), reads the kernel descriptors' metadata
(`llvm-readelf --notes`) and, with --disasm PATTERN, the instruction mix of the kernels whose demangled name contains
PATTERN (`llvm-objdump -d`): vector / scalar / LDS / global-memory / transcendental instruction counts of the code
object (static counts: a loop body is counted once).

    python tools/kernel_resources.py [--filter lvb_loss] [--disasm k_hybrid_loss_pair32] [--dump-dir DIR]
"""
from __future__ import annotations

import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "exploring-discrete-denoising-diffusion-models-for-protein-design_amd", "csrc")
SOURCES = ["tables.hip", "image_structured.hip", "protein.hip", "logistic.hip"]
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True,
                         check=True).stdout.splitlines()
    return [o.replace("d3pm::", "") for o in out]


def device_object(src: str, outdir: str) -> str:
    obj = os.path.join(outdir, src.replace(".hip", ".gfx950.o"))
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "--offload-device-only", "-std=c++17", "-c",
           os.path.join(CSRC, src), "-o", obj] + os.environ.get("D3PM_HIPCC_FLAGS", "").split()
    subprocess.run(cmd, cwd=CSRC, check=True)
    # the compiler writes an offload bundle: take the gfx950 code object out of it
    elf = obj.replace(".o", ".co")
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--input=" + obj, "--output=" + elf, "--unbundle"], check=True)
    return elf


def kernels_of(obj: str):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", obj], capture_output=True, text=True,
                         check=True).stdout
    i = txt.index("amdhsa.kernels")
    meta = yaml.safe_load(txt[txt.rindex("---", 0, i):txt.index("...", i)])
    return meta["amdhsa.kernels"]


CLASSES = [("trans", re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_")), ("mfma", re.compile(r"^v_mfma")),
           ("dpp", None), ("valu", re.compile(r"^v_")), ("salu", re.compile(r"^s_(?!waitcnt|nop|barrier|endpgm|branch|cbranch|sleep|setprio)")),
           ("lds", re.compile(r"^ds_")), ("vmem", re.compile(r"^(global|buffer|flat|scratch)_")),
           ("smem", re.compile(r"^s_(load|buffer_load)")), ("wait", re.compile(r"^s_waitcnt")),
           ("branch", re.compile(r"^s_(c?branch)"))]


def instruction_mix(obj: str, pattern: str):
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", obj], capture_output=True,
                         text=True, check=True).stdout
    out = {}
    cur = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            out[cur] = collections.Counter()
            continue
        if cur is None or not line.startswith("\t") and not line.startswith(" "):
            continue
        ins = line.strip().split()
        if not ins:
            continue
        op = ins[0]
        c = out[cur]
        c["total"] += 1
        if op.startswith("v_") and ("dpp" in line or "row_" in line or "quad_perm" in line):
            c["dpp"] += 1
        for name, rx in CLASSES:
            if rx is not None and rx.match(op):
                if name == "smem":
                    c["salu"] -= 1
                c[name] += 1
                break
    names = list(out)
    dm = demangle(names)
    return {d: out[n] for n, d in zip(names, dm) if pattern in d}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="", help="only kernels whose demangled name contains this")
    ap.add_argument("--disasm", default=None, help="instruction mix of the kernels whose name contains this")
    ap.add_argument("--dump-dir", default=None, help="keep the device objects here (default: a temporary directory)")
    a = ap.parse_args()
    outdir = a.dump_dir or tempfile.mkdtemp(prefix="d3pm_res_")
    os.makedirs(outdir, exist_ok=True)
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'spill':>6} {'scratch':>8} {'lds':>7}  kernel")
    for src in SOURCES:
        obj = device_object(src, outdir)
        ks = kernels_of(obj)
        names = demangle([k[".name"] for k in ks])
        for k, n in sorted(zip(ks, names), key=lambda kn: kn[1]):
            if a.filter in n:
                print(f"{k['.vgpr_count']:>5} {k.get('.agpr_count', 0):>5} {k['.sgpr_count']:>5} "
                      f"{k['.vgpr_spill_count']:>6} {k['.private_segment_fixed_size']:>8} "
                      f"{k['.group_segment_fixed_size']:>7}  {n[:170]}")
        if a.disasm:
            for n, c in instruction_mix(obj, a.disasm).items():
                print(f"    [{n[:150]}]")
                print("     " + "  ".join(f"{key}={c[key]}" for key in ("total", "valu", "trans", "dpp", "mfma", "salu", "smem",
                                                                        "lds", "vmem", "wait", "branch")))


if __name__ == "__main__":
    sys.exit(main())
