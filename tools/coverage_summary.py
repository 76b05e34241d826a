#!/usr/bin/env python3
"""Kernel templates of the library against the kernels the GPU tests launched.

    python tools/coverage_summary.py gpurun_out/kernel_coverage.txt > profiles/rNN_kernel_coverage.txt

Reads the kernel names of the gfx950 code objects (compiled here, as tools/kernel_resources.py does) and the
`<launches> <kernel name>` lines tools/kernel_coverage.sh wrote on the GPU box, and prints per template how many of
its instantiations the tests launched."""
from __future__ import annotations

import collections
import re
import sys
import tempfile

sys.path.insert(0, __file__.rsplit("/", 1)[0])
import kernel_resources as kr          # noqa: E402


def template_of(name: str) -> str:
    name = name.replace("d3pm::", "")
    name = re.sub(r"^void ", "", name)
    return re.split(r"[<(]", name, 1)[0]


def signature(name: str) -> str:
    """The name with its template arguments, without the parameter list."""
    name = re.sub(r"^void ", "", name.replace("d3pm::", ""))
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def main():
    built = collections.defaultdict(set)
    outdir = tempfile.mkdtemp(prefix="d3pm_cov_")
    for src in kr.SOURCES:
        for n in kr.demangle([k[".name"] for k in kr.kernels_of(kr.device_object(src, outdir))]):
            built[template_of(n)].add(signature(n))
    launched = collections.defaultdict(set)
    calls = collections.Counter()
    for line in open(sys.argv[1]):
        cnt, name = line.strip().split(" ", 1)
        t = template_of(name)
        if t not in built:                  # a kernel of torch or of the runtime
            continue
        launched[t].add(signature(name))
        calls[t] += int(cnt)
    print("Kernel templates of libd3pm_hip.so against the kernels `pytest tests -m gpu` launches (rocprofv3 --kernel-trace of\n"
          "the test run, tools/kernel_coverage.sh; tests/test_gpu_ddp.py left out: its ranks are processes of their own).\n"
          "instantiations launched / instantiations in the library (one per lane layout = per class count K, noise kind,\n"
          "gradient or not), launches in the run, template.\n")
    missing = []
    for t in sorted(built):
        got = len(launched[t] & built[t]) if launched[t] & built[t] else len(launched[t])
        print(f"{got:>3} / {len(built[t]):>3}  {calls[t]:>7}  {t}")
        if not launched[t]:
            missing.append(t)
    print("\n" + ("Every template is launched by the tests." if not missing else "NOT launched: " + ", ".join(missing)))
    few = [t for t in ("k_hybrid_loss_pair32", "k_lvb_loss_tpl", "k_hybrid_loss_tpl") if t in built]
    for t in few:
        rest = sorted(built[t] - launched[t])
        print(f"{t}: " + ("all instantiations launched" if not rest else "not launched: " + "; ".join(rest)))


if __name__ == "__main__":
    main()
