#!/usr/bin/env python3
"""In-kernel timeline of the fused loss kernel (debug build of the library).

    hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -shared -DD3PM_TIMELINE [-DD3PM_TL_ROWS] \
        csrc/tables.hip csrc/image_structured.hip csrc/protein.hip csrc/logistic.hip -o tools/micro/libd3pm_timeline.so
    D3PM_LIB=tools/micro/libd3pm_timeline.so python tools/timeline.py

Prints, per recorded point, the spread over workgroups of the 100 MHz wall clock
relative to the earliest workgroup start of the launch.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import d3pm_amd
from d3pm_amd._lib import lib, check, ptr, stream_ptr, LOSS

B, L, K, T = 256, 512, 32, 500
dev = torch.device("cuda", 0)
betas = d3pm_amd.generate_betas(0.0, 0.0, T, "jsd")
d = d3pm_amd.DiscreteDiffusion(betas=betas, transition_mat_type="uniform", num_bits=5, transition_bands=None,
                               model_prediction="x_start", model_output="logits", loss_type="hybrid",
                               hybrid_coeff=0.001, device=dev)
tb = d.tables()
x0 = torch.randint(0, K, (B, L, 1, 1), device=dev)
logits = torch.randn(B, L, 1, 1, K, device=dev)
t = torch.randint(1, T, (B,), dtype=torch.int32, device=dev)
xt = d.q_sample(x0, t)
dl = torch.empty_like(logits)
loss = torch.empty(B, device=dev)
wsb = lib.d3pm_hybrid_loss_workspace(B, L, K)
ws = torch.empty(max(wsb // 4, 1), device=dev)
sp = stream_ptr()


def run():
    check(lib.d3pm_hybrid_loss(tb.desc(), ptr(logits), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, L, LOSS["hybrid"],
                               0.001, 1.0 / B, None, None, None, ptr(loss), ptr(dl), None, None, 0, ptr(ws), wsb, sp))


import time
MODE = sys.argv[1] if len(sys.argv) > 1 else "back2back"
if MODE == "nograd":
    dl = None
    def run():
        check(lib.d3pm_hybrid_loss(tb.desc(), ptr(logits), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, L, LOSS["hybrid"],
                                   0.001, 1.0 / B, None, None, None, ptr(loss), None, None, None, 0, ptr(ws), wsb, sp))
if MODE == "next":              # the launch that also corrupts the next batch
    t_n = torch.empty(B, dtype=torch.int32, device=dev)
    xt_n = torch.empty_like(xt)
    def run():
        check(lib.d3pm_hybrid_loss_then_q_sample(tb.desc(), ptr(logits), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, L,
                                                 LOSS["hybrid"], 0.001, 1.0 / B, None, None, None, ptr(loss), ptr(dl),
                                                 ptr(ws), wsb, ptr(x0), 8, 7, 0, None, ptr(t_n), ptr(xt_n), sp))
for _ in range(20):
    run()
torch.cuda.synchronize()
if MODE == "isolated":          # one launch on an idle GPU
    time.sleep(0.05)
    run()
    torch.cuda.synchronize()
print("mode:", MODE)
raw = C.CDLL(os.environ["D3PM_LIB"])
buf = np.zeros((256, 16, 9), dtype=np.uint64)
assert raw.d3pm_debug_timeline(buf.ctypes.data_as(C.c_void_p)) == 0
tl = buf[:B].astype(np.int64)
SHADER = "SHADERCLK" in os.environ.get("D3PM_LIB", "")   # per-wave shader-clock ticks instead of global microseconds
names = ["start", "t loaded", "rows arrived, staged in LDS", "phase A done", "tables staged",
         "phase B arithmetic done", "gradient rows in LDS, patched", "next batch drawn (next mode)",
         "global stores issued"]
nw = int((tl[0, :, 0] != 0).sum())
t0 = tl[:, :nw, 0].min()
print(f"{nw} waves per workgroup; median over workgroups, microseconds from the earliest start, one column per wave")
print(" " * 30 + "".join(f"{w:7d}" for w in range(nw)))
for i, n in enumerate(names):
    v = np.median((tl[:, :nw, i] - t0) * (0.01 if not SHADER else 1 / 2160.0), axis=0)
    print(f"{n:30s}" + "".join(f"{x:7.2f}" for x in v))
