#!/usr/bin/env python3
"""Per-phase time of the dense K=256 loss kernel (debug build of the library, see tools/timeline.py)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import d3pm_amd
from d3pm_amd._lib import lib, check, ptr, stream_ptr, LOSS

B, L, K, T = int(os.environ.get("TL_B", 128)), 1024, 256, 1000          # TL_B=16: one workgroup per CU
dev = torch.device("cuda", 0)
betas = d3pm_amd.generate_betas(1e-4, 0.02, T, "linear")
d = d3pm_amd.DiscreteDiffusion(betas=betas, transition_mat_type="gaussian", num_bits=8, transition_bands=None,
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
for _ in range(3):
    check(lib.d3pm_hybrid_loss(tb.desc(), ptr(logits), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, L, LOSS["hybrid"],
                               0.001, 1.0 / B, None, None, None, ptr(loss), ptr(dl), None, None, 0, ptr(ws), wsb, stream_ptr()))
torch.cuda.synchronize()
raw = C.CDLL(os.environ["D3PM_LIB"])
buf = np.zeros((256, 16, 9), dtype=np.uint64)
assert raw.d3pm_debug_timeline(buf.ctypes.data_as(C.c_void_p)) == 0
tl = buf[:, :4].astype(np.int64)
names = ["", "1 softmax -> LDS", "2 contraction fwd", "3 epilogue (KL, d/df2)", "4 contraction bwd", "5 softmax bwd, store"]
tot = (tl[:, :, 6] - tl[:, :, 0]) * 0.01
print(f"first 256 workgroups; lifetime of a workgroup: median {np.median(tot):.1f} us")
wg = tot[:, 0]
print("lifetime percentiles (us): " + " ".join(f"p{q}={np.percentile(wg, q):.0f}" for q in (0, 10, 50, 90, 99, 100)))
g0 = tl[:, :, 0].min()
print(f"starts within {((tl[:, 0, 0] - g0) * 0.01).max():.1f} us; last end at {((tl[:, 0, 6] - g0) * 0.01).max():.1f} us")
for i in range(1, 6):
    v = tl[:, :, i] * 0.01
    print(f"phase {names[i]:26s} median {np.median(v):7.2f} us  ({100 * np.median(v) / np.median(tot):4.1f}%)   per wave: "
          + " ".join(f"{x:6.2f}" for x in np.median(v, axis=0)))

hw = tl[:, 0, 7]
xcc = tl[:, 0, 8] & 0xf
wave_slot, simd, cu, sh, se = hw & 0xf, (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
t0 = tl[:, 0, 0].min()
print("workgroup  xcc se sh cu  slot(wave0)  start_us  end_us")
order = np.lexsort((tl[:, 0, 0], cu, sh, se, xcc))
for i in order[:48]:
    print(f"{i:9d}  {xcc[i]:3d} {se[i]:2d} {sh[i]:2d} {cu[i]:2d}  {wave_slot[i]:4d}         {(tl[i,0,0]-t0)*0.01:8.2f} {(tl[i,0,6]-t0)*0.01:8.2f}")
