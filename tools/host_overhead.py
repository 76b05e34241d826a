#!/usr/bin/env python3
"""Host time per eager call of q_sample through each layer of the package (facade method, registered operator, the
C ABI through ctypes), launches queued back to back without waiting for the GPU: what bounds an eager call when the
kernel itself takes 4 us.

    python tools/host_overhead.py
"""
import time, sys, os
sys.path.insert(0, os.getcwd())
import torch
import d3pm_amd
from d3pm_amd._lib import lib, check, ptr, stream_ptr, NOISE_PHILOX
from d3pm_amd import _ops
dev = torch.device("cuda", 0)
B, L, K, T = 256, 512, 32, 500
d = d3pm_amd.DiscreteDiffusion(betas=d3pm_amd.generate_betas(0.0, 0.0, T, "jsd"), transition_mat_type="uniform", num_bits=5,
                               transition_bands=None, model_prediction="x_start", model_output="logits", loss_type="hybrid",
                               hybrid_coeff=0.001, device=dev)
x0 = torch.randint(0, K, (B, L, 1, 1), device=dev)
t = torch.randint(0, T, (B,), dtype=torch.int32, device=dev)
tb = d.tables()
def timeit(name, fn, n=3000):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:40s} host {1e6*(t1-t0)/n:6.2f} us/call   with drain {1e6*(t2-t0)/n:6.2f} us/call", flush=True)
timeit("d.q_sample(x0, t)", lambda: d.q_sample(x0, t))
args = tb.op_args()
from d3pm_amd.d3pm.diffusion_discrete import ops, i64
timeit("ops.q_sample(*args...)", lambda: ops.q_sample(*args, x0, t, None, NOISE_PHILOX, 1, 0, None))
out = torch.empty_like(x0)
desc = tb.desc(); px0 = ptr(x0); pt = ptr(t); po = ptr(out); sp = stream_ptr()
timeit("ctypes prebuilt pointers", lambda: lib.d3pm_q_sample(desc, px0, 8, pt, 4, B, L, None, NOISE_PHILOX, 1, 0, None, po, sp))
timeit("ctypes + ptr() + empty_like + stream", lambda: lib.d3pm_q_sample(tb.desc(), ptr(x0), 8, ptr(t), 4, B, L, None, NOISE_PHILOX, 1, 0, None, ptr(torch.empty_like(x0)), stream_ptr()))
timeit("torch.empty_like", lambda: torch.empty_like(x0))
timeit("d.tables()", lambda: d.tables())
timeit("tb.op_args()", lambda: tb.op_args())
timeit("stream_ptr()", lambda: stream_ptr())
