#!/bin/bash
# The default bench on the GPU box, its JSON line kept under gpurun_out/ and summarised on stdout.
mkdir -p gpurun_out
OUT=gpurun_out/${1:-bench}.json
shift
timeout -k 10 400 python -u bench.py "$@" > $OUT 2> ${OUT%.json}.err
rc=$?
tail -c 400 ${OUT%.json}.err
echo "rc=$rc"
[ $rc -eq 0 ] || exit $rc
python - "$OUT" <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("value %.3e %s  ms/step %.4f  frac %.3f  cache-resident %s  kernel_ms %.5f  timed_s %.4f" % (
    d["value"], d["unit"], d["ms_per_step"], r["frac"], r.get("frac_cache_resident"), r["kernel_ms"], d["config"]["timed_seconds"]), " two-launch ms/step", d["config"].get("two_launches_per_step_ms_per_step"), " with-next kernel", r.get("kernel_ms_with_next_q_sample"), r.get("frac_with_next_q_sample"), " replayed", r.get("kernel_ms_graph_replayed"), r.get("frac_graph_replayed"))
for k in ("absorbing", "reverse_sample", "image_k256", "protein_lvb"):
    s = d.get(k)
    if s:
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in s.items() if a in ("value", "ms_per_step", "kernel_ms", "end_to_end_tiny_denoiser", "tokens_per_s")},
              "frac %.3f" % s["roofline"]["frac"], "cpu", (s.get("cpu_baseline") or {}).get("value"))
lb = d.get("large_batch")
if lb:
    print("large_batch hybrid %.2f us frac %.3f   lvb %.2f us frac %.3f" % (
        lb["hybrid_loss"]["kernel_ms"] * 1e3, lb["hybrid_loss"]["frac"], lb["lvb_loss"]["kernel_ms"] * 1e3, lb["lvb_loss"]["frac"]))
print("cpu_baseline", d.get("cpu_baseline"))
P
