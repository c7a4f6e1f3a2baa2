#!/usr/bin/env python3
"""Per-workgroup / per-wave phase timeline of the fused GEMV.  Needs the diagnostic build
(`make -C paroquant_amd/csrc clean all DIAG=1`) and PARO_GEMV_PD=31: wave 0 of every workgroup records
s_memtime (shader clock) at entry / first loads issued / first coefficients arrived / stages done / fragments +
sums done / first unit's tiles consumed / all units done / every wave done / partials staged / output
written; every wave records its own start, first-coefficient arrival, first unit done, all units done.
The stamps are data-dependent (asm with a VGPR input), so they cannot be scheduled above the waits they
follow.  s_memtime is per XCD: only differences inside a workgroup are meaningful.
    PARO_GEMV_PD=31 python tools/timeline_gemv.py --model llama3-8b --linear o_proj --tpw 1 --waves 16"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import layer_shapes, synth_packed
from paroquant_amd import ops, _native as nat

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama3-8b"); ap.add_argument("--linear", default="o_proj")
ap.add_argument("--tpw", type=int, default=1); ap.add_argument("--waves", type=int, default=16)
ap.add_argument("--clock_mhz", type=float, default=100.0, help="s_memtime tick rate in MHz")
a = ap.parse_args()
dev = torch.device("cuda:0"); gen = torch.Generator(device=dev); gen.manual_seed(0)
name, K, sizes, _ = [s for s in layer_shapes(a.model) if s[0] == a.linear][0]
packs = [synth_packed(K, sizes, dev, gen) for _ in range(12)]
x = torch.randn(1, K, device=dev, dtype=torch.float16, generator=gen)
for i in range(8):
    ops.w4a16_gemv_tuned(x, packs[i], a.tpw, 1, a.waves, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.w4a16_gemv_tuned(x, packs[9], a.tpw, 1, a.waves, 0); e1.record(); torch.cuda.synchronize()
ncb = sum((s // 16 + a.tpw - 1) // a.tpw for s in sizes)
ws = packs[9].workspace
W = 80   # 64-bit words per workgroup (gemv_impl.hpp, DIAG == 3)
raw = ws[nat.PARO_WS_COUNTER_BYTES:nat.PARO_WS_COUNTER_BYTES + ncb * W * 8].view(torch.int64).cpu().numpy().reshape(ncb, W)
ws[nat.PARO_WS_COUNTER_BYTES:nat.PARO_WS_COUNTER_BYTES + ncb * W * 8].zero_()
t = np.concatenate([raw[:, :9], raw[:, 10:11]], axis=1).astype(np.float64)
# s_memtime counters are per XCD and not synchronised across XCDs: report each phase RELATIVE TO THE
# WORKGROUP'S OWN START, in shader cycles (divide by the shader clock, ~2.1-2.4 GHz, for time).
d = t - t[:, :1]
names = ["wg_start", "loads_issued", "coeffs_arrived", "rotation_done", "tiles_consumed", "units_done(w0)",
         "output_written", "all_waves_done", "partials_staged", "stages_done"]
order = [0, 1, 2, 9, 4, 5, 7, 8, 6]
print(f"{a.model} {name} tpw={a.tpw} waves={a.waves} workgroups={ncb}  kernel {e0.elapsed_time(e1)*1e3:.1f} us incl. launch"
      f" (wave 0 of each workgroup; shader cycles since its own start)")
for k in order:
    col = d[:, k]
    print(f"  {names[k]:16s} mean {col.mean():8.0f}  p5 {np.percentile(col,5):8.0f}  p95 {np.percentile(col,95):8.0f}  max {col.max():8.0f}")
# chip-wide real-time counter (100 MHz): when each workgroup entered / left, relative to the first entry
rt = raw[:, 11:13].astype(np.float64) * 10.0   # ns
rt -= rt[:, 0].min()
print(f"  chip-wide clock, ns since the first workgroup's entry:  entry mean {rt[:,0].mean():6.0f} p95 {np.percentile(rt[:,0],95):6.0f} max {rt[:,0].max():6.0f}"
      f"   exit min {rt[:,1].min():6.0f} mean {rt[:,1].mean():6.0f} p95 {np.percentile(rt[:,1],95):6.0f} max {rt[:,1].max():6.0f}")
wv = raw[:, 16:16 + 4 * a.waves].astype(np.float64).reshape(ncb, a.waves, 4) - t[:, :1, None]
print("  by wave index, mean cycles since workgroup start:  start / first coefficients arrived / first unit done / all units done")
for w in range(a.waves):
    print(f"    wave {w:2d}: " + " / ".join(f"{wv[:, w, k].mean():7.0f}" for k in range(4)))
