#!/usr/bin/env python3
"""Diagnostic: per-wave cycle accounting of the persistent encoder (a build of libmtc with -DMTC_ENCODE_STAMPS,
made by tools/encode_stamps.sh as matryoshka_transcoder_amd/libmtc_diag.so; the product library carries no stamps).
Prints medians over workgroups for wave 0 (group 0) and wave 4 (group 1), and the in-kernel clock."""
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MTC_LIB"] = os.path.join(ROOT, "matryoshka_transcoder_amd", "libmtc_diag.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from matryoshka_transcoder_amd import _lib  # noqa: E402
from matryoshka_transcoder_amd.engine import HotPath, _p, _stream  # noqa: E402

A, groups, k, B = bench.WORKLOADS["gemma2-2b"]
A = int(sys.argv[1]) if len(sys.argv) > 1 else A
D = sum(groups)
dt = torch.bfloat16
g = torch.Generator("cpu").manual_seed(0)
x = torch.randn(B, A, generator=g).to(dt).cuda()
W = (torch.randn(D, A, generator=g) * (6.0 / D) ** 0.5).to(dt).cuda()
b = (torch.randn(D, generator=g) * 0.01).to(dt).cuda()
hp = HotPath(A, A, bench.cumulative(groups), dt, False)
d = hp.desc(B)
lib = _lib.load()
acts = torch.empty(B, D, dtype=dt, device="cuda")
for _ in range(200):          # a second or so of back-to-back launches first (clock settles under load)
    lib.mtc_encode_fwd(C.byref(d), _p(x), _p(W), _p(b), _p(acts), _stream())
torch.cuda.synchronize()
n = 3 * 256 * 8 * 8
buf = (C.c_ulonglong * n)()
fn = lib.mtc_diag_read_encode_stamps
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_size_t]
assert fn(buf, n) == 0
names = ["total", "prologue", "epilogue_issue", "first_wait_after_epilogue", "other_phase3_wait+barrier", "realtime_100MHz",
         "first_barrier_after_epilogue", "tiles"]
out = {}
for w in (0, 4):
    rows = [[buf[(wg * 8 + w) * 8 + i] for i in range(8)] for wg in range(256)]
    med = [statistics.median(r[i] for r in rows) for i in range(8)]
    out[f"wave{w}"] = dict(zip(names, med))
    out[f"wave{w}"]["clock_GHz"] = med[0] / med[5] * 0.1
    iv = [[buf[256 * 64 + (wg * 8 + w) * 8 + i] for i in range(8)] for wg in range(256)]
    steps = med[7] * (A // 64)
    # interval i = from barrier i to barrier i+1 of a k-step, cycles per step (barrier 7 -> next step's barrier 0 is interval 0)
    out[f"wave{w}"]["cycles_per_interval"] = [round(statistics.median(r[i] for r in iv) / steps, 1) for i in range(8)]
    ex = [[buf[2 * 256 * 64 + (wg * 8 + w) * 8 + i] for i in range(4)] for wg in range(256)]
    # inside the wave's own slots, cycles per k-step: phase 0's load slot up to its barrier, phase 3's staging issue and
    # counted wait, and the MFMA cluster of phase 0 (from behind barrier 0 to in front of barrier 1, fragment wait included)
    out[f"wave{w}"]["within_slots"] = dict(zip(["phase0_issue", "phase3_stage_issue", "phase3_vmcnt_wait", "phase0_mfma_slot"],
                                               [round(statistics.median(r[i] for r in ex) / steps, 1) for i in range(4)]))
print(json.dumps(out, indent=1))
