#!/usr/bin/env python3
"""Instruction mix of the lane = column depthwise kernels from the ISA (developer tool): per instantiation the static counts of
one pipeline interval (the code between two s_barrier that holds the FMA block) -- VALU / packed FMAs / LDS / SALU -- and the
number of SGPR-spill reloads (v_readlane_b32) in the whole kernel body.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -S --cuda-device-only mammo_clip_amd/csrc/conv_lane.hip -o /tmp/conv_lane.s
       python scripts/lane_isa_mix.py [/tmp/conv_lane.s]"""
import collections
import re
import sys

src = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/conv_lane.s").read()
print(f"{'K,S,NCOL,MODE,G':18s} {'interval':>8s} {'VALU':>6s} {'pk_fma':>7s} {'LDS':>5s} {'SALU':>6s} {'readlane (kernel)':>18s}")
for m in re.finditer(r"^_ZN4lane22dwconv_lane_fwd_kernelI(\w+?)EEv\w+:.*?s_endpgm", src, re.S | re.M):
    name = m.group(1).replace("Li", "").replace("E", ",").rstrip(",")
    segs, cur = [], []
    for ln in m.group(0).splitlines():
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        cur.append(t.split()[0])
        if cur[-1] == "s_barrier":
            segs.append(cur)
            cur = []
    big = [s for s in segs if sum(1 for o in s if o in ("v_pk_fma_f32", "v_pk_mul_f32")) >= 16]
    if not big:
        continue
    c = collections.Counter(big[0])
    print(f"{name:18s} {len(big[0]):8d} {sum(v for k, v in c.items() if k.startswith('v_')):6d} {c['v_pk_fma_f32']:7d} "
          f"{sum(v for k, v in c.items() if k.startswith('ds_')):5d} {sum(v for k, v in c.items() if k.startswith('s_')):6d} {m.group(0).count('v_readlane_b32'):18d}")
