"""Developer check for conv_lane.hip: in every interval of every instantiation, the prefetch loads must stay in flight across
the compute section -- report any `s_waitcnt vmcnt(N)` the compiler placed between a global_load and the FMA block that
follows it (register copies of in-flight loads force such waits and serialise load latency with compute).
usage: python scripts/check_lane_waits.py [conv_lane.s]   (hipcc -S --cuda-device-only output)"""
import re
import sys

src = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/conv_lane.s").read()
bad_total = 0
for m in re.finditer(r"^(_ZN4lane22dwconv_lane_fwd_kernelI\w+):.*?s_endpgm", src, re.S | re.M):
    name, body = m.group(1), m.group(0).splitlines()
    inst = re.sub(r"_ZN4lane22dwconv_lane_fwd_kernelI|EEv.*", "", name)
    state, fma_run, bad, pend = 0, 0, 0, None
    for ln in body:
        t = ln.strip()
        if t.startswith("global_load_dwordx4"):
            state, pend = 1, None
        elif state == 1 and t.startswith("s_waitcnt") and "vmcnt" in t:
            pend = t
        elif state == 1 and t.startswith("v_pk_fma_f32"):
            fma_run += 1
            if fma_run >= 16:                 # the compute section has started
                if pend:
                    bad += 1
                state, fma_run, pend = 0, 0, None
        elif t.startswith("s_barrier"):
            state, fma_run, pend = 0, 0, None
    print(f"{inst:40s} forced waits before compute: {bad}")
    bad_total += bad
print("TOTAL", bad_total)
