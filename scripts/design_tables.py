#!/usr/bin/env python3
"""Markdown tables of DESIGN.md section 7 from the committed bench records (profiles/rNN_bench_*.json).
usage: python scripts/design_tables.py [r05]"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def rec(name):
    p = os.path.join(P, f"{tag}_bench_{name}.json")
    return json.load(open(p)) if os.path.exists(p) else None


rows = [("cfg4", "config #4 (headline): B5, global batch 1024, N = 1, 32 micro-batches"), ("cfg4_f16", "the same, f16 storage build (parity configuration)"),
        ("as2", "one rank's share of the N = 2 point (512 pairs)"), ("as4", "one rank's share of the N = 4 point (256 pairs)"),
        ("cfg3", "config #3: B5, 32 pairs / GPU in one pass"), ("cfg3_f16", "the same, f16 storage build"),
        ("cfg2", "config #2: B2, 64 pairs, 912 x 912"), ("cfg1", "config #1: B2, 4 pairs, 224 x 224 (launch-bound parity case)"),
        ("cfg5", "config #5: cfg4's model, fp8 operands in the late 1x1 convs, global batch 2048")]
print("| workload | pairs/s | ms / step | peak HBM allocated / reserved (GB) |")
print("|---|---|---|---|")
for k, what in rows:
    d = rec(k)
    if d:
        print(f"| {what} | **{d['value']:.1f}** | {d['ms_per_step']:.1f} | {d['config'].get('peak_hbm_gb')} / {d['config'].get('peak_reserved_gb')} |")
d = rec("cfg4")
if d and d.get("n8_load"):
    n8 = d["n8_load"]
    print(f"| N = 8 per-GPU load on one GPU (`n8_load`: 128 pairs, 4 kept mode-3 graphs, no collectives) | {n8['pairs_per_s_per_gpu']:.1f} per GPU | {n8['ms_per_step']:.1f} | {n8['peak_hbm_gb']} |")
if d and d.get("cpu_baseline"):
    c = d["cpu_baseline"]
    print(f"| `cpu_baseline` (oracle, {c['cores']} host threads, 1 pair at the workload's size) | {c['value']} | — | — |")
print()
print("| run | class (entry point) | bound | achieved | frac of peak | launches / step | avg launch (µs) | algorithmic MB / launch | PMC MB / launch |")
print("|---|---|---|---|---|---|---|---|---|")
for k in ("cfg4", "cfg3"):
    d = rec(k)
    if not d:
        continue
    for key in ("roofline", "roofline_runner_up", "roofline_third"):
        r = d.get(key)
        if r:
            tr = f"{r['traffic'] / 1e6:.0f}" if r.get("traffic") else "—"
            print(f"| {k} | `{r['class']}` | {r['bound']} | {r['achieved']:.0f} {r['unit']} | **{r['frac']:.3f}** | {r['launches_per_step']} | {r['avg_launch_us']} | "
                  f"{r['algorithmic_bytes_per_launch'] / 1e6:.0f} | {tr} |")
