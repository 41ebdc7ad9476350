#!/usr/bin/env python3
"""Whole-step HBM traffic from a per-kernel PMC table (profiles/rNN_cfg3_pmc_by_kernel.csv, written by scripts/pmc_table.py):
sum over kernels of launches x (2 x FETCH_SIZE + WRITE_SIZE) per launch, divided by the profiled steps (profile_round*.sh runs one
survey step + one timed step), and the same per kernel family with its share of GPU time.
usage: python scripts/pmc_sum.py [profiles/r05_cfg3_pmc_by_kernel.csv] [steps=2]"""
import csv
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05_cfg3_pmc_by_kernel.csv"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = [r for r in csv.reader(open(path)) if r and not r[0].startswith("#")][1:]
tot = sum(int(r[2]) * (float(r[4]) + float(r[5])) for r in rows) / 1e3
print(f"{path}: sum of PMC bytes = {tot:.1f} GB over {steps} profiled steps = {tot / steps:.1f} GB per step")
fam = {}
for r in rows:
    k = r[0].split("<")[0]
    f = fam.setdefault(k, [0, 0.0, 0.0])
    f[0] += int(r[2]); f[1] += float(r[1]); f[2] += int(r[2]) * (float(r[4]) + float(r[5])) / 1e3
print(f"{'kernel family':44s} {'launches/step':>13s} {'GPU time %':>10s} {'GB/step':>9s}")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    if v[1] >= 0.3:
        print(f"{k:44s} {v[0] / steps:13.0f} {v[1]:10.2f} {v[2] / steps:9.1f}")
bn = sum(v[1] for k, v in fam.items() if k.startswith("bnact"))
print(f"BatchNorm / squeeze-excite pass family (bnact_*): {bn:.1f} % of GPU time")
