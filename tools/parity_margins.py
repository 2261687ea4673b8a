#!/usr/bin/env python
"""Render gpurun_out/parity_margins.json (written by a pytest session, tests/conftest.py) as a markdown table: per test,
how far its gradient comparisons were from a purely relative tolerance.  Usage: python tools/parity_margins.py > note.md"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_margins.json")
m = json.load(open(src))
print("| test | tensors | entries | beyond pure rel. tol. | share | worst share in one tensor | worst rel. error, significant entries | worst norm-wise error |")
print("|---|---|---|---|---|---|---|---|")
tot_e = tot_b = 0
for k in sorted(m):
    v = m[k]
    tot_e += v["entries"]
    tot_b += v["beyond_rel"]
    print(f"| `{k.split('::', 1)[-1]}` | {v['tensors']} | {v['entries']} | {v['beyond_rel']} | {100 * v['beyond_rel'] / max(1, v['entries']):.3f} % | "
          f"{100 * v['worst_floor_fraction']:.2f} % | {v['worst_rel_significant']:.2e} | {v['worst_normwise']:.2e} |")
print(f"\nall tests: {tot_b} of {tot_e} compared entries ({100 * tot_b / max(1, tot_e):.4f} %) were outside the purely relative tolerance "
      "and passed on the absolute floor term.")
