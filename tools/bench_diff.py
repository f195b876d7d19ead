#!/usr/bin/env python3
"""Side-by-side of the headline fields of two or more bench.py JSON lines: python tools/bench_diff.py a.json b.json ..."""
import json
import sys

KEYS = ["repeat_blocks", "single_alignment", "exact_arith", "ica_mode", "lc_stream_shared_keyframes", "early_exit_on", "c4_dense", "tracked_frame"]
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{f}: value {d['value']:.4g} {d['unit']}, {d['ms_per_step']:.4f} ms/step, roofline {r['achieved']:.0f} {r['unit']} = {r['frac']:.3f}")
    for k in KEYS:
        v = d.get(k)
        if isinstance(v, dict):
            print("   ", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if isinstance(b, (int, float))})
