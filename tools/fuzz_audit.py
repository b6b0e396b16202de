"""Histogram of the fuzz draws that tests/test_gpu_fuzz.py skips as ill-posed: device-vs-checker error of the loss and of the
gradient per skip reason, from the JSON lines written under ODINN_FUZZ_AUDIT=<file> (see _skip there).

  ODINN_FUZZ_AUDIT=gpurun_out/fuzz_skips.jsonl ODINN_FUZZ_SEEDS=0:1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -n 8 -q \
      -k "gradient_matches or velocity_loss or aggregated" --timeout 180
  python tools/fuzz_audit.py gpurun_out/fuzz_skips.jsonl > profiles/r04/fuzz_skips.txt
"""
import collections
import json
import math
import sys


def decade(x):
    if x is None:
        return "n/a"
    if x == 0.0:
        return "0"
    return "1e%+03d" % math.floor(math.log10(x))


def main(path):
    rows = [json.loads(l) for l in open(path) if l.strip()]
    print(f"{len(rows)} skipped draws in {path}")
    by = collections.defaultdict(list)
    for r in rows:
        by[(r["test"], r["reason"])].append(r)
    for (test, reason), rs in sorted(by.items()):
        print(f"\n{test}: {reason}...  ({len(rs)} draws)")
        for key in ("loss_relerr", "grad_relerr"):
            h = collections.Counter(decade(r[key]) for r in rs)
            order = sorted(h, key=lambda d: (d == "n/a", d == "0", d))
            print(f"  {key:12s} " + "  ".join(f"{d}: {h[d]}" for d in order))
        worst = max(rs, key=lambda r: r["grad_relerr"] or 0.0)
        print(f"  worst gradient: seed {worst['seed']} law {worst['law']} mode {worst['mode']} relerr {worst['grad_relerr']}")
        modes = collections.Counter((r["law"], r["mode"]) for r in rs)
        print("  (law, mode): " + ", ".join(f"{k}: {v}" for k, v in sorted(modes.items(), key=lambda kv: -kv[1])[:8]))


if __name__ == "__main__":
    main(sys.argv[1])
