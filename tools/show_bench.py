#!/usr/bin/env python3
"""Print the headline fields of one bench.py JSON line: proofs/s, single-proof latency and the
per-kernel milliseconds of the profiled proof (optionally only kernels whose name starts with a prefix).
Usage: python tools/show_bench.py <bench.json> [kernel-prefix]"""
import json
import sys


def main():
    d = json.load(open(sys.argv[1]))
    prefix = sys.argv[2] if len(sys.argv) > 2 else ""
    roof = d.get("roofline") or {}
    print("value %s %s   ms_per_step %s   single_proof_latency_ms %s" % (d["value"], d["unit"], d["ms_per_step"],
                                                                        d["config"].get("single_proof_latency_ms")))
    print("roofline: kernel %s achieved %s %s frac %s avg_launch_ms %s" % (roof.get("kernel"), roof.get("achieved"), roof.get("unit"),
                                                                         roof.get("frac"), roof.get("avg_launch_ms")))
    for k, v in (roof.get("per_kernel_ms") or {}).items():
        if k.startswith(prefix):
            print("  %-28s %8.3f" % (k, v))


if __name__ == "__main__":
    main()
