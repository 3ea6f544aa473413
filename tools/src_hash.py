#!/usr/bin/env python3
"""Print bench.kernel_src_hash(): sha256 (16 hex digits) over the comment-stripped kernel sources and the Makefile. The
csrc/Makefile stamps it into libamdzk.so (build_stamp.h -> amdzk_build_info()); tools/summarize_prof.py stamps the same
value into the rocprofv3 summaries under profiles/."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (no torch, no HIP at import time)

print(bench.kernel_src_hash())
