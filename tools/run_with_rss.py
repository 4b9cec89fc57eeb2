#!/usr/bin/env python3
"""Run a command as a child process and report its wall time and peak resident set size: run_with_rss.py CMD [ARGS...]"""
import resource
import subprocess
import sys
import time

t0 = time.perf_counter()
rc = subprocess.call(sys.argv[1:])
dt = time.perf_counter() - t0
rss = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 2 ** 20
print(f"[run_with_rss] exit code {rc}, {dt:.1f} s wall, peak RSS {rss:.2f} GiB: {' '.join(sys.argv[1:])}")
sys.exit(rc)
