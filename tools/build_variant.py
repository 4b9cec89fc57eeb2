#!/usr/bin/env python3
"""Build a variant of the library next to the product one: tools/build_variant.py NAME [hipcc flags...]
-> build/libabd_hip_NAME.so (use it with ABD_HIP_LIB=$PWD/build/libabd_hip_NAME.so).
  stamps : -DABD_STAMPS  in-kernel s_memrealtime stamps (tools/probe_stamps.py)
  tuning : -DABD_TUNING  development knobs readable from the environment (abd_host.hpp: tune_int)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

name = sys.argv[1]
flags = sys.argv[2:] or {"stamps": ["-DABD_STAMPS"], "tuning": ["-DABD_TUNING"]}.get(name, [])
out = os.path.join(ROOT, "build", f"libabd_hip_{name}.so")
print(g.build_hip(force=True, extra_flags=flags, lib=out))
