#!/usr/bin/env python3
"""Development (GPU box): tools/foreign_inst.py for library variants (tools/ab_build.sh), alternating, each in a child process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB = os.path.join(ROOT, "lz4_frame_conduit_amd", "build", "ab")
for rnd in range(2):
    for n in sys.argv[1:]:
        env = dict(os.environ)
        if n != "base": env["LZ4F_MI355X_LIB"] = os.path.join(AB, "lib_%s.so" % n)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "foreign_inst.py"), "6"], env=env, capture_output=True, text=True, timeout=300)
        v = [float(l.split("in front of them ")[1].split(")")[0]) for l in r.stdout.splitlines() if "in front of them" in l]
        t = [float(l.split("total ")[1].split()[0]) for l in r.stdout.splitlines() if "total " in l]
        print("%-6s in front of the parse kernels: %s  total: %s" % (n, " ".join("%.3f" % x for x in v), " ".join("%.3f" % x for x in t)), flush=True)
