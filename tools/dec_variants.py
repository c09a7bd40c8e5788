#!/usr/bin/env python3
"""Development (GPU box): tools/dec_time.py per library variant (tools/ab_build.sh), each in a child process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB = os.path.join(ROOT, "lz4_frame_conduit_amd", "build", "ab")
for n in sys.argv[1:]:
    env = dict(os.environ)
    if n != "base": env["LZ4F_MI355X_LIB"] = os.path.join(AB, "lib_%s.so" % n)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dec_time.py")], env=env, capture_output=True, text=True, timeout=240)
    print("%-8s %s" % (n, (r.stdout.strip().splitlines() or ["FAILED " + r.stderr[-300:]])[-1]), flush=True)
