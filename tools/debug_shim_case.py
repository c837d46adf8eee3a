#!/usr/bin/env python3
"""debug helper: python tools/debug_shim_case.py <case> [env K=V ...] -- runs jm_hip on a test_jm_shim case and prints its stderr head"""
import os, sys, pathlib, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_jm_shim as t
name = sys.argv[1]
env = dict(a.split("=", 1) for a in sys.argv[2:])
env.setdefault("JMHIP_SHIM_STATS", "1")
d = pathlib.Path(tempfile.mkdtemp())
t.prepare(d, name, frames=4)
want = t.run("jm_plain", d)
import subprocess
e = dict(os.environ); e.update(env)
r = subprocess.run([os.path.join(t.RDIR, "jm_hip"), "-d", "case.cfg"], cwd=d, env=e, capture_output=True, text=True, timeout=900)
print("returncode", r.returncode)
print(r.stdout[-1500:])
got = (open(d / "out.264", "rb").read(), open(d / "out_rec.yuv", "rb").read(), r.stderr)
print("bitstream equal:", got[0] == want[0], " recon equal:", got[1] == want[1])
lines = got[2].splitlines()
print("\n".join(lines[:40]))
print("...")
print("\n".join(lines[-30:]))
