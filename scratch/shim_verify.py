import sys, tempfile, pathlib
sys.path.insert(0, '.')
from tests import test_jm_shim as t
name = sys.argv[1]
d = pathlib.Path(tempfile.mkdtemp())
t.prepare(d, name)
got = t.run("jm_hip", d, {"JMHIP_SHIM": "05", "JMHIP_SHIM_VERIFY": "1"})
lines = [l for l in got[2].splitlines() if "VERIFY" in l]
print(len(lines)); print("\n".join(lines[:40]))
