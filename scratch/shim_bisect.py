import sys, tempfile, pathlib
sys.path.insert(0, '.')
from tests import test_jm_shim as t
name = sys.argv[1]
d = pathlib.Path(tempfile.mkdtemp())
t.prepare(d, name)
want = t.run("jm_plain", d)
for m in (sys.argv[2:] or ["01", "04", "08", "10", "20", "40", "ff"]):
    got = t.run("jm_hip", d, {"JMHIP_SHIM": m, "JMHIP_SHIM_STATS": "1"})
    print(m, got[0] == want[0], got[1] == want[1], flush=True)
    if m == "ff": print(got[2][-900:])
