import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g._load_pkg()
from tests import oracle
rng = np.random.default_rng(64*131+48)
w,h=64,48
Y = rng.integers(0,256,(h,w),dtype=np.uint8)
ctx = pkg.Context(w,h,yuv_format=0)
ctx.ref_upload(0,Y); ctx.interp_luma(0)
got = ctx.download_luma_planes(0); want = oracle.interp_luma(Y)
for py in range(4):
    for px in range(4):
        bad = np.argwhere(got[py,px]!=want[py,px])
        print(py,px,len(bad), np.bincount(bad[:,1]%4, minlength=4) if len(bad) else '', bad[:3].tolist())
print(got[0,2,0,16:32]); print(want[0,2,0,16:32]); print(got[0,0,0,16:32])
