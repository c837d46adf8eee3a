import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g._load_pkg()
from tests import oracle
w,h=64,48
for name, Y in (("xramp", np.tile(np.arange(w, dtype=np.uint8)*3, (h,1))), ("yramp", np.tile((np.arange(h, dtype=np.uint8)*5)[:,None], (1,w)))):
    ctx = pkg.Context(w,h,yuv_format=0)
    ctx.ref_upload(0,Y); ctx.interp_luma(0)
    got = ctx.download_luma_planes(0).astype(int); want = oracle.interp_luma(Y).astype(int)
    for (py,px) in ((0,2),(2,0),(2,2)):
        bad = np.argwhere(got[py,px]!=want[py,px])
        print(name, py,px,len(bad))
        if len(bad):
            rows = np.unique(bad[:,0]); cols=np.unique(bad[:,1])
            print(" rows", rows[:40]); print(" cols", cols[:40])
            j,i = bad[0]
            print(" at", j,i, "got", got[py,px,j,i-3:i+5], "want", want[py,px,j,i-3:i+5])
    ctx.close()
