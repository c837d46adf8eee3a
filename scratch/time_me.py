import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g._load_pkg()
from h264_amd.jmhip import ME_MB_DTYPE
W,H,R = 1920,1088,32
rng = np.random.default_rng(20260410)
B = rng.integers(0,256,(H//8+8, W//8+8))
B = np.kron(B, np.ones((8,8)))
k = np.ones(9)/9
B = np.apply_along_axis(lambda m: np.convolve(m,k,mode='same'),1,B); B = np.apply_along_axis(lambda m: np.convolve(m,k,mode='same'),0,B)
ref = np.clip(np.round(B[32:32+H,32:32+W]+rng.normal(0,2,(H,W))),0,255).astype(np.uint8)
cur = np.clip(np.round(B[32-3:32-3+H,32+5:32+5+W]+rng.normal(0,2,(H,W))),0,255).astype(np.uint8)
U = np.full((H//2,W//2),128,np.uint8)
ctx = pkg.Context(W,H,yuv_format=1,max_refs=1,search_range=R)
ctx.ref_upload(0,ref,U,U); ctx.cur_upload(cur,U,U)
mbs = np.zeros((W//16)*(H//16), dtype=ME_MB_DTYPE)
for i in range(len(mbs)):
    mbs[i]['mb_x'], mbs[i]['mb_y'] = i%(W//16), i//(W//16)
    mbs[i]['ref_is_0']=1
mbs['pred_mv'] = rng.integers(-8,9,(len(mbs),41,2))
prm = pkg.MeParams(); prm.search_mode=-1; prm.search_range=R; prm.rdopt=1; prm.level_mv_min=-511; prm.level_mv_max=511
lam = int(65536*np.sqrt(0.85*2**((28-12)/3.0))+0.5)
prm.lambda_[0]=prm.lambda_[1]=prm.lambda_[2]=lam; prm.subpel=1; prm.partition_mask=(1<<41)-1
ctx.timing_enable(True)
for it in range(3):
    ctx.interp_luma(0); ctx.interp_chroma(0)
    ctx.me_frame_async(prm, mbs); ctx.sync()
    print(it, {k:(round(v[0],3),v[1]) for k,v in ctx.timing_read().items()}, flush=True)
res = ctx.me_results(len(mbs))
print("mv16x16 hist", np.unique(res['mv'][:,0,:], axis=0, return_counts=True)[0][:5])
# uniform preds (FastFull-like)
mbs['pred_mv'] = np.repeat(rng.integers(-8,9,(len(mbs),1,2)),41,axis=1)
for it in range(2):
    ctx.me_frame_async(prm, mbs); ctx.sync()
    print('uniform', {k:(round(v[0],3),v[1]) for k,v in ctx.timing_read().items()}, flush=True)
