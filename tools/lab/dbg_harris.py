import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import oracle_lib as O
from okvis2_amd import capi, synth
w,h=752,480
fe=capi.Frontend(w,h,38.0,0,150,700,max_batch=1)
img=synth.corners_image(w,h,11)
d=torch.from_numpy(img).cuda(); sc=torch.empty((h,w),dtype=torch.int32,device='cuda')
fe.harris_score_device(d.data_ptr(),1,sc.data_ptr(),None); torch.cuda.synchronize()
got=sc.cpu().numpy(); ref=O.harris_score(img)
bad=np.argwhere(got!=ref)
print(len(bad), bad[:10], bad[-5:])
if len(bad):
    ys,xs=bad[:,0],bad[:,1]
    print('x unique', np.unique(xs)[:40], 'count per x', np.bincount(xs).nonzero()[0][:50])
    print('y unique', np.unique(ys)[:40])
    for (y,x) in bad[:5]: print(y,x,got[y,x],ref[y,x])
