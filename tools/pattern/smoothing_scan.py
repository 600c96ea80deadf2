import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import simulate as S, recovered as R
from scipy import ndimage
pts,sig=R.points(); pr=R.pairs()
voc=np.fromfile('tests/golden/small_voc_desc.bin',dtype=np.uint8).reshape(-1,48)
bv=np.unpackbits(voc,axis=1,bitorder='little').astype(float)
C=np.corrcoef(bv.T); iu=np.triu_indices(384,1)
imgs=S.default_images(3)[1:]
kps=[S.keypoints(im) for im in imgs]
scale=2.468
# ring of each pair for a per-ring breakdown
ring=np.concatenate([[q]*n for q,n in enumerate(R.COUNTS)])
inner=(ring[pr[:,0]]<=2)&(ring[pr[:,1]]<=2)   # bits among centre/X/R1
outer=(ring[pr[:,0]]>=4)&(ring[pr[:,1]]>=4)   # bits among R3/R4
def blockcorr(Cs,m):
    idx=np.where(m)[0]; a=Cs[np.ix_(idx,idx)]; b=C[np.ix_(idx,idx)]; i2=np.triu_indices(len(idx),1)
    return np.corrcoef(a[i2],b[i2])[0,1]
for k in (0.58,1.0,1.5):
  for b in (0.0,1.5,3.0,4.5):
    Bs=[]
    for im,xy in zip(imgs,kps):
        imb=ndimage.gaussian_filter(im.astype(float),b) if b>0 else im
        v=S.sample_values(imb,xy,pts*scale,sig*scale*k); Bs.append((v[:,pr[:,0]]>v[:,pr[:,1]]).astype(float))
    B=np.concatenate(Bs); Cs=np.corrcoef(B.T)
    near=(B[:1500]@(1-bv).T+(1-B[:1500])@bv.T).min(1).mean()
    print("k %.2f blur %.1f: all %.3f inner %.3f outer %.3f near %.1f"%(k,b,np.corrcoef(Cs[iu],C[iu])[0,1],blockcorr(Cs,inner),blockcorr(Cs,outer),near),flush=True)
