import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import simulate as S, recovered as R
from scipy import ndimage
pts,sig=R.points(); pr=R.pairs()
voc=np.fromfile('tests/golden/small_voc_desc.bin',dtype=np.uint8).reshape(-1,48)
bv=np.unpackbits(voc,axis=1,bitorder='little').astype(float)
C=np.corrcoef(bv.T); iu=np.triu_indices(384,1)
def ham(A,B): return (A@(1-B).T+(1-A)@B.T)
imgs=S.default_images(4)
for label,ims in (("real image",imgs[:1]),("synthetic",imgs[1:])):
  for blur,scale in ((0,1.0),(0,1.6),(0,2.4),(2.0,1.6)):
    bits=[]
    for img in ims:
        xy=S.keypoints(img); im=ndimage.gaussian_filter(img.astype(float),blur) if blur>0 else img
        v=S.sample_values(im,xy,pts*scale,sig*scale)
        bits.append((v[:,pr[:,0]]>v[:,pr[:,1]]).astype(float))
    B=np.concatenate(bits); Cs=np.corrcoef(B.T)
    near=ham(B[:2000],bv).min(1).mean()
    rnd=(np.random.default_rng(0).random((2000,384))<0.5).astype(float)
    print(label,"blur",blur,"scale",scale,"n",len(B),"corr of corr matrices %.3f"%np.corrcoef(Cs[iu],C[iu])[0,1],"nearest word %.1f (random %.1f)"%(near,ham(rnd,bv).min(1).mean()),flush=True)
