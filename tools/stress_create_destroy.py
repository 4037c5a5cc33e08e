import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from mrs_optic_flow_amd import FftMethod, FastSpacedBMMethod, ScaleRotationEstimator
t0=time.time()
free0 = torch.cuda.mem_get_info()[0]
rng=np.random.default_rng(0)
import sys as _s
N_CYC = int(_s.argv[1]) if len(_s.argv) > 1 else 150
for i in range(N_CYC):
    n = [32,64,120,128][i%4]
    fm = FftMethod(n*2, n, 80.0, peak_model=i%2)
    f = rng.integers(0,256,(n*2,n*2),dtype=np.uint8)
    fm.processImage(f); fm.processImage(np.roll(f,(1,2),(0,1)))
    del fm
    if i%10==0:
        bm = FastSpacedBMMethod(16, 8, 8, (96, 160)); bm.processImage(rng.integers(0,256,(96,160),dtype=np.uint8)); del bm
    if i%25==0:
        sr = ScaleRotationEstimator(240, 40.0); sr.processImage(rng.integers(0,256,(240,240),dtype=np.uint8)); del sr
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("cycles ok in %.1fs; device free before %.1f MB after %.1f MB"%(time.time()-t0, free0/1e6, free1/1e6))
