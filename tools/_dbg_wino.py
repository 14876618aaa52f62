import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from simpleinfer_amd import hipops
np.set_printoptions(linewidth=200, precision=3, suppress=True)
n,h,w,ic,oc=1,8,16,16,32
wt=np.zeros((oc,ic,3,3),np.float32)
for c in range(ic): wt[c,c,1,1]=1.0
for ch in [0,4,8,12]:
    x=np.zeros((n,h,w,ic),np.float32); x[0,:,:,ch]=1.0
    got=hipops.conv2d_winograd(x,wt,None,(1,1))
    print("input ch",ch,"-> output channel sums", {o: float(got[0,:,:,o].sum()) for o in range(oc) if abs(got[0,:,:,o]).sum()>0})
# all-ones weights on center tap from channel 0 only to every oc: shows which V planes are non-zero
x=np.zeros((n,h,w,ic),np.float32); x[0,:,:,1]=1.0
wt=np.zeros((oc,ic,3,3),np.float32); wt[:,1,1,1]=1.0
got=hipops.conv2d_winograd(x,wt,None,(1,1)); print("ch1 broadcast:", float(got.sum()))
