"""Timing probe (not a product path): phase timestamps of k_conv32 from an instrumented scratch build (tools/ab_libs/libconv32_stamp.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rho_diffusion_amd import hip
from rho_diffusion_amd.engine import ops
dev="cuda"; lib=hip.lib()
N,D,H,W,c=2,128,128,128,32
x=(torch.randn(N,D,H,W,c,device=dev)*0.5).to(torch.bfloat16)
w=ops.prep_conv_weight(torch.randn(c,c,3,3,3,device=dev)*0.05,torch.bfloat16)
b=torch.zeros(c,device=dev); y=torch.empty(N,D,H,W,c,device=dev,dtype=torch.bfloat16)
pa,pb=torch.ones(N,c,device=dev),torch.zeros(N,c,device=dev)
for tag,kw in (("plain",{}),("pre",dict(pre_a=pa,pre_b=pb,pre_silu=True))):
    d=ops.make_conv_desc(x,None,w,b,kernel=(3,3,3),cout=c,split=c,y=y,y2=None,**kw)
    for _ in range(3): ops.conv_launch(d)
    torch.cuda.synchronize()
    buf=np.zeros((512,8),dtype=np.uint64)
    assert lib.rho_dbg_stamps32(C.c_void_p(buf.ctypes.data))==0
    st=buf[:256,:5].astype(np.float64)*10.0/1000.0
    dl=np.diff(st,axis=1)
    print(tag,"us: head (issue loads) %.2f | taps %.2f | barrier %.2f | epilogue %.2f | tile %.2f"%(*np.median(dl,axis=0),np.median(st[:,4]-st[:,0])))
