"""A/B of the RoI crop + pool forward kernels: block per cell (default) vs channel slices per XCD (MTLSSL_ROI_FWD=xcd)
with 8 / 4 / 2 / 1 slices. Usage: python tools/bench_roi_fwd.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.build()
from mtl_ssl_amd import ops
torch.manual_seed(0)
for (B,H,W,C,R,crop,pk) in [(2,38,64,1024,512,14,2),(2,38,64,1024,2064,14,2),(2,38,64,1024,128,14,2),(1,38,50,512,256,14,2),(1,50,84,1088,256,17,1)]:
    feat = torch.randn(B,H,W,C,device="cuda")
    yx = torch.rand(R,2,device="cuda")*0.7; hw = torch.rand(R,2,device="cuda")*0.3+0.02
    boxes = torch.cat([yx,yx+hw],1).contiguous()
    bi = (torch.arange(R,device="cuda")*B//R).int()
    res=[]
    for tag,env in (("cells",{}),("s8",{"MTLSSL_ROI_FWD":"xcd","MTLSSL_ROI_SLICES":"8"}),("s4",{"MTLSSL_ROI_FWD":"xcd","MTLSSL_ROI_SLICES":"4"}),("s2",{"MTLSSL_ROI_FWD":"xcd","MTLSSL_ROI_SLICES":"2"}),("s1",{"MTLSSL_ROI_FWD":"xcd","MTLSSL_ROI_SLICES":"1"})):
        for k in ("MTLSSL_ROI_FWD","MTLSSL_ROI_SLICES"): os.environ.pop(k,None)
        os.environ.update(env)
        for _ in range(3): ops.roi_crop_pool_fwd(feat,boxes,bi,crop,pk,pk)
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): ops.roi_crop_pool_fwd(feat,boxes,bi,crop,pk,pk)
        e.record(); e.synchronize()
        res.append("%s %.1f us" % (tag, s.elapsed_time(e)*50))
    print((B,H,W,C,R,crop,pk), " | ".join(res))
