import sys; sys.path.insert(0,'.')
import torch, smsut_amd
from smsut_amd import ops, _hip as H
B,h,ci,co,ks=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3]),int(sys.argv[4]),3
mode=sys.argv[5]
x=torch.randn(B,ci,h,h,device='cuda').contiguous(memory_format=torch.channels_last)
gy=torch.randn(B,co,h,h,device='cuda').contiguous(memory_format=torch.channels_last)
w=ops.new_weight(co,ci,ks,ks,device='cuda'); w.copy_(torch.randn(co,ci,ks,ks,device='cuda')*0.05)
y=torch.empty_like(gy); gx=torch.empty_like(x); gw=torch.empty_like(w)
ws=torch.empty(H.call("smsut_conv2d_wgrad_mfma_ws",B,h,h,ci,co,ks),device='cuda')
for _ in range(5):
    if mode=='fwd': H.call("smsut_conv2d_fwd_mfma",x,w,y,B,h,h,ci,co,ks,0,H.stream_ptr())
    elif mode=='dgrad': H.call("smsut_conv2d_fwd_mfma",gy,w,gx,B,h,h,co,ci,ks,1,H.stream_ptr())
    else: H.call("smsut_conv2d_wgrad_mfma",x,gy,gw,ws,B,h,h,ci,co,ks,H.stream_ptr())
torch.cuda.synchronize()
