"""d_act of the reverse scan with / without the in-launch input gradient, and that gradient against float64, over the block
structure's edge lengths (gru_scan_bwd_feed.hip, LOOPDX).  python tools/dx_inloop_check.py"""
import torch, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from hpmn_amd import ops
dev=torch.device('cuda:0')
H,B=64,5
for D in (16,32):
  for T in (17,18,33,34,50,200,1001):
    g=torch.Generator(device="cpu").manual_seed(T*10+D)
    def w(*shape, scale=0.3): return (torch.randn(*shape, generator=g)*scale).to(dev)
    gates=torch.rand(B,T,3*H,generator=g); gates[...,2*H:]=gates[...,2*H:]*2-1; gates=gates.to(dev)
    wg,wc,hs=w(D+H,2*H),w(D+H,H),w(B,T+1,H,scale=0.7)
    d_last=w(B,H,scale=0.1); d_y=w(B,T,H,scale=0.1)
    plain=ops.gru_scan_bwd(wg,wc,D,hs,gates,d_last,d_y,1)
    dx=torch.full((B,T,D),7.0,device=dev)
    d_act=ops.gru_scan_bwd(wg,wc,D,hs,gates,d_last,d_y,1,d_x=dx)
    torch.cuda.synchronize()
    diff=(plain-d_act).abs()
    bad=(diff>0).nonzero()
    w64=torch.cat([wg[:D].double(),wc[:D].double()],dim=1)
    want=d_act.double()@w64.t()
    e=(dx.double()-want).abs()
    print(D,T,'d_act maxdiff',float(diff.max()),'rel',float(diff.max()/plain.abs().max()),'nbad',len(bad), 'first', bad[0].tolist() if len(bad) else None, 'dx err', float(e.max()/want.abs().max()))
