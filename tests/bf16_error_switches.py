"""Round 6: bf16 error of reduced-width ResNet-UNets vs the fp32 oracle (a test tool: the oracle is the checker) with the engine's
decompositions switched off one by one -- sub-pixel phase convs, the fused bridge level, head hoisting: the error level is the same
(0.4 - 4.5 % over seeds), i.e. it is bf16 activation storage through a deep stack, not a decomposition."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'oracle'))
import celldetection_amd as cda, cpn_oracle as orc
from celldetection_amd.synth import calibrate_heads, synth_state_dict
dev=torch.device('cuda:0')
NAMES=('scores','locations','refinement','fourier')
def run(fam, kw, seed, n,h,w, env, subpixel):
    for k in ('CPN_BRIDGE','CPN_PAIR','CPN_S1F','CPN_S1Q','CPN_HOIST'):
        os.environ.pop(k,None)
    os.environ.update(env)
    model=getattr(cda.models,fam)(3, **kw)
    sd=synth_state_dict(model.state_dict(), seed=seed)
    x=torch.rand(n,3,h,w,generator=torch.Generator().manual_seed(seed))
    sd,_=calibrate_heads(sd, lambda s_: orc.core_forward(s_, x))
    model.load_state_dict(sd); model=model.to(dev); model.subpixel=subpixel
    ref=orc.core_forward(sd,x); ref=(torch.sigmoid(ref[0]),)+tuple(ref[1:4])
    got=model.core_forward(x.to(dev))
    return {nm: round(((g.cpu().float()-e).norm()/(e.norm()+1e-12)).item(),4) for nm,g,e in zip(NAMES,got,ref)}
for fam,kw in (('CpnResNet34UNet',dict(backbone_kwargs={'backbone_kwargs':{'base_channel':8}})),('CpnResNet34UNet',dict(backbone_kwargs={'backbone_kwargs':{'base_channel':32}})),('CpnResNet18UNet',dict(backbone_kwargs={'backbone_kwargs':{'base_channel':8}}))):
  for seed in (11,12,13):
    for tag,env,sp in (('default',{},True),('no subpixel',{},False),('no bridge',{'CPN_BRIDGE':'0'},True),('no subpixel, no bridge, no hoist',{'CPN_BRIDGE':'0','CPN_HOIST':'0'},False)):
        print(fam, kw['backbone_kwargs']['backbone_kwargs'], seed, tag, run(fam,kw,seed,1,112,160,env,sp), flush=True)
