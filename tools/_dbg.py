import sys; sys.path.insert(0,'.')
import numpy as np, torch
from litepose_amd import arch_zoo, config
from litepose_amd.models import pose_mobilenet
from oracle import synth, net_ref
arch=arch_zoo.get('search-XS'); cfg=config.get_cfg(); sd=synth.make_state_dict(arch)
m=pose_mobilenet.get_pose_net(cfg,cfg_arch=arch); m.load_state_dict(sd)
x=synth.make_images(1,64,seed=7)
taps={}
with torch.no_grad(): net_ref.forward(x,sd,arch,taps=taps)
m(x.cuda()); torch.cuda.synchronize()
for name in ['first','stage.0.0','stage.0.1','stage.1.0','stage.1.1']:
    got=m.tap(name).cpu().numpy().reshape(taps[name].shape); ref=taps[name].numpy()
    d=np.abs(got-ref); print(name, d.max(), np.unravel_index(d.argmax(), d.shape))
    if name=='first':
        print(np.round(d[0,0,:8,:16],3))
