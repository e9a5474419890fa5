import sys, json, numpy as np, torch
sys.path.insert(0,'.')
from mcmc_amd.engine import DeviceChains, DeviceProblem
g=json.load(open('tests/golden/mh_trace_3layer_rotate.json'))
nl=g['n_layers']
noise={k:(np.array(v) if isinstance(v,list) else v) for k,v in g['noise_params'].items()}
for rotflag in (0,1):
    prob=DeviceProblem(nl,g['vpvs'],g['periods'],g['obs'],g['noise_dist'],noise,g['bounds'])
    ch=DeviceChains(prob,np.array([g['start']]),g['width0'])
    ch.init_likelihood()
    dev=prob.device
    cauchy=np.tan(np.pi*(np.array(g['u_prop'])-0.5))
    for s in range(8):
        out=ch.steps(torch.tensor([[g['idx'][s]]],dtype=torch.int32,device=dev),torch.tensor([[cauchy[s]]],device=dev),torch.tensor([[g['u_acc'][s]]],device=dev),rotate=rotflag)
        print(rotflag,s,out['accepted'].item(),out['status'].item(),ch.logL.item(),g['steps'][s]['logL'],ch.m[0,:2].tolist(), int(ch.n_cov[0]))
