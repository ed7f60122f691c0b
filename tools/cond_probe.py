"""How well-conditioned is a discriminating random-init network?  The fp32 oracle against the exact-arithmetic statement of the SAME
network (oracle.forward_ex, accumulate_f64) next to the distance between neighbouring positions -- CPU only, test infrastructure
(tests/test_round4_gpu.py quotes its figures).  python tools/cond_probe.py "(blocks, channels, batch, conv_gain, policy_gain, value_gain)" ..."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
g.package()
from dream_go_amd import synth, api
from oracle import oracle
import bench
positions = bench.real_game_positions(api)
F=lambda a: a.view(np.float16).astype(np.float32)
def probe(blocks,ch,batch,cg,pg,vg):
    text = synth.to_json(synth.make_centered_tensors(blocks,ch,seed=2020,conv_gain=cg,policy_gain=pg,value_gain=vg))
    tensors,_=oracle.load_weights(text); net=oracle.OracleNetwork(tensors)
    f=bench.real_game_features(positions,batch)
    v,p,l,vp=[F(a) for a in net.forward_ex(f,batch)]
    xv,xp,xl,xvp=[F(a) for a in net.forward_ex(f,batch,accumulate_f64=True)]
    P=p.reshape(batch,362); XP=xp.reshape(batch,362)
    d=np.abs(P[:,None,:]-P[None,:,:]).max(axis=2); np.fill_diagonal(d,np.inf)
    pm=P.max(1)
    print(f"{blocks}x{ch} B={batch} conv {cg} gains {pg}/{vg}: pmax {pm.min():.3f}..{pm.max():.3f} argmax {len(set(P.argmax(1)))} value {v.min():.3f}..{v.max():.3f} | oracle vs exact: dp {np.abs(P-XP).max():.2e} dv {np.abs(v-xv).max():.2e} | nearest other {d.min():.2e}", flush=True)
for a in sys.argv[1:]:
    probe(*eval(a))
