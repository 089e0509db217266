"""Walker convergence and timing on real RINEX scenarios (GPU box): python tools/real_scenarios_passes.py"""
import sys, time, numpy as np
sys.path.insert(0,'.')
from __graft_entry__ import load_pkg
pkg=load_pkg()
import torch
torch.cuda.init(); torch.zeros(1,device='cuda')
for start,dur in [('2022/02/20,12:00:00',120),('2022/02/20,06:00:00',120),('2022/02/20,18:30:00',300)]:
    rows=pkg.Scenario('tests/golden/20feb2022.rnx', llh=(-6,51,100), start=start, duration_s=dur, iono_enable=False).all()
    f=rows['f_carr']; act=rows['prn'][0]>0
    print(start,dur,'SVs',int(act.sum()),'doppler range at start',np.round(f[0][act]).tolist())
    with pkg.SynthEngine(device=0) as eng:
        eng.plan(rows)
        out=torch.empty(eng.output_bytes()//2,dtype=torch.int16,device='cuda')
        eng.execute(out.data_ptr()); st,stats=eng.finish()
        t=time.perf_counter(); eng.execute(out.data_ptr()); st,stats=eng.finish(); dt=time.perf_counter()-t
    print('   passes',stats['walk_passes'],'mismatch',stats['chain_mismatch'],'ms_walk %.2f ms_synth %.2f total %.2f ms'%(stats['ms_walk'],stats['ms_synth'],dt*1e3))
