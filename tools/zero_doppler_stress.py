"""Worst case for the speculative carrier walker: channels sweeping through zero Doppler (legs without a
wrap).  Prints passes / times; output must still equal the oracle.  python tools/zero_doppler_stress.py"""
import sys, time, hashlib, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from __graft_entry__ import load_pkg
from oracle_binding import oracle_run
pkg = load_pkg()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 300
p = pkg.workloads.make_synthetic(n_epochs=E, n_chan=12, seed=99)
for j in range(12):
    f = np.linspace(30 - 5 * j, -20 - 3 * j, E)
    p['f_carr'][:, j] = f
    p['f_code'][:, j] = 1.023e6 + f * 0.0006493506493506494
with pkg.SynthEngine(device=0) as eng:
    t = time.perf_counter()
    iq, st, stats = eng.run_host(p)
    dt = time.perf_counter() - t
print('epochs', E, 'passes', stats['walk_passes'], 'mismatch', stats['chain_mismatch'], 'ms_walk %.1f ms_synth %.1f wall %.1f ms' % (stats['ms_walk'], stats['ms_synth'], dt * 1e3))
if E <= 60:
    ref, _ = oracle_run(p, 260000, 2.6e6)
    print('bit-exact vs oracle:', np.array_equal(ref, iq))
