import sys, time, numpy as np
sys.path.insert(0,'.')
from picasso_amd import backend as be
g=np.load('tests/golden/zfit_calib3d.npz')
rng=np.random.default_rng(0)
N=5_000_000
z=rng.uniform(-400,400,N)
sx=np.polyval(g['cx'],z).astype(np.float32)*rng.normal(1,0.02,N).astype(np.float32)
sy=np.polyval(g['cy'],z).astype(np.float32)*rng.normal(1,0.02,N).astype(np.float32)
for _ in range(3):
    t0=time.perf_counter(); zz,sq=be.zfit_arrays(sx,sy,g['cx'],g['cy']); dt=time.perf_counter()-t0
    print(f"zfit {N} locs (host buffers): {dt*1e3:.1f} ms  {N/dt/1e6:.1f} M/s, median |dz| {np.median(np.abs(zz-z)):.2f}")
