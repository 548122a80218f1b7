"""debug aid: Legendre kernels at large lmax against the long-double oracle on a subset of rings"""
import sys, os, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixell_amd import sht
from oracle import sht_oracle as so
lmax = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
nr = lmax+2; nph = 8
th = np.arange(nr)*np.pi/(nr-1); th[0] = 1e-4; th[-1] = np.pi-1e-4
sub = np.unique(np.concatenate([np.arange(0, 12), np.arange(12, nr//2, max(1, nr//60)), nr-1-np.arange(0, 12), [nr//2]]))
ms = so._tri_mstart(lmax, lmax)
def kw(t): return dict(theta=t, nphi=np.full(len(t), nph, np.uint64), phi0=np.full(len(t), 0.1), ringstart=np.arange(len(t), dtype=np.uint64)*nph, lmax=lmax, mstart=ms)
for spin in (0, 2):
	alm = so.rand_alm_simple(lmax, 1 if spin == 0 else 2, 6, spin=(spin,))
	out = sht.synthesis(alm=alm, spin=spin, **kw(th)).reshape(-1, nr, nph)
	t0 = time.time(); ref = so.synthesis(alm=alm, spin=spin, **kw(th[sub])).reshape(-1, len(sub), nph); t1 = time.time()
	err = np.max(np.abs(out[:, sub]-ref), axis=(0, 2))/np.max(np.abs(ref))
	print("spin", spin, "synthesis max rel err", err.max(), "oracle %.0fs" % (t1-t0), flush=True)
	bad = np.where(err > 1e-10)[0]
	print("  bad rings:", [(int(sub[i]), "%.3g" % th[sub[i]], "%.2g" % err[i]) for i in bad[:20]], flush=True)
	rng = np.random.default_rng(1)
	pix = np.zeros((out.shape[0], nr, nph)); pix[:, sub] = rng.standard_normal((out.shape[0], len(sub), nph))
	oa = sht.adjoint_synthesis(map=pix.reshape(out.shape[0], -1), spin=spin, **kw(th))
	ra = so.adjoint_synthesis(map=pix[:, sub].reshape(out.shape[0], -1), spin=spin, **kw(th[sub]))
	ra[:, :lmax+1] = ra[:, :lmax+1].real
	d = np.abs(oa-ra)/np.sqrt(np.mean(np.abs(ra)**2))
	print("spin", spin, "adjoint rms rel err", np.sqrt(np.mean(d**2)), "max", d.max(), flush=True)
	if d.max() > 1e-9:
		i = np.argsort(d.max(0))[-8:]
		# which (l,m)
		mm = np.searchsorted(ms.astype(np.int64), i, side="right")-1
		print("  worst (l,m):", [(int(ii-int(ms[m_])), int(m_), "%.2g" % d[:, ii].max()) for ii, m_ in zip(i, mm)], flush=True)
