"""CPU: the C port used for bench.py's cpu_baseline agrees with the long-double oracle."""
import numpy as np
import pytest
from oracle import sht_oracle as so, sht_port

@pytest.mark.parametrize("spin,lmax,nr", [(0, 40, 42), (2, 40, 42), (1, 33, 35), (0, 200, 202), (2, 150, 152)])
def test_port_matches_oracle(spin, lmax, nr):
	theta = np.arange(nr)*np.pi/(nr-1)
	nc = 1 if spin == 0 else 2
	alm = so.rand_alm_simple(lmax, nc, 7, spin=(spin,)); ms = so._tri_mstart(lmax, lmax)
	ref = so.alm2leg(alm, spin, lmax, lmax, ms, theta.astype(np.longdouble))          # [nc, nr, nm]
	msel = np.arange(lmax+1)
	a = np.zeros((lmax+1, nc, lmax+1), complex)
	for m in msel: a[m, :, m:] = alm[:, int(ms[m])+m:int(ms[m])+lmax+1]
	out = sht_port.leg(spin, lmax, msel, theta, alm=a)                              # [nm, nc, nr]
	assert np.max(np.abs(np.transpose(out, (1, 2, 0))-ref)) < 1e-11*np.max(np.abs(ref))
	rng = np.random.default_rng(1)
	lg = rng.standard_normal((nc, nr, lmax+1))+1j*rng.standard_normal((nc, nr, lmax+1))
	ra = so.leg2alm(lg, spin, lmax, lmax, ms, theta.astype(np.longdouble), alm.shape[1])
	oa = sht_port.leg(spin, lmax, msel, theta, leg=np.transpose(lg, (2, 0, 1)))
	for m in msel:
		l0 = max(m, spin)
		assert np.max(np.abs(oa[m, :, l0:]-ra[:, int(ms[m])+l0:int(ms[m])+lmax+1])) < 1e-11*np.max(np.abs(ra))
