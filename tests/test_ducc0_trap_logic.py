"""tests/test_ducc0_live.py skips until a box has ducc0.  So that it does not fail there for reasons of its own (keywords, shapes, masks), its body
runs here once with the long-double ORACLE standing in for ducc0.sht.experimental -- the same keyword interface (oracle/sht_oracle.py:412-572) -- on the
simulator, at the two smallest grids.  This checks the trap, not the product against ducc0."""
import sys, types, importlib
import numpy as np, pytest
from pixell_amd import _lib

@pytest.mark.hostsim
@pytest.mark.parametrize("geometry,nt,nph,lmax,spin", [("F1", 64, 128, 63, 0), ("F1", 65, 130, 64, 2), ("CC", 65, 128, 63, 1)])
def test_trap_body_with_the_oracle_as_ducc0(monkeypatch, geometry, nt, nph, lmax, spin):
	assert _lib.is_hostsim()
	from oracle import sht_oracle as so
	fake = types.ModuleType("ducc0"); fake.__version__ = "oracle-stand-in"; fake.sht = types.ModuleType("ducc0.sht"); fake.sht.experimental = so
	monkeypatch.setitem(sys.modules, "ducc0", fake)
	monkeypatch.delitem(sys.modules, "test_ducc0_live", raising=False)
	sys.path.insert(0, __file__.rsplit("/", 1)[0])
	try: live = importlib.import_module("test_ducc0_live")
	finally: sys.path.pop(0)
	assert live.ducc0 is fake
	live.test_against_ducc0(geometry, nt, nph, lmax, spin)
	sys.modules.pop("test_ducc0_live", None)
