"""CPU, world_size 2 over gloo: the sharding + alm all-gather logic of pixell_amd/dist.py (the N>1 path of bench.py)."""
import os, sys, socket
import numpy as np
import pytest

def _free_port():
	s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

def _worker(rank, world, port, nmaps, q):
	import torch, torch.distributed as dist
	sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
	from pixell_amd import dist as pd
	os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
	dist.init_process_group("gloo", rank=rank, world_size=world)
	try:
		ncomp, nelem = 3, 21
		def fake_transform(i):      # stands for curvedsky.map2alm on map i: deterministic function of the map index
			g = torch.Generator().manual_seed(1000+i)
			return torch.complex(torch.randn((ncomp, nelem), generator=g, dtype=torch.float64), torch.randn((ncomp, nelem), generator=g, dtype=torch.float64))
		out = pd.map2alm_sharded(list(range(nmaps)), None, 5, [0, 2], fake_transform)
		ref = torch.stack([fake_transform(i) for i in range(nmaps)], 0)
		i0, i1 = pd.shard_range(nmaps, rank, world)
		fin = pd.allgather_alm(ref[i0:i1], nmaps, async_op=True)      # async variant, as overlapped in bench.py
		out2 = fin()
		# the preallocated gather bench.py --config c4 uses (uneven shards are padded to the largest)
		rows = [3*(pd.shard_range(nmaps, r, world)[1]-pd.shard_range(nmaps, r, world)[0]) for r in range(world)]
		mine = ref[i0:i1].reshape(-1, nelem).contiguous()
		g = pd.AlmGather(mine, rows, "cpu", backend="gloo"); g.run(mine)
		out3 = torch.cat(g.result(), 0).reshape(nmaps, ncomp, nelem)
		q.put((rank, bool(torch.equal(out, ref)) and bool(torch.equal(out2, ref)) and bool(torch.equal(out3, ref)), tuple(out.shape)))
	finally:
		dist.destroy_process_group()

@pytest.mark.parametrize("nmaps", [4, 5])
def test_shard_and_allgather_world2(nmaps):
	import torch.multiprocessing as mp
	ctx = mp.get_context("spawn")
	q = ctx.Queue(); port = _free_port()
	procs = [ctx.Process(target=_worker, args=(r, 2, port, nmaps, q)) for r in range(2)]
	for p in procs: p.start()
	res = [q.get(timeout=120) for _ in procs]
	for p in procs: p.join(timeout=60)
	for rank, ok, shape in res:
		assert ok and shape == (nmaps, 3, 21)

def test_shard_ranges():
	from pixell_amd import dist as pd
	for n in [1, 7, 8, 64, 100]:
		for w in [1, 2, 3, 8]:
			r = [pd.shard_range(n, k, w) for k in range(w)]
			assert r[0][0] == 0 and r[-1][1] == n and all(r[k][1] == r[k+1][0] for k in range(w-1))
			assert max(b-a for a, b in r)-min(b-a for a, b in r) <= 1
