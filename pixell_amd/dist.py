"""Multi-GPU use of the path: independent maps shard across ranks (one process per GPU), each rank
runs the single-GPU transforms on its slice, and the resulting alm are all-gathered once over RCCL
(torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).  No collective sits on the
data path of a transform."""
import numpy as np

def shard_range(n, rank, world):
	"""contiguous slice [i0, i1) of n independent maps owned by `rank`; sizes differ by at most one"""
	base, rem = divmod(n, world)
	i0 = rank*base+min(rank, rem)
	return i0, i0+base+(1 if rank < rem else 0)

def shard_sizes(n, world):
	return [shard_range(n, r, world)[1]-shard_range(n, r, world)[0] for r in range(world)]

def allgather_alm(local_alm, nmaps, group=None, async_op=False):
	"""local_alm: tensor [n_local, ncomp, nelem] (complex) on this rank -> [nmaps, ncomp, nelem] on every rank.
	Uneven shards are padded to the largest shard so that ONE all_gather_into_tensor moves everything."""
	import torch, torch.distributed as dist
	world = dist.get_world_size(group); rank = dist.get_rank(group)
	sizes = shard_sizes(nmaps, world); nmax = max(sizes)
	assert local_alm.shape[0] == sizes[rank], "local shard has %d maps, expected %d" % (local_alm.shape[0], sizes[rank])
	real = torch.view_as_real(local_alm.contiguous())                    # RCCL has no complex dtype
	if sizes[rank] < nmax:
		pad = torch.zeros((nmax-sizes[rank],)+tuple(real.shape[1:]), dtype=real.dtype, device=real.device)
		real = torch.cat([real, pad], 0)
	out = torch.empty((world*nmax,)+tuple(real.shape[1:]), dtype=real.dtype, device=real.device)
	work = dist.all_gather_into_tensor(out, real.contiguous(), group=group, async_op=async_op)
	def finish():
		if work is not None: work.wait()
		parts = [out[r*nmax:r*nmax+sizes[r]] for r in range(world)]
		return torch.view_as_complex(torch.cat(parts, 0).contiguous())
	return finish if async_op else finish()

def map2alm_sharded(maps, wcs, lmax, spin, transform, group=None):
	"""maps: the FULL list/array of nmaps independent maps (host or device; only this rank's slice is touched).
	transform(map_i) -> alm tensor [ncomp, nelem] for one map (e.g. a closure over curvedsky.map2alm)."""
	import torch, torch.distributed as dist
	world = dist.get_world_size(group); rank = dist.get_rank(group)
	i0, i1 = shard_range(len(maps), rank, world)
	local = [transform(maps[i]) for i in range(i0, i1)]
	local = torch.stack(local, 0) if local else None
	if local is None: raise ValueError("more ranks than maps")
	return allgather_alm(local, len(maps), group=group)

class AlmGather:
	"""Preallocated all-gather of the per-rank alm block [rows_r, nelem] (complex128) into one [world, rows_max, nelem] buffer on
	every rank: ONE all_gather_into_tensor over RCCL per step (direct schedule on the fully connected xGMI mesh), issued on a side
	stream so that it runs under the next transform.  Uneven shards are padded to the largest one (the pad rows are never read).
	backend "gloo" is the CPU / rehearsal path (no device collective: staged through host memory)."""
	def __init__(self, local, rows_per_rank, device, backend="nccl"):
		import torch
		self.rows = list(rows_per_rank); self.world = len(self.rows); self.rmax = max(self.rows); self.backend = backend
		self.nelem = local.shape[-1]
		self.buf = torch.empty((self.world, self.rmax, self.nelem), dtype=local.dtype, device=device)
		self.pad = torch.zeros((self.rmax, self.nelem), dtype=local.dtype, device=device) if min(self.rows) < self.rmax else None
	def run(self, local, stream=None):
		import torch, torch.distributed as dist
		src = local      # even shards, and the full-size shards of an uneven split, go out of the caller's buffer as they are
		if self.pad is not None and local.shape[0] < self.rmax:
			self.pad[:local.shape[0]] = local; src = self.pad
		if self.backend == "nccl":
			dist.all_gather_into_tensor(torch.view_as_real(self.buf).view(self.world, -1), torch.view_as_real(src.contiguous()).view(-1))
		else:
			if stream is not None: stream.synchronize()
			host = [torch.empty(src.shape, dtype=src.dtype) for _ in range(self.world)]
			dist.all_gather(host, src.cpu())
			for r in range(self.world): self.buf[r].copy_(host[r])
	def result(self):
		"""list of the per-rank blocks (views into the gather buffer)"""
		return [self.buf[r, :self.rows[r]] for r in range(self.world)]
	def describe(self):
		return "all_gather_into_tensor (%s), %d ranks x %.3f GB per step" % ("RCCL" if self.backend == "nccl" else self.backend, self.world, self.rmax*self.nelem*16/1e9)
