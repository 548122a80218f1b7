"""CPU: `python bench.py --gpus N` really starts N ranks (the driver's contract; round 2's bench parsed the flag and ignored it).
--dry runs the launch / sharding / gather skeleton over gloo without any transform, so this needs no GPU."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def _run(args, env_extra=None, timeout=300):
	env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
	env["OMP_NUM_THREADS"] = "1"
	if env_extra: env.update(env_extra)
	return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")]+args, env=env, capture_output=True, text=True, timeout=timeout)

def _line(p):
	lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
	assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
	return json.loads(lines[0])

def test_gpus_flag_spawns_ranks_c4():
	r = _line(_run(["--gpus", "2", "--dry", "--config", "c4", "--steps", "2", "--warmup", "1"], {"PXS_BENCH_NBATCH": "5"}))
	assert r["n_gpus"] == 2 and r["rccl_ranks_seen"] == 2 and r["dry"] is True
	assert r["scaling"] == "strong" and r["config"]["maps_total"] == 5 and r["config"]["maps_per_gpu"] == 3

def test_gpus_flag_spawns_ranks_weak():
	r = _line(_run(["--gpus", "2", "--dry", "--steps", "1", "--warmup", "0"]))
	assert r["n_gpus"] == 2 and r["rccl_ranks_seen"] == 2 and r["scaling"] == "weak" and r["config"]["maps_total"] == 2

def test_single_rank_dry():
	r = _line(_run(["--dry", "--steps", "1", "--warmup", "0"]))
	assert r["n_gpus"] == 1

def test_world_size_must_match_the_flag():
	p = _run(["--gpus", "2", "--dry"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
	assert p.returncode != 0 and "--gpus 2" in p.stderr and "WORLD_SIZE=1" in p.stderr
