"""`python bench.py --gpus N` must start N ranks by itself (round-3 review item 9: outside torchrun the flag used to be a
warning and the line said n_gpus 1).  The reference starts its ranks with `python -m torch.distributed.launch
--nproc_per_node=N run_experiment.py` (README.md:280, src/run_experiment.py:70-82,146-153); bench.py re-executes itself
under torch.distributed.run on 127.0.0.1.  No GPU here: --dry-run-ranks joins the ranks over gloo, all-reduces a one per
rank and prints how many joined."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_flag_spawns_that_many_ranks():
    r = _run(["--gpus", "2", "--dry-run-ranks"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["world_size_env"] == 2 and d["requested"] == 2, d


def test_too_few_devices_is_an_error():
    """without --dry-run-ranks the launcher refuses to start more ranks than there are devices (0 here)"""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2
    assert "device(s) visible" in r.stderr


def test_launcher_and_flag_must_agree():
    """under an external launcher WORLD_SIZE wins over nothing: a mismatch is an error, not a warning"""
    r = _run(["--gpus", "2", "--dry-run-ranks"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0",
                                                           "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert r.returncode == 3        # one rank joined, two were requested
