"""bench.py --gpus N without a launcher must itself create N ranks (one process per GPU) and give each
rank its dist.shard_bounds range of the global batch.  CPU rehearsal: `--dry-run` uses the gloo backend and
makes no GPU call (the loop being sharded is arboris/core.py:1356-1363, one independent world per index)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from arboris_python_amd.dist import shard_bounds


def _env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = ""          # nothing below may need a GPU
    return env


def test_gpus_2_starts_two_ranks_and_shards():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--batch", "96"],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["requested_gpus"] == 2
    assert r["global_batch"] == 192
    assert r["shards"] == [list(shard_bounds(192, k, 2)) for k in range(2)] == [[0, 96], [96, 192]]
    assert r["gather_ok"] is True


def _dry(argv):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])


def test_eight_ranks_rehearsal_configs_4_and_5_and_a_ragged_batch():
    """The driver's N = 8 launch, rehearsed on CPU (gloo, eight processes): BASELINE config 4's 2048 worlds per GPU = 16384,
    config 5's 8192 per GPU = 65536, and a global batch that is NOT a multiple of eight (ragged last shard); every rank takes
    its dist.shard_bounds range, state and per-rollout costs come back in world order."""
    r = _dry(["--gpus", "8", "--dry-run", "--config", "4"])
    assert r["n_gpus"] == 8 and r["global_batch"] == 16384 and r["gather_ok"] is True
    assert r["shards"] == [[2048 * k, 2048 * (k + 1)] for k in range(8)]
    r = _dry(["--gpus", "8", "--dry-run", "--config", "5"])
    assert r["global_batch"] == 65536 and r["shards"][-1] == [57344, 65536] and r["gather_ok"] is True
    r = _dry(["--gpus", "8", "--dry-run", "--config", "5", "--global-batch", "65531"])
    assert r["shards"] == [list(shard_bounds(65531, k, 8)) for k in range(8)] and r["shards"][-1] == [57344, 65531]
    assert r["gather_ok"] is True


def test_single_rank_dry_run_and_mismatch_is_refused():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], env=_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    r = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert r["n_gpus"] == 1 and r["shards"] == [[0, 4096]]
    # a launcher that created a different number of ranks than --gpus says: no mislabelled line
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 2
    assert b"refusing" in p.stderr


def test_launcher_command_is_torch_distributed_run():
    import bench
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "20"])
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "20"] and cmd[-5].endswith("bench.py")
