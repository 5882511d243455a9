"""`python bench.py --gpus N` typed bare (no launcher, WORLD_SIZE unset) must start its own N workers the way the driver
would -- torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 -- as a child process, relay rank 0's single
JSON line and the exit code (VERDICT r5 task 6).  Checked here without a GPU through --dry-launch: the workers join a
gloo group over the launcher's rendezvous and rank 0 prints what every rank was handed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=300)


def test_bare_gpus_2_spawns_two_ranks_and_relays_one_json_line():
    r = _run(["--gpus", "2", "--dry-launch"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # rank 0's line and nothing else on stdout
    out = json.loads(lines[0])
    assert out["dry_launch"] is True and out["n_gpus"] == 2
    ranks = sorted(out["ranks"], key=lambda d: d["rank"])
    assert [d["rank"] for d in ranks] == [0, 1]
    assert [d["local_rank"] for d in ranks] == [0, 1]
    assert all(d["world_size"] == 2 for d in ranks)
    assert all(d["master"].startswith("127.0.0.1:") for d in ranks)
    assert len({d["pid"] for d in ranks}) == 2          # two processes, one per GPU
    # what multi-process GPU work on this pool needs is in every worker's environment
    assert all(d["hsa_enable_ipc_mode_legacy"] == "0" for d in ranks)
    assert all(d["gpu_max_hw_queues"] for d in ranks)


def test_gpus_mismatch_with_a_launcher_environment_is_an_error():
    r = _run(["--gpus", "2", "--dry-launch"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert b"WORLD_SIZE=1" in r.stderr
