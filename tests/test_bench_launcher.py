"""`python bench.py --gpus N` must run N ranks by itself (it ran one and printed n_gpus 1).  Without a GPU what can be
checked is the launcher and the rank plumbing: `--dry-run` makes no HIP call -- it generates each rank's streams, builds the gloo
group, barriers, MAX-reduces a made-up time, gathers the seeds and prints the one line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=300)
    return r


def _line(r):
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr)  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_with_disjoint_seeds():
    r = _run(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr
    out = _line(r)
    assert out["n_gpus"] == 2 and out["dry_run"] is True
    assert [x["rank"] for x in out["ranks"]] == [0, 1] and [x["local_rank"] for x in out["ranks"]] == [0, 1]
    seeds = [s for x in out["ranks"] for s in x["seeds"]]
    assert len(set(seeds)) == len(seeds)          # every rank generates its own streams
    assert out["elapsed_max_s"] == 2.0            # MAX over ranks (rank r reports 1 + r)
    # every rank reports what it ran on which device, so that a rank that silently ran half the batch (or the wrong pictures) shows in the one line
    assert [x["device"] for x in out["ranks"]] == [0, 1] and all(x["clips_per_gpu"] == 24576 and "verified_ok" in x for x in out["ranks"])


def test_ranks_split_the_host_and_announce_the_product_leg():
    """N > 1: every rank takes its share of the host (cpus pinned, MOBI_PARSE_THREADS = a thread per two of ITS cpus) and runs the product
    path too (frame-parallel groups, all ranks at once); the dry run shows the per-rank fields the real line fills"""
    out = _line(_run(["--gpus", "2", "--dry-run"], drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MOBI_PARSE_THREADS")))
    ncpu = len(os.sched_getaffinity(0))
    for x in out["ranks"]:
        assert "end_to_end_groups" in x and x["host"]["cpus_visible"] == ncpu
        if ncpu >= 2:
            assert x["host"]["cpus_pinned"] == ncpu // 2 and x["host"]["parse_threads"] == max(2, min(64, x["host"]["cpus_pinned"] // 2))


def test_a_single_rank_on_another_device_ordinal():
    """LOCAL_RANK names the device (one rank per GPU): a lone rank on GPU 1 reports device 1, not 0"""
    out = _line(_run(["--gpus", "1", "--dry-run"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "1"}, drop=()))
    assert out["ranks"][0]["device"] == 1 and out["ranks"][0]["local_rank"] == 1


def test_gpus_1_is_one_rank_and_the_line_is_unchanged_in_shape():
    out = _line(_run(["--gpus", "1", "--dry-run"]))
    assert out["n_gpus"] == 1 and len(out["ranks"]) == 1


def test_rank_count_must_match_gpus():
    # under an external launcher the environment decides the ranks: a mismatch is refused, not mislabelled
    r = _run(["--gpus", "4", "--dry-run"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_more_gpus_than_the_node_has_is_refused():
    # (no GPU here: device_count() == 0)
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "HIP device" in r.stderr
