"""bench.py's own N-rank launcher (north_star: "reported at 1, 2, 4 and 8 GPUs"), exercised without a GPU: the parent must start
N child ranks before anything touches a device, relay rank 0's JSON line, and fail loudly when a rank dies, when --gpus
disagrees with the launcher's WORLD_SIZE, or when the machine has fewer GPUs than ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _run(args, env, timeout=240):
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout)


def test_parent_starts_the_ranks_and_relays_rank0_line():
    r = _run(["--gpus", "2", "--launch-dry-run"], _clean_env())
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # exactly ONE JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["gpus_arg"] == 2 and out["rank_sum"] == 3.0
    assert out["master_addr"] == "127.0.0.1"
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node 2" in r.stderr


def test_a_dying_rank_fails_the_run():
    r = _run(["--gpus", "2", "--launch-dry-run"], _clean_env(BENCH_DRYRUN_FAIL_RANK="1"))
    assert r.returncode != 0
    assert "failed" in r.stderr


def test_gpus_must_match_the_launchers_world_size():
    r = _run(["--gpus", "4", "--launch-dry-run"], _clean_env(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"))
    assert r.returncode == 2
    assert "does not match WORLD_SIZE" in r.stderr
    assert r.stdout.strip() == ""


def test_more_ranks_than_gpus_is_refused_before_anything_runs():
    import torch
    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 2)], _clean_env())
    assert r.returncode == 3
    assert "only %d GPU(s) are visible" % have in r.stderr
    assert r.stdout.strip() == ""


def test_the_parent_counts_gpus_without_loading_the_hip_runtime():
    """the parent of the ranks must not have initialised the GPU when it starts the launcher: it counts devices from the KFD topology
    in sysfs (or in a throw-away child), so neither libamdhip64 nor torch is in its address space at that point"""
    r = _run(["--gpus", "64"], _clean_env(BENCH_REPORT_MAPS="1"))
    assert r.returncode == 3                                # 64 GPUs are not there: refused before anything runs
    line = [ln for ln in r.stderr.splitlines() if "parent maps:" in ln]
    assert len(line) == 1, r.stderr
    rep = json.loads(line[0].split("parent maps:", 1)[1])
    assert rep["hip_mapped"] is False and rep["torch_imported"] is False
    assert rep["visible_gpus"] < 64


def test_visible_device_variables_restrict_the_count():
    r = _run(["--gpus", "2"], _clean_env(BENCH_REPORT_MAPS="1", HIP_VISIBLE_DEVICES="0"))
    assert r.returncode == 3
    rep = json.loads([ln for ln in r.stderr.splitlines() if "parent maps:" in ln][0].split("parent maps:", 1)[1])
    assert rep["visible_gpus"] <= 1


def test_schur_form_reaches_the_ranks():
    """--schur-form is part of the command line every rank gets (the scaling run can ask for north_star's rows + all-gather form)"""
    r = _run(["--gpus", "2", "--launch-dry-run", "--schur-form", "rows"], _clean_env())
    assert r.returncode == 0, r.stderr
    assert "--schur-form rows" in r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert out["schur_form"] == "rows"
