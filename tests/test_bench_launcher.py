"""bench.py started as `python bench.py --gpus N` brings up N ranks itself (one process per GPU) and rank 0 prints
exactly one JSON line that says how many ranks it saw."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def test_launcher_starts_two_ranks_without_a_gpu():
    rc, out, err = _run(["--gpus", "2", "--dry-run"])
    assert rc == 0, err
    lines = [x for x in out.splitlines() if x.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_seen"] == 2 and d["dry_run"] is True


def test_dry_run_of_the_eight_gpu_configuration():
    """The line BASELINE.json's configs[2] is run with -- 64M particles x 360 beams on the 16384^2 grid over eight
    GPUs -- through the launcher and the rendezvous, without a GPU: eight ranks, the layout of the global particle
    vector, no limit of the library exceeded."""
    rc, out, err = _run(["--gpus", "8", "--particles", "8388608", "--beams", "360", "--map", "16384", "--dry-run"], timeout=600)
    assert rc == 0, err[-2000:]
    d = json.loads([x for x in out.splitlines() if x.strip()][0])
    assert d["n_gpus"] == 8 and d["world_size_seen"] == 8
    plan = d["plan"]
    assert plan["particles_total"] == 1 << 26 and plan["problems"] == []
    assert plan["first_index_per_rank"][7] == 7 * (1 << 23) and plan["chunks_of_8192_per_rank"] == 1024 and plan["fused_tail"]
    assert plan["particle_state_bytes_per_gpu"] < 4 << 30
    assert plan["evaluations_per_step"] == float(1 << 26) * 360
    # a layout the library refuses is reported before anything is launched on a GPU
    rc, out, err = _run(["--gpus", "2", "--particles", "100000", "--dry-run"])
    assert rc != 0
    d = json.loads([x for x in out.splitlines() if x.strip()][0])
    assert "8192" in d["plan"]["problems"][0]


def test_launcher_refuses_more_ranks_than_gpus():
    """The launcher counts GPUs from the driver's topology files (no torch, no HIP in the parent) and refuses at once;
    where there is no topology to read (a box without the driver) the ranks fail by themselves and it reports that."""
    sys.path.insert(0, ROOT)
    import bench
    have = bench.visible_gpus()
    rc, out, err = _run(["--gpus", str((have or 0) + 2)])
    assert rc != 0 and out.strip() == ""
    if have is not None:
        assert "GPU(s) visible" in err


def test_gpu_count_from_topology_files(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    for k, simd in enumerate((0, 1024, 1024, 0, 1024)):            # CPU nodes have no SIMDs
        d = tmp_path / str(k)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (0 if simd else 64, simd))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpus(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpus(str(tmp_path)) == 2
    assert bench.visible_gpus(str(tmp_path / "nothing")) is None
    src = open(BENCH).read()
    head = src[src.index("def launch_ranks"):src.index("def cpu_baseline")]
    assert "import torch" not in head                            # the parent of the ranks stays clear of the GPU runtime


def test_a_failing_rank_fails_the_launcher():
    # --share-gpu without gloo is refused inside the ranks: the parent must report failure, not hang
    rc, out, err = _run(["--gpus", "2", "--share-gpu", "--backend", "nccl"], timeout=120)
    assert rc != 0 and out.strip() == ""


SMALL = ["--steps", "2", "--warmup", "1", "--particles", "65536", "--beams", "36", "--map", "512", "--cpu-particles", "0"]


@pytest.mark.gpu
def test_one_rank_line():
    rc, out, err = _run(["--gpus", "1"] + SMALL)
    assert rc == 0, err[-2000:]
    lines = [x for x in out.splitlines() if x.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["bound"] == "valu"
    assert d["config"]["particles_per_gpu"] == 65536


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_through_the_launcher():
    """The real multi-process path (two rank processes, cross-rank resampling) on the test box's one GPU: gloo
    carries the exchanges, as RCCL refuses two ranks on one device."""
    rc, out, err = _run(["--gpus", "2", "--backend", "gloo", "--share-gpu"] + SMALL)
    assert rc == 0, err[-2000:]
    lines = [x for x in out.splitlines() if x.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_seen"] == 2 and d["value"] > 0
    assert "device-driven" in d["config"]["sharding"] and d["exchange"]["host_reads_per_step"] == 1
    rc, out, err = _run(["--gpus", "2", "--backend", "gloo", "--share-gpu", "--host-collectives"] + SMALL)
    assert rc == 0, err[-2000:]
    d = json.loads([x for x in out.splitlines() if x.strip()][0])
    assert d["n_gpus"] == 2 and "host-combined" in d["config"]["sharding"]
