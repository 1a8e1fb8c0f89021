"""bench.py's own multi-rank launcher on CPU: `python bench.py --gpus 2` with no launcher around it must start two rank
processes, build the library communicator from a file-borne id (no torch), shard the candidate axis and print ONE JSON line
whose `ranks` the communicator itself reported.  The kernels run through the interpreter build (tests/hipemu), RCCL through
the shared-memory stand-in (tests/hipemu/fake_rccl.cpp)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHAPE = ["--n", "200", "--d", "4", "--m", "601", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--lean"]


def _run(extra, env_extra=None, timeout=600):
    sys.path.insert(0, os.path.join(HERE, "hipemu"))
    import build_emu
    env = dict(os.environ, ROBO_RCCL_LIB=build_emu.build_fake_rccl(), ROBO_BENCH_COMM_TIMEOUT="120", HIPEMU_DEVICES="4")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ROBO_BENCH_RENDEZVOUS", "ROBO_BENCH_FORCE_DIST"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--lib", build_emu.build()] + SHAPE + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def _line(res):
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout            # the contract: ONE JSON line on stdout
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def single():
    return _line(_run(["--gpus", "1"]))


def test_single_rank_line(single):
    assert single["n_gpus"] == 1 and single["ranks"] == [0] and single["comm_world"] == 1
    assert single["scaling"] == "weak" and single["config"]["candidates_total"] == 601
    assert "exchange" in single


@pytest.mark.parametrize("world", [2, 3])
def test_self_launch_strong(single, world):
    """SURVEY 8(d)'s strong-scaled headline: the SAME 601 candidates split over the ranks -> the single-rank argmax"""
    out = _line(_run(["--gpus", str(world), "--scaling", "strong"]))
    assert out["n_gpus"] == world and out["ranks"] == list(range(world)) and out["devices"] == list(range(world))
    assert out["comm_world"] == world and out["launcher"] == "spawn" and "librobo_hip" in out["exchange"]
    assert out["scaling"] == "strong" and out["config"]["candidates_total"] == 601
    assert out["config"]["candidates_per_gpu"] == -(-601 // world)          # rank 0 holds the larger shard
    assert out["argmax"] == single["argmax"]                                 # value AND global index, bit for bit
    # the other scaling of the same job rides along: weak = 601 candidates on every rank
    assert out["other_scaling"]["scaling"] == "weak" and out["other_scaling"]["candidates_total"] == 601 * world


def test_self_launch_weak():
    out = _line(_run(["--gpus", "2"]))
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["ranks"] == [0, 1]
    assert out["config"]["candidates_total"] == 1202
    assert out["other_scaling"]["scaling"] == "strong" and out["other_scaling"]["candidates_total"] == 601


def test_config4_sharded_information_gain_per_unit_cost():
    """bench.py --config c4: robo_ig_eval_per_cost_cand on one rank == its sharded form on two (same 300 candidates)"""
    shape = ["--config", "c4", "--n", "200", "--d", "4", "--m", "300"]
    one = _line(_run(["--gpus", "1"] + shape))
    two = _line(_run(["--gpus", "2", "--scaling", "strong"] + shape))
    assert two["ranks"] == [0, 1] and two["argmax"] == one["argmax"] and two["config"]["candidates_total"] == 300


def test_world_size_mismatch_is_an_error():
    """a launcher that started another number of ranks than --gpus asks for must not yield a line"""
    res = _run(["--gpus", "2"], env_extra={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and not res.stdout.strip()
    assert "WORLD_SIZE" in res.stderr


def test_forced_one_rank_communicator():
    """ROBO_BENCH_FORCE_DIST-style one-rank communicator through the self-launcher's rendezvous (no torch on CPU)"""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        out = _line(_run(["--gpus", "1"], env_extra={"ROBO_BENCH_RENDEZVOUS": d, "RANK": "0", "LOCAL_RANK": "0",
                                                      "WORLD_SIZE": "1"}))
    assert out["ranks"] == [0] and out["launcher"] == "spawn" and "librobo_hip" in out["exchange"]


def test_a_dead_rank_ends_the_job():
    """one rank cannot start (bad library path for it) -> the launcher stops the others and returns non-zero"""
    res = _run(["--gpus", "2"], env_extra={"ROBO_BENCH_COMM_TIMEOUT": "5", "ROBO_RCCL_LIB": "/nonexistent/librccl.so"},
               timeout=120)
    assert res.returncode != 0 and not res.stdout.strip()


# ---- --launcher inproc: ONE process, G emulated devices (robo_amd/csrc/multi.hip) -----------------------------------------
@pytest.mark.parametrize("world", [2, 3])
def test_inproc_strong(single, world):
    """`bench.py --gpus G --launcher inproc`: the same line shape as the one-process-per-GPU forms (ranks, devices,
    launcher) and, for the strong-scaled headline, the single-device argmax -- value and global index, bit for bit"""
    out = _line(_run(["--gpus", str(world), "--launcher", "inproc", "--scaling", "strong"],
                     env_extra={"HIPEMU_DEVICES": str(world)}))
    assert out["n_gpus"] == world and out["ranks"] == list(range(world)) and out["devices"] == list(range(world))
    assert out["launcher"] == "inproc" and out["comm_world"] == world and out["worker_threads"] == world
    assert out["scaling"] == "strong" and out["config"]["candidates_total"] == 601
    assert out["config"]["candidates_per_gpu"] == -(-601 // world)
    assert out["argmax"] == single["argmax"]
    assert out["other_scaling"]["scaling"] == "weak" and out["other_scaling"]["candidates_total"] == 601 * world
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data",
                "roofline", "gp_fit_ms", "exchange"):
        assert key in out, key


def test_inproc_eight_devices_dry_run(single):
    """the round-end scaling run's shape, `bench.py --gpus 8 --launcher inproc --scaling strong`, on eight emulated
    devices: one line, eight shards, and the strong-scaled argmax IS the one-device argmax (value and global index)"""
    out = _line(_run(["--gpus", "8", "--launcher", "inproc", "--scaling", "strong"], env_extra={"HIPEMU_DEVICES": "8"}))
    assert out["n_gpus"] == 8 and out["devices"] == list(range(8)) and out["worker_threads"] == 8
    assert out["config"]["candidates_total"] == 601 and out["config"]["candidates_per_gpu"] == 76
    assert out["argmax"] == single["argmax"]
    assert out["other_scaling"]["candidates_total"] == 601 * 8


def test_inproc_two_contexts_on_one_device(single):
    out = _line(_run(["--gpus", "2", "--launcher", "inproc", "--devices", "0,0", "--scaling", "strong"]))
    assert out["devices"] == [0, 0] and out["argmax"] == single["argmax"]


def test_inproc_sample_and_per_cost_shards():
    """config 3 (sample shard) and config 4 (information gain per unit cost) in one process == their one-device runs"""
    c3 = ["--config", "c3", "--n", "100", "--d", "3", "--m", "150", "--steps", "1", "--warmup", "0"]
    one = _line(_run(["--gpus", "1"] + c3))
    three = _line(_run(["--gpus", "3", "--launcher", "inproc"] + c3, env_extra={"HIPEMU_DEVICES": "3"}))
    assert three["config"]["parallelism"].startswith("sample-shard x3 (17/17/16)")
    assert three["argmax"][1] == one["argmax"][1] and abs(three["argmax"][0] - one["argmax"][0]) <= 1e-12 * abs(one["argmax"][0])
    c4 = ["--config", "c4", "--n", "150", "--d", "4", "--m", "200", "--scaling", "strong", "--steps", "1", "--warmup", "0"]
    one = _line(_run(["--gpus", "1"] + c4))
    two = _line(_run(["--gpus", "2", "--launcher", "inproc"] + c4, env_extra={"HIPEMU_DEVICES": "2"}))
    assert two["argmax"] == one["argmax"] and two["config"]["candidates_total"] == 200 and two["launcher"] == "inproc"


def test_inproc_refuses_a_multi_rank_launcher():
    res = _run(["--gpus", "2", "--launcher", "inproc"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and not res.stdout.strip()
