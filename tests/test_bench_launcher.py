"""bench.py --gpus N without a launcher starts N ranks itself.  Rehearsed here on the CPU with the
tests-only emulator library passed explicitly (HHE_LIB): the JSON line is marked with that backend and its throughput is
meaningless -- what is checked is the rank plumbing: two ranks, gloo barrier, MAX time / SUM units."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_without_env_spawns_two_ranks(emu_lib):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HHE_LIB=os.path.join(ROOT, "tests", "emu", "libhhe_emu.so"), OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--params", "tiny", "--batch", "2",
                        "--steps", "1", "--warmup", "0", "--cpu-baseline", "0", "--extras", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 prints the ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["backend"].startswith("cpu-emulator")
    units = d["value"] * d["ms_per_step"] * 1e-3 * d["steps"]
    assert abs(units - 2 * 2) < 1e-6  # batch 2 per rank x 2 ranks: the whole-job aggregate
    assert d["scaling"] == "weak" and d["config"]["sharding"].startswith("2 rank(s)")


def test_gpus_2_fails_loudly_when_the_devices_are_missing():
    import torch
    if torch.cuda.device_count() >= 2:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HHE_LIB", "HHE_BENCH_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "GPU(s) are visible" in p.stderr and not p.stdout.strip()


def test_config5_primes_are_seals_create(orc):
    sys.path.insert(0, ROOT)
    import bench
    assert bench.Q_CONFIG5 == orc.coeff_modulus_create(65536, [60] * 6)
    assert bench.Q_CONFIG2 == orc.coeff_modulus_create(32768, [60] * 4)


def test_mnist_e2e_gpus_2_shards_samples_over_ranks(emu_lib):
    """BASELINE config 3 as a bench workload: `--workload mnist-e2e --gpus 2` -- every rank runs hhe_decompose + the FC rows on its
    contiguous sample range (sharding.shard_samples), rank 0 prints samples/s = SUM of samples / MAX of time.  Rehearsed on the
    emulator at N = 1024 with 300-word records (a 784-word record needs N >= 2048)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HHE_LIB=os.path.join(ROOT, "tests", "emu", "libhhe_emu.so"), OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "mnist-e2e", "--params", "tiny", "--batch", "1",
                        "--record-words", "300", "--neurons", "2", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["unit"] == "samples/s" and d["scaling"] == "weak"
    samples = d["value"] * d["ms_per_step"] * 1e-3 * d["steps"]
    assert abs(samples - 2) < 1e-6          # 1 sample per rank x 2 ranks
    assert d["rank0"]["samples"] == 1 and d["roofline"]["key_switches_per_row_executed"] < d["roofline"]["reference_op_count"]["key_switches_per_row"]
    assert d["roofline"]["decompose"]["frac"] > 0 and d["config"]["sharding"].startswith("2 rank(s), contiguous sample ranges")
