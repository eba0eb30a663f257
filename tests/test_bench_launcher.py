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
