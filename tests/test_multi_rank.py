"""CPU tests of the N>1 path: two gloo ranks shard the crowd by character and exchange the
AgentSweepState snapshot (config 5). The compute backend here is the CPU oracle (no GPU in this
container); the host code under test — sharding, padding, all-gather, import — is the same
module the HIP product uses with RCCL."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TOTAL = 45
STEPS = 25


def _setup(sge, ob, n_total, first, count):
    from scenes import build_scene

    class Cap:
        def resize(self, n):
            pass

        def upload(self, **kw):
            self.kw = kw

    eng = ob.oracle_engine()
    ybot, terrain, _ = build_scene(sge, eng, 1, terrain_cells=(24, 18), rings=3, segments=3)
    cap = Cap()
    sge.crowd.spawn_crowd(cap, ybot, n_total, terrain, seed=77, mode="ccd", agents=True)
    # crowd the characters together so that capsule-capsule sweeps actually hit
    cap.kw["bodies"]["position"][:, 0] *= 0.25
    cap.kw["bodies"]["position"][:, 2] *= 0.25
    eng.resize(count)
    eng.upload(**{k: v[first:first + count] for k, v in cap.kw.items()})
    return eng


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sge = importlib.import_module("swift-game-engine_amd")
    import oracle_binding as ob

    first, count = sge.parallel.shard_range(N_TOTAL, rank, world)
    eng = _setup(sge, ob, N_TOTAL, first, count)
    ex = sge.parallel.AgentExchange(eng, N_TOTAL, rank, world, torch.device("cpu"), dist)
    for _ in range(STEPS):
        ex.step(stages=sge.abi.STAGE_ALL_FIXED)
    d = eng.download(what=("bodies", "controllers"))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), bodies=d["bodies"], controllers=d["controllers"], first=first)
    dist.destroy_process_group()


def test_two_ranks_match_single_process(sge, tmp_path):
    import oracle_binding as ob
    from scenes import assert_struct_equal

    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # single process reference: the whole crowd in one world, same exchange code with world = 1
    eng = _setup(sge, ob, N_TOTAL, 0, N_TOTAL)
    ex = sge.parallel.AgentExchange(eng, N_TOTAL, 0, 1, torch.device("cpu"), None)
    for _ in range(STEPS):
        ex.step(stages=sge.abi.STAGE_ALL_FIXED)
    ref = eng.download(what=("bodies", "controllers"))
    # and the oracle's own built-in snapshot path (no import) agrees with the exchange path
    eng2 = _setup(sge, ob, N_TOTAL, 0, N_TOTAL)
    for _ in range(STEPS):
        eng2.tick(stages=sge.abi.STAGE_ALL_FIXED | sge.abi.STAGE_AGENTS)
    ref2 = eng2.download(what=("bodies", "controllers"))
    assert_struct_equal(ref["bodies"], ref2["bodies"], "bodies(single vs builtin)")
    got_b, got_c = [], []
    for r in range(2):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        got_b.append(z["bodies"])
        got_c.append(z["controllers"])
    assert_struct_equal(np.concatenate(got_b), ref["bodies"], "bodies")
    assert_struct_equal(np.concatenate(got_c), ref["controllers"], "controllers")
    # the agents really interacted: the result differs from a run without character-vs-character sweeps
    eng3 = _setup(sge, ob, N_TOTAL, 0, N_TOTAL)
    for _ in range(STEPS):
        eng3.tick(stages=sge.abi.STAGE_ALL_FIXED)
    free = eng3.download(what=("bodies",))["bodies"]
    assert np.abs(free["position"] - ref["bodies"]["position"]).max() > 1e-3


def test_bench_launcher_starts_ranks_itself_and_reports_their_failure():
    """`python bench.py --gpus 2` with no launcher in front of it (the shape of the driver's N=1 command with another N): the parent
    starts two fresh rank processes before it imports torch. There is no GPU in this container, so every rank must end with the
    product's "no CPU fallback" message and the parent must return non-zero without a JSON line — and without hanging on a
    rendezvous. (On a GPU box: tests/test_multi_gpu.py.)"""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by tests/test_multi_gpu.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert out.stderr.count("there is no CPU fallback") == 2, out.stderr[-2000:]
    assert "rank exit codes" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # a launcher's world size that disagrees with --gpus is refused by the rank itself
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="4", RANK="0"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "must agree" in out.stderr
