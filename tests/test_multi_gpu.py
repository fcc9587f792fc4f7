"""Two MI355X, two processes, RCCL: the character-vs-character exchange of configs[4] (one all-gather of the 32-byte AgentSweepState
per step, Systems.swift:1592-1611, 1837-1841) on real hardware. Skipped on a one-GPU box (the gpurun pool); collected everywhere.

  - AgentExchange's stream-ordered branch (export kernel -> all_gather_into_tensor -> import on the engine's stream, no host
    synchronisation) against the same exchange with host synchronisation around the collective (SGE_EXCHANGE_SYNC=1), against the
    staged gloo branch, and against ONE process holding the whole crowd: bodies and controllers byte-identical;
  - sge_agents_allgather with a real two-rank ncclComm_t, the way a Swift / C++ host drives it (tests/cpp/allgather_two_rank.cpp):
    two processes, the ncclUniqueId travels through a file, each rank's result against the single-process run of the same crowd;
  - bench.py --gpus 2 starts its own ranks (no launcher) and prints one JSON line for world size 2.
Ranks are started as fresh child processes, never by re-executing this process."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from scenes import assert_struct_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "swift-game-engine_amd")
WORKER = os.path.join(ROOT, "tests", "multi_gpu_worker.py")
pytestmark = pytest.mark.gpu


def _gpus():
    import torch
    return torch.cuda.device_count()  # (does not initialise the GPU on this image)


needs_two = pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs (one process per GPU over RCCL)")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(cmd, world, extra_env=None, timeout=900):
    """Start `world` fresh rank processes of `cmd`; returns when all have ended. Any non-zero exit fails the test with the output."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d exited with %d:\n%s" % (r, p.returncode, outs[r][-4000:])
    return outs


def _load(dump, tag, world):
    return [np.load(os.path.join(dump, "%s_world%d_rank%d.npz" % (tag, world, r))) for r in range(world)]


@needs_two
def test_two_ranks_on_rccl_match_one_process(tmp_path):
    dump = str(tmp_path)
    chars, steps = 4096, 60
    base = [sys.executable, WORKER, "--chars", str(chars), "--steps", str(steps), "--dump", dump]
    launch(base + ["--backend", "nccl", "--tag", "stream"], 2)
    launch(base + ["--backend", "nccl", "--tag", "synced"], 2, {"SGE_EXCHANGE_SYNC": "1"})
    launch(base + ["--backend", "gloo", "--tag", "staged"], 2)
    launch(base + ["--tag", "one"], 1)
    one = _load(dump, "one", 1)[0]
    assert int(one["overflow"]) == 0
    for tag in ("stream", "synced", "staged"):
        ranks = _load(dump, tag, 2)
        assert [int(r["first"]) for r in ranks] == [0, chars // 2]
        for k in ("bodies", "controllers"):
            assert_struct_equal(np.concatenate([r[k] for r in ranks]), one[k], "%s (%s exchange, 2 ranks vs 1 process)" % (k, tag), skip=())
        assert all(int(r["overflow"]) == 0 for r in ranks)
        # both ranks assembled the same snapshot, and each rank's own records sit at its slot
        assert np.array_equal(ranks[0]["gathered"], ranks[1]["gathered"]), tag
    stream = _load(dump, "stream", 2)
    assert all(str(r["self_check"]) == "passed" for r in stream)          # the first exchange was repeated with host synchronisation
    assert "stream-ordered" in str(stream[0]["path"]) and "staged" in str(_load(dump, "staged", 2)[0]["path"])
    # the exchange mattered: some sweep met an agent of the OTHER rank — without the gather the result is another one
    g = stream[0]["gathered"]
    slot = g.shape[0] // 2
    solid = g[:, 3] >= 0
    assert solid[:slot].any() and solid[slot:].any()
    d = np.linalg.norm(g[:slot, None, [0, 2]] - g[None, slot:, [0, 2]], axis=2)[np.ix_(g[:slot, 3] >= 0, g[slot:, 3] >= 0)]
    assert d.min() < 6.0, "no agent of rank 0 ever came near an agent of rank 1: the test crowd does not exercise the exchange"


def test_two_rank_worker_rehearsal_on_one_gpu(tmp_path):
    """What a one-GPU box can run of the test above: the same worker, two fresh rank processes sharing cuda:0 (--single-device),
    the exchange staged through gloo, against one process with the whole crowd — everything but the RCCL transport."""
    dump = str(tmp_path)
    chars, steps = 2048, 40
    base = [sys.executable, WORKER, "--chars", str(chars), "--steps", str(steps), "--dump", dump, "--single-device"]
    launch(base + ["--backend", "gloo", "--tag", "staged"], 2)
    launch(base + ["--tag", "one"], 1)
    one = _load(dump, "one", 1)[0]
    ranks = _load(dump, "staged", 2)
    assert [int(r["first"]) for r in ranks] == [0, chars // 2] and int(one["overflow"]) == 0
    for k in ("bodies", "controllers"):
        assert_struct_equal(np.concatenate([r[k] for r in ranks]), one[k], "%s (staged exchange, 2 ranks on one GPU vs 1 process)" % k, skip=())
    assert np.array_equal(ranks[0]["gathered"], ranks[1]["gathered"]) and "staged" in str(ranks[0]["path"])
    g = ranks[0]["gathered"]
    slot = g.shape[0] // 2
    d = np.linalg.norm(g[:slot, None, [0, 2]] - g[None, slot:, [0, 2]], axis=2)[np.ix_(g[:slot, 3] >= 0, g[slot:, 3] >= 0)]
    assert d.min() < 6.0, "no agent of rank 0 came near an agent of rank 1"


def _build_cpp(tmp_path, name):
    exe = str(tmp_path / name)
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    subprocess.check_call([hipcc, "-std=c++17", os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-I" + os.path.join(ROOT, "include"),
                           "-L" + PKG, "-lsge_amd", "-lrccl", "-Wl,-rpath," + PKG, "-o", exe])
    return exe


def test_two_rank_allgather_sample_on_one_gpu(tmp_path):
    """The same program with one rank (world size 1: no communicator needed) — what a one-GPU box can run of it."""
    exe = _build_cpp(tmp_path, "allgather_two_rank")
    out = subprocess.run([exe, "0", "1", str(tmp_path / "id"), str(tmp_path / "one.bin")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "allgather rank 0/1 ok" in out.stdout, out.stdout + out.stderr
    assert os.path.getsize(tmp_path / "one.bin") == 64 * 96


@needs_two
def test_sge_agents_allgather_with_a_two_rank_communicator(tmp_path):
    exe = _build_cpp(tmp_path, "allgather_two_rank")
    one = subprocess.run([exe, "0", "1", str(tmp_path / "id1"), str(tmp_path / "one.bin")], capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stdout + one.stderr
    procs = [subprocess.Popen([exe, str(r), "2", str(tmp_path / "id2"), str(tmp_path / ("rank%d.bin" % r))], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0 and ("allgather rank %d/2 ok" % r) in outs[r], outs[r]
    whole = open(tmp_path / "one.bin", "rb").read()
    halves = open(tmp_path / "rank0.bin", "rb").read() + open(tmp_path / "rank1.bin", "rb").read()
    assert halves == whole, "two ranks over RCCL and one process disagree on the bodies after 180 steps"


def test_bench_starts_its_own_ranks_on_one_gpu():
    """`python bench.py --gpus 2` without a launcher: the parent starts two fresh ranks itself. On a one-GPU box both ranks share
    cuda:0 and the exchange is staged through gloo (--single-device --dist-backend gloo): the launcher, the rendezvous, the
    sharding and the one JSON line are the production ones."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--dist-backend", "gloo", "--workload", "agents",
           "--chars", "1500", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks"]["world_size"] == 2 and rec["ranks"]["self_launched"] and len(rec["ranks"]["per_rank"]) == 2
    assert rec["config"]["characters_total"] == 3000 and rec["value"] > 0 and "staged" in rec["ranks"]["agent_exchange"]["path"]


@needs_two
def test_bench_starts_its_own_ranks_on_rccl():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "agents", "--chars", "4000", "--steps", "10", "--warmup", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["ranks"]["backend"] == "nccl" and rec["ranks"]["agent_exchange"]["self_check"] == "passed"
    assert sorted(r["device"] for r in rec["ranks"]["per_rank"]) == [0, 1]
