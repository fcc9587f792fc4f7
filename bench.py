#!/usr/bin/env python3
"""Headline benchmark: characters/sec through the per-frame character update on MI355X.

    python bench.py --gpus N --steps K --warmup W            (any N: without a launcher's WORLD_SIZE in the environment this
                                                              process starts the N ranks itself, before touching the GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form for N > 1)

A "step" is one fixed step (dt = 1/60) of the whole hot path over one batch of synthetic
characters already resident in HBM: intent -> gravity -> capsule-CCD move-and-slide ->
locomotion -> action -> pose + palette -> 4-weight LBS of every vertex.
Workload (BASELINE.json configs[2]): 10,000 Y-Bot clones per GPU
(65 bones, 14,080 skinned vertices each) against the 17-Cheese triangle mesh (71,680 triangles,
FBX-derived, tests/golden/cheese_static.npz, used directly as the collision mesh:
`collisionMesh ?? mesh`, CollisionQuery.swift:344); characters shard by index across GPUs with
no data-path collective ("weak" scaling: per-GPU work fixed).
`--workload lbs` is configs[1] (pose + LBS only), `--workload mixed` configs[3]'s mixed-motion
crowd (default scene: the merged cheese + mirror + semla), `--workload agents` is configs[4]'s
character-vs-character exchange (one RCCL all-gather of capsule state per step).
`--scene` and `--mesh` choose the static scene and the skinned mesh independently.

Rank 0 prints ONE JSON line. `roofline` prices the LBS kernel (the HBM-bound kernel of the
path) from HIP events recorded on the stream it runs on; `cpu_baseline` times the CPU oracle
(a float32 port of the reference's Swift path, which cannot be built here) on a bounded
sample of the same crowd, on this box's host cores.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
SETTLE_STEPS = 120     # scene preparation: let the spawned crowd land and reach its walk/run speed


def _self_launch(gpus):
    """`python bench.py --gpus N` with N > 1 and no launcher: this process — which has not imported torch and never touches the GPU
    — starts N fresh rank processes of this same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's one JSON
    line, forwards every rank's stderr, and returns the worst child's exit code. A rank that fails ends the others' wait through
    the process-group timeout; nothing is re-executed in this process."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SGE_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    worst = max((abs(c) for c in codes), default=0)
    if line and worst == 0:
        print(line[-1])
    else:
        sys.stderr.write("bench.py: rank exit codes %s\n%s" % (codes, out0 or ""))
    return worst or (0 if line else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--chars", type=int, default=10000, help="characters per GPU")
    ap.add_argument("--workload", choices=["ccd", "lbs", "agents", "mixed"], default="ccd")
    ap.add_argument("--scene", choices=["cheese", "merged", "synthetic"], default=None,
                    help="static collision scene: cheese = 17-Cheese (71,680 tris + ground quad, the mesh configs[2] names; default), "
                         "merged = cheese + ornate mirror + semla (135,928 tris + ground quad; default of --workload mixed), "
                         "synthetic = a 71,680-triangle height field")
    ap.add_argument("--mesh", choices=["synthetic", "ybot"], default=None,
                    help="skinned mesh: synthetic = 14,080 vertices around the real Y-Bot skeleton (BASELINE's '~14k verts'; default), "
                         "ybot = the FBX-derived Y-Bot (35,440 welded vertices)")
    ap.add_argument("--assets", choices=["synthetic", "real"], default=None,
                    help="shorthand of round 1: real = --mesh ybot with the cheese / merged scene, synthetic = --mesh synthetic --scene synthetic")
    ap.add_argument("--layout", choices=["packed", "padded16"], default="packed")
    ap.add_argument("--overlap", dest="overlap", action="store_true", default=True,
                    help="(default) skin(n) runs on a second stream beside move(n+1) + pose(n+1): SGE_OPT_OVERLAP_SKIN")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false", help="one stream, stages strictly back to back")
    ap.add_argument("--refit", action="store_true",
                    help="also refit every character's acceleration structure after skinning (SURVEY 8 f2, the reference's next step; "
                         "not part of BASELINE.json's metric, so off by default); adds a `refit` object to the JSON line")
    ap.add_argument("--fuse", action="store_true", help="with --refit: SGE_OPT_FUSE_BLAS_REFIT, the LBS kernel reduces the boxes itself")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) in production; gloo only to rehearse N>1 on one GPU")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-sync", choices=["resident", "world"], default="resident",
                    help="resident (default, the headline): state stays in HBM for the whole timed region. world: the TIMED loop is the drop-in loop of "
                         "GPUCharacterStepSystem.fixedUpdate — pinned push of every character's intent, tick, asynchronous pull of bodies + controllers + "
                         "locomotion + actions, wait for it, copy into host arrays — every step (metric name says so; not the headline)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs outside the timed region that the default N=1 line carries: `world_sync` (the drop-in loop beside the resident "
                         "figure) and `real_mesh` (the same workload on the FBX-derived 35,440-vertex Y-Bot)")
    ap.add_argument("--dump-lbs-events", default=None, help="write the HIP-event duration of every skin launch of the timed region to this file")
    ap.add_argument("--cpu-sample-chars", type=int, default=2048)
    ap.add_argument("--cpu-sample-steps", type=int, default=16)
    args = ap.parse_args()
    if args.assets == "real":
        args.mesh = args.mesh or "ybot"
    elif args.assets == "synthetic":
        args.mesh, args.scene = args.mesh or "synthetic", args.scene or "synthetic"
    args.mesh = args.mesh or "synthetic"
    args.scene = args.scene or ("merged" if args.workload == "mixed" else "cheese")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_self_launch(args.gpus))  # (before torch is imported: the parent never initialises the GPU)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)

    import __graft_entry__
    sge = __graft_entry__.build()
    abi = sge.abi

    # ---- world: replicated per GPU ------------------------------------------------
    eng = sge.CharacterEngine(local_rank)
    eng.set_option(abi.OPT_SKIN_LAYOUT, abi.LAYOUT_PADDED16 if args.layout == "padded16" else abi.LAYOUT_PACKED)
    eng.set_option(abi.OPT_OVERLAP_SKIN, 1 if args.overlap else 0)
    ybot = sge.assets.YBotAssets()
    terrain = _build_world(sge, eng, ybot, args)
    n_total = args.chars * world
    first, count = sge.parallel.shard_range(n_total, rank, world)
    mode = "lbs" if args.workload == "lbs" else "ccd"
    # every rank draws the whole seeded crowd and keeps its contiguous block
    full_eng_state = None
    eng.resize(count)
    state = _spawn_block(sge, eng, ybot, n_total, first, count, terrain, mode, agents=args.workload == "agents",
                         mixed=args.workload == "mixed")
    stages = abi.STAGE_ALL if mode == "ccd" else (abi.STAGE_LOCOMOTION | abi.STAGE_ACTION | abi.STAGE_POSE | abi.STAGE_WRITEBACK | abi.STAGE_SKIN)
    if args.refit:
        eng.blas_build(eng.mesh["indices"])
        eng.set_option(abi.OPT_FUSE_BLAS_REFIT, 2 if args.fuse else 0)  # explicit either way: the refit kernel keeps its own line
        stages |= abi.STAGE_BLAS_REFIT
    exchange = None
    if args.workload == "agents":
        exchange = sge.parallel.AgentExchange(eng, n_total, rank, world, torch.device("cuda", local_rank), dist)

    world_loop = _WorldLoop(sge, eng, "immediate") if args.host_sync == "world" else None

    def step():
        if exchange is not None:
            exchange.step(stages=stages)
        elif world_loop is not None:
            world_loop.step(stages)
        else:
            eng.tick(stages=stages)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if mode == "ccd":
        for _ in range(SETTLE_STEPS):
            step()
    for _ in range(args.warmup):
        step()
    eng.synchronize()
    eng.set_option(abi.OPT_PROFILE, 1)
    eng.profile_read(reset=True)
    eng.move_stats(reset=True)
    if args.refit:
        eng.blas_profile(reset=True)

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if args.dump_lbs_events:
        ev = eng.skin_launch_times()
        with open(args.dump_lbs_events, "w") as f:
            f.write("# HIP-event duration (ms) of every skin launch of the timed region, in launch order: `python bench.py %s`\n" % " ".join(sys.argv[1:]))
            f.write("# launches %d  mean %.6f  min %.6f  max %.6f\n" % (len(ev), float(ev.mean()) if len(ev) else 0.0, float(ev.min()) if len(ev) else 0.0, float(ev.max()) if len(ev) else 0.0))
            f.writelines("%.6f\n" % x for x in ev)
    prof = eng.profile_read(reset=True)
    stats = eng.move_stats(reset=True)
    import ctypes as _C
    _q, _cpw = _C.c_int32(0), _C.c_int32(1)
    eng.t.lib.sge_debug_skin_form(eng.h, _C.byref(_q), _C.byref(_cpw))
    lbs_kernel = ("skin_ticket_multi_kernel (4-weight LBS: %.2f resident workgroups per CU, %d characters per loaded source vertex)" % (_q.value / 4.0, _cpw.value)
                  if _q.value > 0 and _cpw.value > 1 else
                  "skin_ticket_kernel (4-weight LBS: %.2f resident workgroups per CU)" % (_q.value / 4.0) if _q.value > 0 else "skin_kernel (4-weight LBS, one workgroup per character)")
    # the LBS kernel alone on the idle chip (outside the timed region): with --overlap its launches inside the timed region share
    # the SIMDs with the next step's collision kernels and stretch; this is the kernel's own rate
    # on an idle chip: a pause before every launch (back to back the kernel goes through a ~10-launch transient of 1.0-1.2 ms and
    # settles at 0.86 ms, tools/lbs_sustained.sh). Few launches, so that the rocprofv3 per-kernel average of the whole run stays
    # the timed region's.
    alone_launches = 8
    eng.synchronize()
    eng.set_option(abi.OPT_OVERLAP_SKIN, 0)  # by itself means: without the residency cap of the overlap schedule as well
    eng.profile_read(reset=True)
    for _ in range(alone_launches):
        eng.synchronize()
        time.sleep(0.001)
        eng.tick(dt=0.0, stages=abi.STAGE_SKIN)
    eng.synchronize()
    alone = eng.profile_read(reset=True)
    eng.set_option(abi.OPT_OVERLAP_SKIN, 1 if args.overlap else 0)
    eng.set_option(abi.OPT_PROFILE, 0)

    V, B = eng.vertex_count, eng.bone_count
    tris = int(eng.collision_counts()[1])
    scene_label = {"cheese": f"the 17-Cheese triangle mesh ({tris:,} triangles incl. the 2-triangle ground quad)",
                   "merged": f"the merged static scene: cheese + ornate mirror + semla ({tris:,} triangles incl. the ground quad)",
                   "synthetic": f"a synthetic height-field mesh ({tris:,} triangles)"}[args.scene]
    # ALGORITHMIC bytes of one LBS launch: per character 40*V written, 4160 B palette read,
    # source streams 64*V read once per launch (shared by all clones of the mesh)
    lbs_bytes = count * (40.0 * V + B * 64.0) + 64.0 * V
    lbs_ms = prof.skin_ms / max(prof.skin_launches, 1)
    achieved = lbs_bytes / (lbs_ms * 1e-3) / 1e9 if lbs_ms > 0 else 0.0
    alone_ms = alone.skin_ms / max(alone.skin_launches, 1)
    alone_gbs = lbs_bytes / (alone_ms * 1e-3) / 1e9 if alone_ms > 0 else 0.0
    traffic = _recorded_traffic(count, V, "skin_ticket_multi_kernel" if (_q.value > 0 and _cpw.value > 1) else "skin_ticket_kernel" if _q.value > 0 else "skin_kernel<")
    value = n_total * args.steps / elapsed
    out = {
        "metric": ("characters/sec (skin+CCD)" if mode == "ccd" else "characters/sec (pose+skin, no CCD)") + (
            " with the World synchronised every step (pinned push of intents, pull of bodies + controllers + locomotion + actions)" if world_loop is not None else ""),
        "value": value,
        "unit": "characters/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic crowd state; " + {"synthetic": "synthetic 14,080-vertex skinned mesh on the real Y-Bot skeleton and motion profiles",
                                              "ybot": "FBX-derived Y-Bot skinned mesh"}[args.mesh]
                + ("" if mode == "lbs" else "; " + {"cheese": "FBX-derived 17-Cheese static mesh", "merged": "FBX-derived cheese + ornate mirror + semla static meshes",
                                                    "synthetic": "synthetic height-field static mesh"}[args.scene]) + " (tests/golden/)",
        "config": {
            "workload": {"ccd": "configs[2]: 10k Y-Bot clones/GPU, pose + LBS + capsule-CCD vs " + scene_label,
                         "lbs": "configs[1]: 10k Y-Bot clones/GPU, Running profile, pose + LBS only (no collision)",
                         "agents": "configs[4]-style: character-vs-character sweeps (RCCL all-gather of capsule state) + pose + LBS + capsule-CCD vs " + scene_label,
                         "mixed": "configs[3]-style: mixed-motion Y-Bots (idle/walk/run/falling, 10% mid-blend) vs " + scene_label}[args.workload],
            "scene": args.scene if mode != "lbs" else None, "mesh": args.mesh,
            "scene_scale": terrain.get("scales") if mode != "lbs" else None,
            "characters_per_gpu": args.chars, "characters_total": n_total, "bones": B, "vertices_per_character": V,
            "static_triangles": tris, "dt": 1.0 / 60.0, "sharding": f"by-character x{world}",
            "skin_layout": args.layout, "overlap_skin_with_next_move": bool(args.overlap), "settle_steps": SETTLE_STEPS if mode == "ccd" else 0, "seed": 1234,
        },
        "roofline": {"bound": "hbm", "kernel": lbs_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "bytes_per_launch": lbs_bytes, "ms_per_launch": lbs_ms,
                     "note": "HIP events on the skin stream over the timed region" + (
                         "; the launches overlap the next step's collision + pose kernels (see lbs_alone for the kernel by itself)" if args.overlap and mode == "ccd" else "")},
        "lbs_alone": {"ms_per_launch": alone_ms, "achieved": alone_gbs, "frac": alone_gbs / HBM_PEAK_GBS, "unit": "GB/s", "launches": alone_launches,
                      "note": "the LBS launch of the serial schedule (skin_kernel, one workgroup per character) by itself after the timed region: no other kernel beside it, a 1 ms pause before every launch (back to back it sustains ~0.86 ms: profiles/r2_lbs_sustained_vs_burst.txt)"},
        "whole_path_hbm_frac": (value / world) * (40.0 * V + 64.0 * V / count + 2 * B * 64.0 + 640.0) / (HBM_PEAK_GBS * 1e9),
        "kernels_ms_per_step": {"move_ccd": prof.move_ms / args.steps, "pose": prof.pose_ms / args.steps,
                                "lbs": prof.skin_ms / args.steps, "agents_grid": prof.agents_ms / args.steps},
        "ccd": {"bvh_queries_per_char_step": stats.queries / max(count * args.steps, 1),
                "candidates_per_query": stats.candidates / max(stats.queries, 1),
                "distance_evals_per_query": stats.sweepIterations / max(stats.queries, 1),
                "traversal_steps_per_query": stats.traversalSteps / max(stats.queries, 1),
                "sweep_trips_per_query": stats.sweepTrips / max(stats.queries, 1),
                "pairs_skipped_by_exact_rejects_per_query": stats.prunedPairs / max(stats.queries, 1),
                "queries_per_s": stats.queries / max(prof.move_ms * 1e-3, 1e-9), "overflow": int(stats.overflow)},
    }
    if args.refit:
        # ALGORITHMIC bytes of one refit launch: the skinned positions read once + one box per entry written
        ms, launches = eng.blas_profile(reset=True)
        info = eng.blas_info
        stride = 16.0 if args.layout == "padded16" else 12.0
        refit_bytes = count * (stride * V + 24.0 * (info.entryCount + 1))
        refit_ms = ms / max(launches, 1)
        out["metric"] += " + acceleration-structure refit"
        out["kernels_ms_per_step"]["blas_refit"] = ms / args.steps
        if args.fuse:
            # one launch does both: its bytes are the LBS kernel's plus the boxes (the positions are not read back)
            fused_bytes = lbs_bytes + count * 24.0 * (info.entryCount + 1)
            out["roofline"]["kernel"] = "skin_refit_kernel (4-weight LBS + box reduction from LDS)"
            out["roofline"]["bytes_per_launch"] = fused_bytes
            out["roofline"]["achieved"] = fused_bytes / (lbs_ms * 1e-3) / 1e9
            out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS
            out["roofline"]["traffic"] = None
        out["refit"] = {"fused_into_lbs": True, "entries": int(info.entryCount), "clusters": int(info.clusterCount)} if args.fuse else {"kernel": "blas_refit_kernel", "bound": "hbm", "bytes_per_launch": refit_bytes, "ms_per_launch": refit_ms,
                        "achieved": refit_bytes / (refit_ms * 1e-3) / 1e9 if refit_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": (refit_bytes / (refit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if refit_ms > 0 else 0.0,
                        "triangles_per_character": int(info.triangleCount), "clusters": int(info.clusterCount),
                        "entries": int(info.entryCount), "wide_nodes": int(info.wideCount)}
    # what a SCALE record needs to be checked: the process group's size and backend, and every rank's output-stream placement
    place = eng.placement()
    ranks = [{"rank": rank, "device": local_rank, "placement_ms": round(place[0], 4), "placements_timed": place[1], "ms_per_step": elapsed / args.steps * 1e3}]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, ranks[0])
        ranks = gathered
    out["ranks"] = {"world_size": dist.get_world_size() if dist is not None else 1, "backend": (dist.get_backend() if dist is not None else None),
                    "self_launched": bool(os.environ.get("SGE_BENCH_SELF_LAUNCHED")), "per_rank": ranks}
    if exchange is not None:
        out["ranks"]["agent_exchange"] = exchange.describe()
    extras = rank == 0 and world == 1 and not args.no_extras and world_loop is None and exchange is None and not args.refit
    if extras:
        out["world_sync"] = _world_sync_leg(sge, eng, stages, count, elapsed / args.steps * 1e3)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = _cpu_baseline(sge, eng, ybot, terrain, stages & ~abi.STAGE_BLAS_REFIT, args, mode)  # the oracle has no refit
    eng.close()
    if extras and mode == "ccd" and args.mesh == "synthetic" and args.workload == "ccd":
        out["real_mesh"] = _real_mesh_leg(sge, args, local_rank, ybot)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


class _WorldLoop:
    """The drop-in loop: what host/swift/GPUCrowd.swift (C++ twin: host/sge_host.hpp GPUCrowd) does around sge_tick every fixed step when the
    engine's World keeps its component stores (Systems.swift:1802-1821 writeBack, World.swift:64-75): push every character's
    MoveIntent, tick, pull bodies + controllers + locomotion + actions and copy them into the host's arrays.
      immediate  pinned push, tick(n), asynchronous pull(n), wait for pull(n): the World holds step n when fixedUpdate returns
      lagged     the same, but the pull waited for is the previous step's: tick(n + 1) is enqueued before pull(n) is read
      pageable   sge_characters_upload / _download (pageable memory, host synchronisation of the whole context): round 3's binding"""

    def __init__(self, sge, eng, mode):
        import numpy as np
        self.np, self.eng, self.abi, self.mode = np, eng, sge.abi, mode
        self.intents = eng.download(what=("intents",))["intents"]
        A = sge.abi
        self.world = {"bodies": np.zeros(eng.count, A.body_dtype), "controllers": np.zeros(eng.count, A.controller_dtype),
                      "locomotion": np.zeros(eng.count, A.locomotion_dtype), "actions": np.zeros(eng.count, A.action_dtype)}
        self.prev = None
        self.host_s = {"push": 0.0, "tick": 0.0, "pull": 0.0, "wait": 0.0, "store": 0.0}  # where the HOST spends the step

    def _store(self, view):
        t0 = time.perf_counter()
        for k, v in view.items():
            # the decode into the World's stores: one memcpy per array here (as bytes: numpy copies structured records field by field)
            self.np.copyto(self.world[k].view(self.np.uint8), v.view(self.np.uint8))
        self.host_s["store"] += time.perf_counter() - t0

    def _wait(self, ticket):
        t0 = time.perf_counter()
        view = self.eng.state_wait(ticket)
        self.host_s["wait"] += time.perf_counter() - t0
        self._store(view)

    def step(self, stages):
        eng, A, H = self.eng, self.abi, self.host_s
        t0 = time.perf_counter()
        if self.mode == "pageable":
            eng.upload(intents=self.intents)
            t1 = time.perf_counter()
            eng.tick(stages=stages)
            t2 = time.perf_counter()
            d = eng.download(what=("bodies", "controllers", "locomotion", "actions"))
            t3 = time.perf_counter()
            H["push"] += t1 - t0; H["tick"] += t2 - t1; H["wait"] += t3 - t2
            self._store(d)
            return
        stg = eng.state_push_begin(A.STATE_INTENTS)
        self.np.copyto(stg["intents"].view(self.np.uint8), self.intents.view(self.np.uint8))
        eng.state_push_commit()
        t1 = time.perf_counter()
        eng.tick(stages=stages)
        t2 = time.perf_counter()
        t = eng.state_pull_async(A.STATE_WORLD)
        t3 = time.perf_counter()
        H["push"] += t1 - t0; H["tick"] += t2 - t1; H["pull"] += t3 - t2
        if self.mode == "immediate":
            self._wait(t)
        else:
            if self.prev is not None:
                self._wait(self.prev)
            self.prev = t

    def finish(self):
        if self.prev is not None:
            self._wait(self.prev)
            self.prev = None


def _world_sync_leg(sge, eng, stages, count, resident_ms, steps=100):
    """Outside the timed region: the same crowd stepped through the drop-in loop, three ways (see _WorldLoop)."""
    res = {"steps": steps, "bytes_per_character": {"push": 32, "pull": 352}, "resident_ms_per_step": resident_ms,
           "what": "push of every character's sge_move_intent + pull of bodies, controllers, locomotion and actions EVERY fixed step "
                   "(KinematicMoveStopSystem.writeBack / World.store traffic, Systems.swift:1802-1821, World.swift:64-75), copied into host arrays"}
    for mode in ("immediate", "lagged", "pageable"):
        loop = _WorldLoop(sge, eng, mode)
        for _ in range(10):
            loop.step(stages)
        loop.finish()
        eng.synchronize()
        for k in loop.host_s:
            loop.host_s[k] = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            loop.step(stages)
        loop.finish()
        eng.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        res[mode] = {"ms_per_step": ms, "characters_per_s": count / (ms * 1e-3), "frac_of_resident": resident_ms / ms,
                     "host_us_per_step": {k: round(v / steps * 1e6, 1) for k, v in loop.host_s.items()}}
    # the resident loop's host cost, for comparison: enqueueing one tick
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.tick(stages=stages)
    t1 = time.perf_counter()
    eng.synchronize()
    res["resident_host_enqueue_us_per_step"] = round((t1 - t0) / steps * 1e6, 1)
    res["note"] = ("immediate = pinned + event-ordered, the World holds step n when fixedUpdate returns (the host waits for move(n) + pose(n) + a 3.5 MB copy, "
                   "not for skin(n)); lagged = the World one step behind, nothing waited for; pageable = sge_characters_upload/_download (round 3)")
    return res


def _real_mesh_leg(sge, args, device, ybot, steps=20, warmup=20):
    """The default workload on the FBX-derived Y-Bot (35,440 welded vertices, real weights and bone indices) instead of the synthetic
    14,080-vertex mesh: a second context, the same scene, crowd, schedule and roofline arithmetic; outside the headline's timed region."""
    abi = sge.abi
    eng = sge.CharacterEngine(device)
    eng.set_option(abi.OPT_OVERLAP_SKIN, 1 if args.overlap else 0)
    ns = argparse.Namespace(**dict(vars(args), mesh="ybot"))
    terrain = _build_world(sge, eng, ybot, ns)
    n = args.chars
    eng.resize(n)
    _spawn_block(sge, eng, ybot, n, 0, n, terrain, "ccd", agents=False, mixed=False)
    for _ in range(SETTLE_STEPS + warmup):
        eng.tick()
    eng.synchronize()
    eng.set_option(abi.OPT_PROFILE, 1)
    eng.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.tick()
    eng.synchronize()
    dt = time.perf_counter() - t0
    prof = eng.profile_read(reset=True)
    V, B = eng.vertex_count, eng.bone_count
    lbs_bytes = n * (40.0 * V + B * 64.0) + 64.0 * V
    lbs_ms = prof.skin_ms / max(prof.skin_launches, 1)
    place = eng.placement()
    overflow = int(eng.move_stats().overflow)
    eng.close()
    return {"mesh": "FBX-derived Y-Bot (tests/golden/ybot_skinned.npz)", "vertices_per_character": V, "characters": n, "steps": steps, "warmup": warmup,
            "value": n * steps / dt, "unit": "characters/s", "ms_per_step": dt / steps * 1e3,
            "roofline": {"bound": "hbm", "bytes_per_launch": lbs_bytes, "ms_per_launch": lbs_ms, "achieved": lbs_bytes / (lbs_ms * 1e-3) / 1e9 if lbs_ms > 0 else 0.0,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (lbs_bytes / (lbs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if lbs_ms > 0 else 0.0},
            "whole_path_hbm_frac": (n * steps / dt) * (40.0 * V + 64.0 * V / n + 2 * B * 64.0 + 640.0) / (HBM_PEAK_GBS * 1e9),
            "placement_ms": round(place[0], 4), "overflow": overflow}


def _build_world(sge, e, ybot, args):
    """Character assets + static scene, identical for the GPU context and the CPU-baseline oracle."""
    if args.mesh == "ybot":
        sge.crowd.upload_ybot_mesh(e, ybot)      # Y Bot.fbx: 35,440 welded vertices, 65 bones, mesh inverse-bind re-bind
    else:
        sge.crowd.upload_character_assets(e, ybot)  # 22 x 10 vertices on 64 bones = 14,080
    if args.scene == "synthetic":
        return sge.crowd.upload_terrain(e)       # 224 x 160 x 2 = 71,680 triangles
    # the props are modelled for one character (the cheese is 3.7 units long): crowd.asset_scene_entities scales each to a
    # 200-unit footprint (the factor is reported as config.scene_scale) and lays them over the demo's ground quad
    which = ("cheese", "mirror", "semla") if args.scene == "merged" else ("cheese",)
    scene = sge.crowd.upload_asset_scene(e, which)  # 71,680 (+2 ground) or 135,928 (+2) triangles
    scene["scales"] = {b["name"]: round(b["scale"], 4) for b in scene["bounds"]}
    return scene


def _spawn_block(sge, eng, ybot, n_total, first, count, terrain, mode, agents, mixed=False):
    """Draw the seeded crowd of all ranks (host numpy, cheap) and upload this rank's block."""
    class _Capture:
        def resize(self, n):
            pass

        def upload(self, **kw):
            self.kw = kw

        def capsule_cast(self, q):  # spawn heights over an asset scene come from the collision world itself
            return eng.capsule_cast(q)

    cap = _Capture()
    sge.crowd.spawn_crowd(cap, ybot, n_total, terrain, seed=1234, mode=mode, agents=agents, mixed=mixed)
    block = {k: v[first:first + count] for k, v in cap.kw.items()}
    eng.upload(**block)
    return block


def skin_source_hash():
    """Identifies the LBS kernel a PMC record was taken from (sge_skin.hip + the layouts it includes)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("sge_skin.hip", "sge_internal.hpp", "sge_blas_dev.hpp"):
        h.update(open(os.path.join(ROOT, "swift-game-engine_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def _recorded_traffic(count, V, kernel):
    """HBM bytes per LBS launch from the committed rocprofv3 --pmc passes (tools/collect_r3.sh writes the record).
    Only a record taken from THIS kernel source, this shape and the LBS kernel form that ran counts; anything else reports null."""
    path = os.path.join(ROOT, "profiles", "lbs_traffic.json")
    try:
        rec = json.load(open(path))
        if rec.get("characters") == count and rec.get("vertices") == V and rec.get("skin_source_hash") == skin_source_hash():
            for name, b in rec.get("hbm_bytes_per_launch_by_kernel", {}).items():
                if kernel in name:
                    return b
    except Exception:
        pass
    return None


def _cpu_baseline(sge, eng, ybot, terrain, stages, args, mode):
    """The oracle (a CPU port of the reference path) on a bounded sample of the SAME settled crowd."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    n = min(args.cpu_sample_chars, eng.count)
    steps = args.cpu_sample_steps
    state = eng.download(0, n)
    cores = min(os.cpu_count() or 1, 16)  # the GPU box's CPU share for one GPU
    res = {}
    for label, threads in (("single_thread", 1), ("all_cores", cores)):
        cpu = ob.oracle_engine()
        _build_world(sge, cpu, ybot, args)
        cpu.resize(n)
        cpu.upload(**state)
        t0 = time.perf_counter()
        for _ in range(steps):
            ob.tick_mt(cpu, threads, stages=stages)
        dt = time.perf_counter() - t0
        res[label] = n * steps / dt
        cpu.close()
    return {"value": res["all_cores"], "unit": "characters/s", "cores": cores, "kind": "port",
            "single_thread_value": res["single_thread"],
            "sample": f"{n} characters of the settled crowd x {steps} fixed steps, same stages, CPU oracle "
                      f"(oracle/, float32 C++ port of the Swift path; the Swift reference has no toolchain here)"}


if __name__ == "__main__":
    main()
