#!/usr/bin/env python3
"""Headline benchmark: characters/sec through the per-frame character update on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

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
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
SETTLE_STEPS = 120     # scene preparation: let the spawned crowd land and reach its walk/run speed


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
    ap.add_argument("--cpu-sample-chars", type=int, default=2048)
    ap.add_argument("--cpu-sample-steps", type=int, default=16)
    args = ap.parse_args()
    if args.assets == "real":
        args.mesh = args.mesh or "ybot"
    elif args.assets == "synthetic":
        args.mesh, args.scene = args.mesh or "synthetic", args.scene or "synthetic"
    args.mesh = args.mesh or "synthetic"
    args.scene = args.scene or ("merged" if args.workload == "mixed" else "cheese")

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
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

    def step():
        if exchange is not None:
            exchange.step(stages=stages)
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
        "metric": "characters/sec (skin+CCD)" if mode == "ccd" else "characters/sec (pose+skin, no CCD)",
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
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = _cpu_baseline(sge, eng, ybot, terrain, stages & ~abi.STAGE_BLAS_REFIT, args, mode)  # the oracle has no refit
    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


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
