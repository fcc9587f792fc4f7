"""Two-rank parity of the HIP product on the ONE GPU a gpurun box has (configs[4]'s exchange, Systems.swift:1592-1611: every
character's sweep sees the start-of-step snapshot of ALL agents).

  python tools/two_rank_parity.py [--chars-per-rank 3000] [--steps 90] [--out profiles/r3_two_rank_parity.json]

The parent (this process, which never touches the GPU) starts
  1. two fresh rank processes through torch.distributed.run (gloo rendezvous on 127.0.0.1; both ranks use cuda:0; the exchange is
     AgentExchange's staged path: export on the GPU -> gloo all-gather through host memory -> import on the GPU), each stepping its
     contiguous shard of a 2 x chars-per-rank crowd with character-vs-character sweeps and dumping its bodies / controllers;
  2. one single-process run of the same 2 x chars-per-rank crowd (world size 1: sge_agents_allgather's own buffers),
and compares the dumps bit for bit. The multi-rank exchange over RCCL itself cannot run here (one GPU per box); what this pins is
everything around the transport: sharding, padding slots, self offsets, snapshot timing, the import path and the grid build over a
gathered snapshot that holds other ranks' agents."""
import argparse, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_role(args):
    import numpy as np
    import torch
    import bench
    import __graft_entry__
    sge = __graft_entry__.build()
    abi = sge.abi
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
    torch.cuda.set_device(0)
    eng = sge.CharacterEngine(0)
    eng.set_option(abi.OPT_OVERLAP_SKIN, 1)
    ybot = sge.assets.YBotAssets()
    ns = argparse.Namespace(mesh="synthetic", scene=args.scene)
    terrain = bench._build_world(sge, eng, ybot, ns)
    n_total = 2 * args.chars_per_rank
    first, count = sge.parallel.shard_range(n_total, rank, world)
    eng.resize(count)
    bench._spawn_block(sge, eng, ybot, n_total, first, count, terrain, "ccd", agents=True)
    ex = sge.parallel.AgentExchange(eng, n_total, rank, world, torch.device("cuda", 0), dist)
    for _ in range(args.steps):
        ex.step(stages=abi.STAGE_ALL)
    eng.synchronize()
    out = eng.download()
    pal = eng.palettes(0, min(count, 32))[0]
    stats = eng.move_stats()
    extra = {}
    if world == 1:  # the characters the two ranks dump as their first 32: global indices 0.. and chars_per_rank..
        for f in (0, args.chars_per_rank):
            extra["palettes_at_%d" % f] = eng.palettes(f, 32)[0]
    np.savez(os.path.join(args.dump, "world%d_rank%d.npz" % (world, rank)), first=first, count=count, palettes=pal,
             overflow=int(stats.overflow), **extra, **{k: out[k] for k in ("bodies", "controllers", "locomotion")})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--role", default="parent")
    ap.add_argument("--chars-per-rank", type=int, default=3000)
    ap.add_argument("--steps", type=int, default=90)
    ap.add_argument("--scene", default="cheese")
    ap.add_argument("--dump", default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r3_two_rank_parity.json"))
    ap.add_argument("--port", type=int, default=29731)
    args = ap.parse_args()
    if args.role != "parent":
        return run_role(args)
    import numpy as np
    dump = tempfile.mkdtemp(prefix="sge_two_rank_")
    common = ["--role", "rank", "--chars-per-rank", str(args.chars_per_rank), "--steps", str(args.steps), "--scene", args.scene, "--dump", dump]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(args.port), os.path.abspath(__file__)] + common, env=env, capture_output=True, text=True)
    if two.returncode != 0:
        print(two.stdout[-2000:], two.stderr[-4000:])
        raise SystemExit("the two-rank run failed")
    one = subprocess.run([sys.executable, os.path.abspath(__file__)] + common, env=dict(env, RANK="0", WORLD_SIZE="1"), capture_output=True, text=True)
    if one.returncode != 0:
        print(one.stdout[-2000:], one.stderr[-4000:])
        raise SystemExit("the single-process run failed")
    single = np.load(os.path.join(dump, "world1_rank0.npz"))
    ranks = [np.load(os.path.join(dump, "world2_rank%d.npz" % r)) for r in (0, 1)]
    result = {"what": "2 ranks sharing one MI355X (gloo-staged AgentExchange, fresh processes) vs one process, same 2 x %d crowd, %d steps, "
                      "overlap schedule, character-vs-character sweeps on, scene %s" % (args.chars_per_rank, args.steps, args.scene),
              "chars_total": 2 * args.chars_per_rank, "steps": args.steps, "arrays": {}}
    ok = True
    for k in ("bodies", "controllers", "locomotion"):
        cat = np.concatenate([r[k] for r in ranks])
        same = cat.tobytes() == single[k].tobytes()
        result["arrays"][k] = {"bytes": int(cat.nbytes), "identical": bool(same)}
        ok &= same
    # palettes: every rank dumps its first 32 characters, the single process dumps the characters at the same GLOBAL indices
    for r, d in enumerate(ranks):
        f = int(d["first"])
        same = d["palettes"].tobytes() == single["palettes_at_%d" % f].tobytes()
        result["arrays"]["palettes_rank%d" % r] = {"first_global_index": f, "identical": bool(same)}
        ok &= same
    # did the exchange matter? (a run in which no sweep ever met another rank's agent would pass trivially)
    # ... a run WITHOUT the exchange must differ, or no sweep ever met the other rank's agents
    moved = np.abs(single["bodies"]["position"][:, [0, 2]]).sum()
    both = np.concatenate([r["bodies"] for r in ranks])
    half = args.chars_per_rank
    pa, pb = both["position"][:half][:, [0, 2]], both["position"][half:][:, [0, 2]]
    step = max(1, half // 512)
    near = float(np.min(np.linalg.norm(pa[::step, None, :] - pb[None, ::step, :], axis=2)))
    result["closest_cross_rank_pair_of_a_sample"] = near
    if near > 6.0:
        raise SystemExit("no character of rank 0 ended near a character of rank 1 (closest sampled pair %.2f units): the crowd does not exercise the exchange" % near)
    result["overflow"] = [int(single["overflow"])] + [int(r["overflow"]) for r in ranks]
    result["identical"] = bool(ok)
    result["position_checksum"] = float(moved)
    json.dump(result, open(args.out, "w"), indent=1)
    print(json.dumps(result))
    if not ok:
        raise SystemExit("two-rank result differs from the single-process result")


if __name__ == "__main__":
    main()
