"""TEST INFRASTRUCTURE: one rank of tests/test_multi_gpu.py, started as a fresh child process (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in the environment, the way bench.py's own launcher and torch.distributed.run start ranks).

    python tests/multi_gpu_worker.py --backend nccl|gloo --chars N --steps K --dump DIR [--single-device]

Steps its contiguous shard of an N-character crowd with character-vs-character sweeps (configs[4]: the start-of-step snapshot
of ALL agents, Systems.swift:1592-1611, 1837-1841) through parallel.AgentExchange and dumps bodies / controllers / the gathered
snapshot of the last step."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--chars", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--dump", required=True)
    ap.add_argument("--single-device", action="store_true")
    ap.add_argument("--tag", default="run")
    args = ap.parse_args()

    import numpy as np
    import torch
    import bench
    import __graft_entry__

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=args.backend)
    sge = __graft_entry__.build()
    abi = sge.abi
    eng = sge.CharacterEngine(local)
    eng.set_option(abi.OPT_OVERLAP_SKIN, 1)
    ybot = sge.assets.YBotAssets()
    sge.crowd.upload_character_assets(eng, ybot, rings=4, segments=4)
    scene = sge.crowd.upload_asset_scene(eng, ("cheese",), footprint=240.0)  # 3.5 units between spawn points, capsules 3 wide: sweeps meet agents within a few steps
    first, count = sge.parallel.shard_range(args.chars, rank, world)
    eng.resize(count)
    bench._spawn_block(sge, eng, ybot, args.chars, first, count, scene, "ccd", agents=True)
    ex = sge.parallel.AgentExchange(eng, args.chars, rank, world, torch.device("cuda", local), dist)
    for _ in range(args.steps):
        ex.step(stages=abi.STAGE_ALL)
    eng.synchronize()
    out = eng.download(what=("bodies", "controllers"))
    gathered = ex.all.cpu().numpy() if getattr(ex, "all", None) is not None and not (ex.product and world == 1) else np.zeros((0, 8), np.float32)
    np.savez(os.path.join(args.dump, "%s_world%d_rank%d.npz" % (args.tag, world, rank)), first=first, count=count, gathered=gathered,
             overflow=int(eng.move_stats().overflow), self_check=str(ex.self_check), path=ex.describe()["path"], **out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
