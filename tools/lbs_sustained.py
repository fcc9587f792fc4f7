"""Diagnostic (GPU box, under rocprofv3 --kernel-trace): the LBS kernel launched 100 times back to back, then 30 times with a pause
of ~1 ms before each launch. The kernel trace gives the duration of every launch in order (tools/lbs_sustained.sh prints them)."""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import __graft_entry__  # noqa: E402

torch.cuda.set_device(0)
sge = __graft_entry__.build()
abi = sge.abi
eng = sge.CharacterEngine(0)
ybot = sge.assets.YBotAssets()
terrain = bench._build_world(sge, eng, ybot, types.SimpleNamespace(mesh="synthetic", scene="cheese"))
eng.resize(10000)
bench._spawn_block(sge, eng, ybot, 10000, 0, 10000, terrain, "lbs", agents=False)
st = abi.STAGE_LOCOMOTION | abi.STAGE_ACTION | abi.STAGE_POSE | abi.STAGE_WRITEBACK | abi.STAGE_SKIN
for _ in range(5):
    eng.tick(stages=st)
eng.synchronize()
time.sleep(0.05)
for _ in range(100):
    eng.tick(dt=0.0, stages=abi.STAGE_SKIN)
eng.synchronize()
for _ in range(30):
    time.sleep(0.001)
    eng.tick(dt=0.0, stages=abi.STAGE_SKIN)
    eng.synchronize()
eng.close()
