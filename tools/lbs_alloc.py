"""Diagnostic: the same skin kernel on several contexts of one process — does its speed depend on where the output
buffers landed?  Prints each context's output pointers and LBS time."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sge = importlib.import_module("swift-game-engine_amd")
abi = sge.abi
ybot = sge.assets.YBotAssets()
n = 10000
st = abi.STAGE_LOCOMOTION | abi.STAGE_ACTION | abi.STAGE_POSE | abi.STAGE_WRITEBACK | abi.STAGE_SKIN
engines = []
probes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    eng = sge.CharacterEngine(0)
    eng.set_option(abi.OPT_PLACEMENT_PROBES, probes)
    sge.crowd.upload_character_assets(eng, ybot)
    sge.crowd.spawn_crowd(eng, ybot, n, None, mode="lbs")
    for _ in range(5):
        eng.tick(stages=st)
    eng.synchronize()
    eng.set_option(abi.OPT_PROFILE, 1)
    ptrs = [C.c_void_p() for _ in range(4)]
    eng.t.lib.sge_crowd_buffers(eng.h, *[C.byref(p) for p in ptrs])
    engines.append((eng, [p.value for p in ptrs]))
for rnd in range(3):
    for k, (eng, ptrs) in enumerate(engines):
        eng.profile_read(reset=True)
        for _ in range(30):
            eng.tick(stages=st)
        eng.synchronize()
        p = eng.profile_read(reset=True)
        print("round %d context %d: skin %.4f ms  pal %x pos %x nrm %x tan %x  (pos %% 2MiB = %x)" % (
            rnd, k, p.skin_ms / p.skin_launches, ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[1] % (2 << 20)), flush=True)
