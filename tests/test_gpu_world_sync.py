"""GPU tests of the drop-in boundary's per-step World synchronisation and of the stream ordering around it:
  - sge_state_pull_async / sge_state_wait / sge_state_push_* (the pinned, event-ordered form of GPUCrowd.pullBack / pushDirtyState,
    Systems.swift:1802-1821, World.swift:64-75) against the synchronous sge_characters_download / _upload, under the three-stream
    overlap schedule;
  - SGE_OPT_OVERLAP_SKIN = 1 on a caller-provided stream keeps ABI version 1's contract (advisor finding, round 3);
  - back-to-back MOVE ticks whose scheduling lists cannot be reused (another range, another threshold) join the list build that
    the previous tick left on the second stream (advisor finding, round 3)."""
import ctypes as C

import numpy as np
import pytest

from scenes import assert_struct_equal

pytestmark = pytest.mark.gpu
KEYS = ("bodies", "controllers", "locomotion", "actions")


def _crowd(sge, n, seed, overlap, rings=6, segments=6, footprint=120.0):
    A = sge.abi
    ybot = sge.assets.YBotAssets()
    eng = sge.CharacterEngine(0)
    eng.set_option(A.OPT_OVERLAP_SKIN, overlap)
    sge.crowd.upload_character_assets(eng, ybot, rings=rings, segments=segments)
    scene = sge.crowd.upload_asset_scene(eng, ("cheese",), footprint=footprint)
    state = sge.crowd.spawn_crowd(eng, ybot, n, scene, seed=seed, mixed=True)
    return eng, state


def test_async_pull_is_the_synchronous_download(sge):
    """The pull of step n, read after sge_state_wait, is byte for byte what sge_characters_download returns after step n on a
    context fed the same calls — under the overlap schedule (skin(n) and pose(n) beside move(n+1)), with the NEXT tick already
    enqueued before the pull is waited for (the snapshot is taken on the device behind step n's kernels, so step n+1 cannot
    show through), with no host synchronisation anywhere on the asynchronous side."""
    A = sge.abi
    n, steps = 3000, 24
    ref, _ = _crowd(sge, n, 41, overlap=1)
    expect = []
    for _ in range(steps):
        ref.tick()
        expect.append({k: v.copy() for k, v in ref.download(what=KEYS).items()})
    ref.close()

    eng, _ = _crowd(sge, n, 41, overlap=1)
    got, prev = [], None
    for s in range(steps):
        eng.tick()
        t = eng.state_pull_async(A.STATE_WORLD)
        if prev is not None:                      # step s is enqueued; now read step s - 1
            got.append({k: v.copy() for k, v in eng.state_wait(prev).items()})
        prev = t
    got.append({k: v.copy() for k, v in eng.state_wait(prev).items()})
    assert eng.state_ready(prev)
    for s in range(steps):
        assert set(got[s]) == set(KEYS)
        for k in KEYS:
            assert_struct_equal(got[s][k], expect[s][k], "%s of step %d (async pull vs synchronous download)" % (k, s), skip=())
    # a stale ticket is refused, not answered with somebody else's memory
    v = A.StateView()
    assert eng.t.lib.sge_state_wait(eng.h, 0, C.byref(v)) == A.SGE_ERR_STATE
    # a sub-range and a subset of the arrays; serial order (everything on the main stream)
    eng.set_option(A.OPT_OVERLAP_SKIN, 0)
    eng.tick()
    t = eng.state_pull_async(A.STATE_BODIES | A.STATE_ACTIONS, first=100, count=700)
    part = eng.state_wait(t)
    assert set(part) == {"bodies", "actions"} and part["bodies"].shape[0] == 700
    d = eng.download(first=100, count=700, what=("bodies", "actions"))
    assert_struct_equal(part["bodies"], d["bodies"], "bodies[100:800]", skip=())
    assert_struct_equal(part["actions"], d["actions"], "actions[100:800]", skip=())
    assert eng.move_stats().overflow == 0
    eng.close()


def test_immediate_wait_mode_and_pinned_push(sge):
    """The drop-in loop of GPUCharacterStepSystem.fixedUpdate: push this step's intents through the pinned staging, tick, pull,
    wait for THIS step's pull, hand the arrays to the World — against the same loop over sge_characters_upload / _download.
    Intents change every step (a steering system writes them), and a block of characters is teleported mid-run through the
    staging's body / locomotion arrays."""
    A = sge.abi
    n, steps = 2048, 30
    rng = np.random.default_rng(7)
    plans = [rng.uniform(-6, 6, (n, 3)).astype(np.float32) * np.array([1, 0, 1], np.float32) for _ in range(steps)]

    def run(pinned):
        eng, state = _crowd(sge, n, 43, overlap=1)
        intents = state["intents"].copy()
        out = []
        for s in range(steps):
            intents["desiredVelocity"] = plans[s]
            if pinned:
                stg = eng.state_push_begin(A.STATE_INTENTS)
                stg["intents"][:] = intents
                eng.state_push_commit()
            else:
                eng.upload(intents=intents)
            if s == 11:                                      # teleport characters 64..127, reset their clocks
                cur = eng.download(first=64, count=64, what=("bodies", "locomotion"))
                cur["bodies"]["position"][:, 1] += 2.5
                cur["bodies"]["linearVelocity"][:] = 0
                cur["locomotion"]["motionTime"][:] = 0
                if pinned:
                    stg = eng.state_push_begin(A.STATE_BODIES | A.STATE_LOCOMOTION, first=64, count=64)
                    stg["bodies"][:] = cur["bodies"]
                    stg["locomotion"][:] = cur["locomotion"]
                    eng.state_push_commit()
                else:
                    eng.upload(first=64, bodies=cur["bodies"], locomotion=cur["locomotion"])
            eng.tick()
            if pinned:
                view = eng.state_wait(eng.state_pull_async(A.STATE_WORLD))
                out.append({k: view[k].copy() for k in KEYS})
            else:
                out.append({k: v.copy() for k, v in eng.download(what=KEYS).items()})
        sk = eng.skinned(first_vertex=0, vertex_count=4 * eng.vertex_count)[0].copy()
        assert eng.move_stats().overflow == 0
        eng.close()
        return out, sk

    (a, ska), (b, skb) = run(True), run(False)
    for s in range(steps):
        for k in KEYS:
            assert_struct_equal(a[s][k], b[s][k], "%s after step %d (pinned loop vs upload / download loop)" % (k, s), skip=())
    assert np.array_equal(ska, skb)


def test_push_staging_contract(sge):
    A = sge.abi
    eng, _ = _crowd(sge, 64, 3, overlap=0)
    lib, h = eng.t.lib, eng.h
    v = A.StateView()
    assert lib.sge_state_push_commit(h) == A.SGE_ERR_STATE                       # commit without begin
    assert lib.sge_state_push_begin(h, 0, 0, 0, C.byref(v)) == A.SGE_ERR_INVALID  # nothing asked for
    assert lib.sge_state_push_begin(h, A.STATE_INTENTS, 60, 10, C.byref(v)) == A.SGE_ERR_INVALID  # out of range
    stg = eng.state_push_begin(A.STATE_LOCOMOTION)
    assert lib.sge_state_push_begin(h, A.STATE_INTENTS, 0, 0, C.byref(v)) == A.SGE_ERR_STATE      # one staging at a time
    cur = eng.download(what=("locomotion",))["locomotion"]
    stg["locomotion"][:] = cur
    stg["locomotion"]["profile"][3, 1] = 99                                      # the kernels index the profile table with this
    assert lib.sge_state_push_commit(h) == A.SGE_ERR_INVALID
    t = C.c_int32(-1)
    assert lib.sge_state_pull_async(h, A.STATE_INTENTS, 0, 0, C.byref(t)) == A.SGE_ERR_INVALID    # intents are push-only
    assert lib.sge_state_pull_async(h, A.STATE_WORLD, 0, 0, None) == A.SGE_ERR_INVALID
    after = eng.download(what=("locomotion",))["locomotion"]
    assert_struct_equal(after, cur, "locomotion after a refused push", skip=())
    eng.close()


def test_overlap_value_1_keeps_the_version_1_contract_on_a_caller_stream(sge):
    """ABI version 1 ignored SGE_OPT_OVERLAP_SKIN on a caller-provided stream. Version 2 keeps that for the value 1: a consumer that
    set the option, moved the context to its own stream and enqueues a copy right behind sge_tick — no sge_skin_wait, a palette
    pointer cached once — still reads that step's skinned positions and that step's palettes. (The value 2 is the opt-in; see
    test_overlap_on_a_caller_stream_with_a_consumer.)"""
    import torch

    A = sge.abi
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    n, steps = 1200, 5
    ser, _ = _crowd(sge, n, 19, overlap=0)
    V, B = ser.vertex_count, ser.bone_count
    expect_pos, expect_pal = [], []
    for _ in range(steps):
        ser.tick()
        expect_pos.append(ser.skinned()[0].copy())
        expect_pal.append(ser.palettes()[0].copy())
    ser.close()

    eng, _ = _crowd(sge, n, 19, overlap=1)
    lib, h = eng.t.lib, eng.h
    caller = torch.cuda.Stream(device=0)
    assert lib.sge_context_set_stream(h, C.c_void_p(caller.cuda_stream)) == 0
    pal0, op = C.c_void_p(), C.c_void_p()
    assert lib.sge_crowd_buffers(h, C.byref(pal0), C.byref(op), None, None) == 0
    dev = torch.device("cuda", 0)
    snaps = [torch.zeros((n * V, 3), dtype=torch.float32, device=dev) for _ in range(steps)]
    pals = [torch.zeros((n, B, 16), dtype=torch.float32, device=dev) for _ in range(steps)]
    torch.cuda.synchronize()
    for k in range(steps):
        eng.tick()
        pal = C.c_void_p()
        assert lib.sge_crowd_buffers(h, C.byref(pal), None, None, None) == 0
        assert pal.value == pal0.value                                           # the pointer a version-1 consumer cached stays valid
        assert hip.hipMemcpyAsync(C.c_void_p(snaps[k].data_ptr()), op, n * V * 12, 3, C.c_void_p(caller.cuda_stream)) == 0
        assert hip.hipMemcpyAsync(C.c_void_p(pals[k].data_ptr()), pal0, n * B * 64, 3, C.c_void_p(caller.cuda_stream)) == 0
    caller.synchronize()
    for k in range(steps):
        assert np.array_equal(snaps[k].cpu().numpy(), expect_pos[k]), "positions copied behind tick %d" % k
        assert np.array_equal(pals[k].cpu().numpy(), expect_pal[k]), "palettes copied behind tick %d" % k
    assert lib.sge_context_set_stream(h, None) == 0
    eng.close()


def test_move_lists_are_rebuilt_behind_the_build_in_flight(sge):
    """launch_move builds the NEXT step's scheduling lists on the second stream behind the grouped launch. A tick that cannot use
    them — another character range, another SGE_OPT_HEAVY_THRESHOLD — rebuilds them on the main stream, and has to join the build
    still in flight first (it writes the same lists, counts, flags and histogram). Back-to-back MOVE ticks that alternate ranges
    and thresholds, with no host synchronisation between them, against the same calls with a synchronisation after every tick:
    a character stepped twice or skipped by mixed lists shows as a different body."""
    A = sge.abi
    n = 6000
    move = A.STAGE_INTENT | A.STAGE_GRAVITY | A.STAGE_MOVE

    def run(sync):
        eng, _ = _crowd(sge, n, 57, overlap=1, rings=3, segments=4, footprint=160.0)
        for _ in range(40):                                   # land, so that sweep costs differ widely between characters
            eng.tick(stages=move)
        eng.synchronize()
        half = n // 2
        plan = []
        for r in range(36):
            plan += [("tick", 0, 0), ("tick", 0, half), ("tick", half, n - half), ("thr", 0 if r % 3 == 0 else (900 if r % 3 == 1 else 4000)),
                     ("tick", 0, 0), ("tick", 100, 4000), ("thr", -1 if r % 4 == 3 else 4000), ("tick", 0, 0)]
        snaps = []
        for k, op in enumerate(plan):
            if op[0] == "thr":
                eng.set_option(A.OPT_HEAVY_THRESHOLD, op[1])
            else:
                eng.tick(stages=move, first=op[1], count=op[2])
                if sync:
                    eng.synchronize()
            if k % 48 == 47:
                snaps.append(eng.download(what=("bodies", "controllers")))
        snaps.append(eng.download(what=("bodies", "controllers")))
        assert eng.move_stats().overflow == 0
        eng.close()
        return snaps

    a, b = run(False), run(True)
    assert len(a) == len(b) > 3
    for k, (x, y) in enumerate(zip(a, b)):
        assert_struct_equal(x["bodies"], y["bodies"], "bodies at checkpoint %d" % k, skip=())
        assert_struct_equal(x["controllers"], y["controllers"], "controllers at checkpoint %d" % k, skip=())
