"""AgentSeparationSystem (Systems.swift:1906-2210), row f3: known answers of the oracle's restatement (CPU), the canonical
character-index order, and — on the GPU — bit-exact parity of the HIP stage with the oracle on a crowded real scene."""
import numpy as np
import pytest

import oracle_binding as ob
from scenes import build_scene, compare_states


def _pair_world(sge, eng, positions, velocities=None, mass=None, solid=None, present=True):
    ybot, _, _ = build_scene(sge, eng, 1, terrain_cells=None, rings=3, segments=3)
    n = len(positions)
    eng.resize(n)
    params = sge.assets.default_controller_params(n)
    if present:
        solid = solid if solid is not None else (True,) * n
        params["agentFlags"] = [sge.abi.AGENT_PRESENT | (sge.abi.AGENT_SOLID if s else 0) for s in solid]
        params["agentMassWeight"] = mass if mass is not None else 1.0
    ctrl = sge.assets.default_controller_state(n)
    ctrl["flags"] = sge.abi.CTRL_GROUNDED | sge.abi.CTRL_GROUNDED_NEAR
    bodies = sge.assets.default_bodies(n, np.asarray(positions, np.float64))
    if velocities is not None:
        bodies["linearVelocity"] = velocities
    eng.upload(bodies=bodies, params=params, controllers=ctrl, intents=sge.assets.default_intents(n),
               locomotion=sge.assets.default_locomotion(n, ybot), actions=sge.assets.default_actions(n, ybot, present=True))
    return ybot


REST_Y = -3.0 + 2.5 + 0.05   # capsule (r 1.5, hh 1.0) resting groundSnapSkin above the quad at y = -3


def test_two_overlapping_agents_are_pushed_apart_symmetrically(sge):
    cpu = ob.oracle_engine()
    _pair_world(sge, cpu, [(0.0, REST_Y, 0.0), (2.0, REST_Y, 0.0)], velocities=[(1.0, 0, 0), (-1.0, 0, 0)])
    cpu.tick(dt=1 / 60, stages=sge.abi.STAGE_SEPARATION)
    d = cpu.download(what=("bodies", "controllers"))
    x = d["bodies"]["position"][:, 0]
    # minDist = r + r + min(separationMargin 0.2, skinWidth 0.3) = 3.2 (:1971-1973); equal weights share the correction
    assert abs((x[1] - x[0]) - 3.2) < 2e-6 and abs(x[0] + 0.6) < 2e-6 and abs(x[1] - 2.6) < 2e-6
    assert np.allclose(d["bodies"]["position"][:, 1], REST_Y, atol=1e-6) and np.allclose(d["bodies"]["position"][:, 2], 0, atol=1e-7)
    # approaching along the normal: the relative normal velocity is cancelled, half each (:1991-2000)
    assert np.allclose(d["bodies"]["linearVelocity"][:, 0], [0.0, 0.0], atol=1e-6)
    # moved + not rising: the post-process snap marks the agent grounded (:2108-2133)
    assert ((d["controllers"]["flags"] & sge.abi.CTRL_GROUNDED) != 0).all()
    cpu.close()


def test_weights_height_separation_and_non_solid(sge):
    cpu = ob.oracle_engine()
    # massWeight 0 -> invWeight 0: the other agent takes the whole correction (:2172-2177, :1986-1990)
    _pair_world(sge, cpu, [(0.0, REST_Y, 0.0), (2.0, REST_Y, 0.0)], mass=(0.0, 1.0))
    cpu.tick(stages=sge.abi.STAGE_SEPARATION)
    x = cpu.download(what=("bodies",))["bodies"]["position"][:, 0]
    assert abs(x[0]) < 1e-7 and abs(x[1] - 3.2) < 2e-6
    # height separated (:1974-1975): nothing happens; positions still go through Float (:2210)
    _pair_world(sge, cpu, [(0.0, REST_Y, 0.0), (1.0, REST_Y + 2.0 + 2.0 + 0.2, 0.0)])
    cpu.tick(stages=sge.abi.STAGE_SEPARATION)
    p = cpu.download(what=("bodies",))["bodies"]["position"]
    assert np.array_equal(p[:, 0], [0.0, 1.0])
    # a non-solid agent is not part of the list; an entity WITHOUT the component counts as a default solid agent (:2171)
    _pair_world(sge, cpu, [(0.0, REST_Y, 0.0), (2.0, REST_Y, 0.0), (0.0, REST_Y, 2.0)], solid=(True, False, True))
    cpu.tick(stages=sge.abi.STAGE_SEPARATION)
    p = cpu.download(what=("bodies",))["bodies"]["position"]
    assert p[1, 0] == 2.0 and abs(p[2, 2] - p[0, 2] - 3.2) < 2e-6
    _pair_world(sge, cpu, [(0.0, REST_Y, 0.0), (2.0, REST_Y, 0.0)], present=False)
    cpu.tick(stages=sge.abi.STAGE_SEPARATION)
    x = cpu.download(what=("bodies",))["bodies"]["position"][:, 0]
    assert abs((x[1] - x[0]) - 3.2) < 2e-6
    cpu.close()


def test_order_is_the_character_index(sge):
    """Three agents in a row: pair (0,1) is resolved before (1,2) and (0,2) sees the stale copy of agent 0 (:1952) — the result
    differs from the mirrored order, so the canonical order is observable."""
    cpu = ob.oracle_engine()
    pts = [(0.0, REST_Y, 0.0), (2.0, REST_Y, 0.0), (4.0, REST_Y, 0.0)]
    _pair_world(sge, cpu, pts, mass=(1, 1, 1), solid=(True, True, True))
    cpu.separation_params(iterations=1)
    cpu.tick(stages=sge.abi.STAGE_SEPARATION)
    a = cpu.download(what=("bodies",))["bodies"]["position"][:, 0].copy()
    _pair_world(sge, cpu, pts[::-1], mass=(1, 1, 1), solid=(True, True, True))
    cpu.tick(stages=sge.abi.STAGE_SEPARATION)
    b = cpu.download(what=("bodies",))["bodies"]["position"][::-1, 0].copy()
    # pair (0,1): -0.6 / +0.6 -> x = (-0.6, 2.6, 4); then (1,2): dist 1.4, penetration 1.8 -> (-0.6, 1.7, 4.9)
    assert np.allclose(a, [-0.6, 1.7, 4.9], atol=3e-6)
    assert np.allclose(b, [-0.9, 2.3, 4.6], atol=3e-6)
    cpu.separation_params()   # defaults again
    cpu.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["default", "one_wavefront"])
def test_separation_stage_gpu_parity(sge, monkeypatch, form):
    """f3 on the GPU: 192 solid agents crowded onto the real cheese + mirror scene, move-and-slide + character-vs-character sweeps +
    AgentSeparationSystem every step: positions, velocities and controller state bit-exact with the oracle. `default` is the dataflow
    over agents (any crowd of more than a few dozen), `one_wavefront` the loop walked by one wavefront with everything in LDS
    (SGE_SEPARATION_FLOW=0: what smaller crowds get)."""
    import torch

    if form == "one_wavefront":
        monkeypatch.setenv("SGE_SEPARATION_FLOW", "0")

    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    n = 192
    for e in (gpu, cpu):
        _, _, st0 = build_scene(sge, e, n, seed=41, mixed=True, agents=True, rings=3, segments=3, asset_scene=("cheese", "mirror"), footprint=60.0)
        p = st0["params"].copy()
        p["agentFlags"][::9] = 0                                   # no AgentCollisionComponent: default solid agent
        p["agentFlags"][4::13] = sge.abi.AGENT_PRESENT               # present, not solid
        p["agentMassWeight"][::5] = 2.5
        p["agentMassWeight"][7::31] = 0.0
        p["agentFlags"][3::17] |= sge.abi.AGENT_RADIUS_OVERRIDE
        p["agentRadiusOverride"][:] = 1.2
        e.upload(params=p)
    ex = sge.parallel.AgentExchange(gpu, n, 0, 1, torch.device("cuda", 0), None)
    st = (sge.abi.STAGE_ALL & ~sge.abi.STAGE_SKIN) | sge.abi.STAGE_SEPARATION
    moved = 0
    for s in range(90):
        before = gpu.download(what=("bodies",))["bodies"]["position"].copy()
        ex.step(stages=st)
        ob.tick_mt(cpu, 8, stages=st | sge.abi.STAGE_AGENTS)
        if s % 6 == 0 or s == 89:
            gpu.synchronize()
            compare_states(sge, gpu, cpu, n)
    # the stage does something in this scene: compare with a run without it
    free = ob.oracle_engine()
    build_scene(sge, free, n, seed=41, mixed=True, agents=True, rings=3, segments=3, asset_scene=("cheese", "mirror"), footprint=60.0)
    for s in range(90):
        ob.tick_mt(free, 8, stages=(st & ~sge.abi.STAGE_SEPARATION) | sge.abi.STAGE_AGENTS)
    a = free.download(what=("bodies",))["bodies"]["position"]
    b = cpu.download(what=("bodies",))["bodies"]["position"]
    assert np.abs(a - b).max() > 1e-2, "the scene must exercise agent separation"
    # capacity and range errors
    with pytest.raises(sge.SgeError):
        gpu.tick(stages=sge.abi.STAGE_SEPARATION, first=1, count=5)
    # a context that holds one shard of the crowd (its snapshot names agents of other contexts) refuses the stage: the pair loop
    # is sequential over the WHOLE crowd in index order, there is nothing a shard could compute that equals it
    other = torch.zeros((2 * n, 8), dtype=torch.float32, device="cuda:0")
    other[:, 3] = -1.0
    torch.cuda.synchronize()
    gpu.agents_import(other.data_ptr(), 2 * n, n)
    with pytest.raises(sge.SgeError, match="whole crowd in one context"):
        gpu.tick(stages=sge.abi.STAGE_SEPARATION)
    for e in (gpu, cpu, free):
        e.close()


@pytest.mark.gpu
def test_separation_dataflow_path_on_a_small_crowd(sge, monkeypatch):
    """The crowd path of the stage (agents as a dataflow: per-agent pass counters, loops drawn from a ticket counter, sge_ccd.hip)
    forced onto the 192-agent scene of the test above (SGE_SEPARATION_FLOW=1): bit-exact with the oracle's sequential loop, and
    without the serial redo (the flags stay clear), so what is compared IS the dataflow's result."""
    import torch

    monkeypatch.setenv("SGE_SEPARATION_FLOW", "1")
    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    n = 192
    for e in (gpu, cpu):
        _, _, st0 = build_scene(sge, e, n, seed=41, mixed=True, agents=True, rings=3, segments=3, asset_scene=("cheese", "mirror"), footprint=60.0)
        p = st0["params"].copy()
        p["agentFlags"][::9] = 0
        p["agentFlags"][4::13] = sge.abi.AGENT_PRESENT
        p["agentMassWeight"][::5] = 2.5
        p["agentMassWeight"][7::31] = 0.0
        p["agentFlags"][3::17] |= sge.abi.AGENT_RADIUS_OVERRIDE
        p["agentRadiusOverride"][:] = 1.2
        e.upload(params=p)
    ex = sge.parallel.AgentExchange(gpu, n, 0, 1, torch.device("cuda", 0), None)
    st = (sge.abi.STAGE_ALL & ~sge.abi.STAGE_SKIN) | sge.abi.STAGE_SEPARATION
    info = np.zeros(4, np.int32)
    redone = 0
    for s in range(45):
        ex.step(stages=st)
        ob.tick_mt(cpu, 8, stages=st | sge.abi.STAGE_AGENTS)
        assert gpu.t.lib.sge_debug_separation(gpu.h, sge.abi.ptr(info)) == 0
        redone += int(info[2] != 0)
        if s % 5 == 0 or s == 44:
            compare_states(sge, gpu, cpu, n)
    assert info[0] > 100 and info[1] >= info[0]
    assert redone == 0, "the dataflow pass fell back to the serial form in %d steps" % redone
    gpu.close()
    cpu.close()


@pytest.mark.gpu
def test_separation_crowd_of_8192(sge):
    """Eight times the old capacity, on the real cheese + mirror scene at a density where most agents overlap a neighbour
    (spacing ~2.2 units, capsule diameter 3): move-and-slide + AgentSeparationSystem, bodies and controllers bit-exact with the
    oracle's sequential loop after every step."""
    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    n = 8192
    for e in (gpu, cpu):
        build_scene(sge, e, n, seed=43, mixed=True, agents=True, rings=3, segments=3, asset_scene=("cheese", "mirror"))
    st = (sge.abi.STAGE_ALL & ~sge.abi.STAGE_SKIN) | sge.abi.STAGE_SEPARATION
    info = np.zeros(4, np.int32)
    dataflow_steps = 0
    for s in range(8):
        gpu.tick(stages=st)
        ob.tick_mt(cpu, 8, stages=st)
        compare_states(sge, gpu, cpu, n)
        assert gpu.t.lib.sge_debug_separation(gpu.h, sge.abi.ptr(info)) == 0
        assert info[0] == n
        dataflow_steps += int(info[2] == 0)
    assert dataflow_steps >= 3, "only %d of 8 steps were not redone by the serial kernel" % dataflow_steps
    assert gpu.move_stats().overflow == 0
    gpu.close()
    cpu.close()


@pytest.mark.gpu
def test_separation_crowd_forms_agree_over_a_settling_crowd(sge, monkeypatch):
    """The crowd path in three settings (sge_ccd.hip): the first form of the loops casts every pair through the BVH one after the other and
    keeps the order of the loops that touch an agent as a counter behind a fence (SGE_SEPARATION_BVH_CASTS=1); the second keeps the
    changing pairs in LDS, takes their casts from per-agent cached triangles in rounds, skips casts that provably hit nothing and
    carries the counter inside the data — once with every loop waiting for the outer ring of its candidates
    (SGE_SEPARATION_NO_DEFER=1) and once, the default, passing the ring whenever it comes due.
    Same crowd, 8,192 agents on the cheese + mirror scene while it settles from its spawn overlaps (24 steps: long pushes, blocked
    pairs, agents beside the ornate mirror): bodies and controllers byte-identical after every step, and in the later steps no pass
    is redone by the serial kernel, so what is compared is the three dataflows."""
    n = 8192
    forms = ({"SGE_SEPARATION_BVH_CASTS": "1", "SGE_SEPARATION_NO_DEFER": "0"}, {"SGE_SEPARATION_BVH_CASTS": "0", "SGE_SEPARATION_NO_DEFER": "1"},
             {"SGE_SEPARATION_BVH_CASTS": "0", "SGE_SEPARATION_NO_DEFER": "0"})
    engines = []
    for _ in forms:
        e = sge.CharacterEngine(0)
        build_scene(sge, e, n, seed=43, mixed=True, agents=True, rings=3, segments=3, asset_scene=("cheese", "mirror"))
        engines.append(e)
    st = (sge.abi.STAGE_ALL & ~sge.abi.STAGE_SKIN) | sge.abi.STAGE_SEPARATION
    info = np.zeros(4, np.int32)
    redone_late = 0
    for s in range(24):
        for form, e in zip(forms, engines):
            for k, v in form.items():
                monkeypatch.setenv(k, v)
            e.tick(stages=st)
            assert e.t.lib.sge_debug_separation(e.h, sge.abi.ptr(info)) == 0
            if s >= 12:
                redone_late += int(info[2] != 0)
        states = [e.download(what=("bodies", "controllers")) for e in engines]
        for other in states[1:]:
            for name in ("bodies", "controllers"):
                assert states[0][name].tobytes() == other[name].tobytes(), "step %d: %s differ between the forms" % (s, name)
    assert redone_late == 0
    for e in engines:
        e.close()


@pytest.mark.gpu
def test_separation_crowd_of_31250_forms_agree(sge, monkeypatch):
    """One GPU's share of configs[3] / configs[4]: 31,250 agents spawned at 1.6 units' spacing with capsules 3 wide — every agent
    overlaps eight others, a loop has 300-500 candidates (several 64-lane chunks, the ring re-queued from every one of them), and
    after the first step the stage takes its candidates from 7 x 7 cells. The two forms of the pair loops on the same crowd, eight
    steps: bodies and controllers byte-identical after every step (the first steps are redone by the serial kernel in both — an
    agent is pushed further than a cell — so they compare that kernel with itself; the later ones compare the two dataflows), every
    solid agent listed, and the stage leaves the crowd less entangled than it found it."""
    n = 31250
    forms = ({"SGE_SEPARATION_BVH_CASTS": "1"}, {"SGE_SEPARATION_BVH_CASTS": "0"})
    engines = []
    for _ in forms:
        e = sge.CharacterEngine(0)
        build_scene(sge, e, n, seed=43, mixed=True, agents=True, rings=3, segments=3, asset_scene=("cheese", "mirror"))
        engines.append(e)
    st = (sge.abi.STAGE_ALL & ~sge.abi.STAGE_SKIN) | sge.abi.STAGE_SEPARATION
    info = np.zeros(4, np.int32)

    def overlapping_pairs(pos, radius=1.5):
        # agents closer than two radii in XZ, counted over a uniform grid (numpy; a measure of entanglement, not of the reference's rule)
        cell = 2 * radius
        key = np.floor(pos[:, [0, 2]] / cell).astype(np.int64)
        order = np.lexsort((key[:, 1], key[:, 0]))
        p = pos[order][:, [0, 2]]
        total = 0
        for shift in range(1, 40):
            d = p[shift:] - p[:-shift]
            total += int(((d * d).sum(1) < (2 * radius) ** 2).sum())
        return total

    before = overlapping_pairs(engines[0].download(what=("bodies",))["bodies"]["position"].astype(np.float64))
    redone = []
    for s in range(8):
        flags = []
        for form, e in zip(forms, engines):
            for k, v in form.items():
                monkeypatch.setenv(k, v)
            e.tick(stages=st)
            assert e.t.lib.sge_debug_separation(e.h, sge.abi.ptr(info)) == 0
            assert info[0] == n
            flags.append(int(info[2]))
        redone.append(flags)
        a, b = (e.download(what=("bodies", "controllers")) for e in engines)
        for name in ("bodies", "controllers"):
            assert a[name].tobytes() == b[name].tobytes(), "step %d: %s differ between the two forms" % (s, name)
    assert redone[-1] == [0, 0] and redone[-2] == [0, 0], "the dataflow passes of the last steps were redone serially: %r" % (redone,)
    after = overlapping_pairs(engines[0].download(what=("bodies",))["bodies"]["position"].astype(np.float64))
    assert after < before, (before, after)
    for e in engines:
        e.close()


@pytest.mark.gpu
def test_separation_crowd_of_2048_soak(sge):
    """The default form of the crowd path against the oracle's sequential loop over time: 2,048 agents packed onto a quarter of the
    cheese + mirror scene (capsules 3 wide at ~2.2 units' spacing: every agent overlaps its neighbours, pairs are blocked at the
    mirror, agents stand on the bumps of the cheese), 40 steps, bodies and controllers bit-exact after every fifth."""
    gpu = sge.CharacterEngine(0)
    cpu = ob.oracle_engine()
    n = 2048
    for e in (gpu, cpu):
        build_scene(sge, e, n, seed=47, mixed=True, agents=True, rings=3, segments=3, asset_scene=("cheese", "mirror"), footprint=100.0)
    st = (sge.abi.STAGE_ALL & ~sge.abi.STAGE_SKIN) | sge.abi.STAGE_SEPARATION
    info = np.zeros(4, np.int32)
    redone = 0
    for s in range(40):
        gpu.tick(stages=st)
        ob.tick_mt(cpu, 8, stages=st)
        assert gpu.t.lib.sge_debug_separation(gpu.h, sge.abi.ptr(info)) == 0
        redone += int(info[2] != 0)
        if s % 5 == 4:
            compare_states(sge, gpu, cpu, n)
    assert info[0] == n
    print("steps redone serially: %d of 40" % redone)
    assert redone <= 10, "%d of 40 steps fell back to the serial kernel: the dataflow was hardly compared" % redone
    gpu.close()
    cpu.close()
