//  GPUCrowd.swift — the World <-> GPU state bridge of the MI355X character path.
//
//  NOT COMPILED IN THIS REPOSITORY (no Swift toolchain in the build image). It is the binding a maintainer of
//  kelian343/swift-game-engine adds next to Game/Systems.swift; it mirrors swift-game-engine_amd/host/sge_host.hpp (which IS compiled
//  and run on the GPU, tests/cpp/host_mirror_smoke.cpp) call for call. C ABI: include/sge_amd.h, imported as module CSGE.
//
//  What it replaces: the per-entity dictionary traffic of `world.store(T.self)[e]` (World.swift:64-75) inside the hot systems. The GPU
//  keeps the authoritative copy of every character's PhysicsBodyComponent / CharacterControllerComponent / locomotion / action state
//  as SoA-of-PODs; the World's copies are refreshed once per fixed step (beginPull right behind sge_tick, pullBack when the host wants
//  them), and whatever other Swift systems wrote into the World since the last step (intents, teleports, dodge / jump edits) is
//  pushed before the step (pushDirtyState). Both directions go through the context's pinned memory, ordered by events
//  (sge_state_pull_async / sge_state_wait, sge_state_push_begin / _commit): no call here synchronises the whole context, and none
//  waits for the step's skin launch.

import simd
import CSGE

public final class GPUCrowd {
    public let ctx: OpaquePointer
    /// index in the GPU arrays = position here. Sorted by entity id: the canonical order SURVEY 8 f3 asks for
    /// (World.query returns Dictionary order, which is hash-seed dependent, World.swift:99-117).
    public private(set) var entities: [Entity] = []
    private var indexOf: [Entity: Int32] = [:]
    private var profileIndex: [String: Int32] = [:]   // MotionProfile.name (Animation.swift:45) -> row of sge_motion_profiles_upload
    private var profileTable: [MotionProfile] = []
    private var skeleton: Skeleton?
    // host mirrors of the PODs, reused every step
    private var bodies: [sge_body_state] = []
    private var params: [sge_controller_params] = []
    private var controllers: [sge_controller_state] = []
    private var intents: [sge_move_intent] = []
    private var locomotion: [sge_locomotion_state] = []
    private var actions: [sge_action_state] = []
    public private(set) var paletteCount: Int = 0

    /// nil when no gfx950 device is visible (there is no CPU fallback) — the shape of RTSkinningEncoder.init?(device:)
    public init?(device: Int32 = 0) {
        guard let c = sge_context_create(device) else { return nil }
        ctx = c
    }
    deinit { sge_context_destroy(ctx) }

    @inline(__always) func check(_ rc: Int32, _ what: StaticString = #function) {
        precondition(rc == SGE_OK, "\(what): \(String(cString: sge_last_error()))")
    }

    // MARK: assets, once ------------------------------------------------------------------------------------------------------

    /// Skeleton.swift:127-173 — every array is already in the layout the ABI wants (matrix_float4x4 = 16 floats, column-major).
    public func upload(skeleton s: Skeleton) {
        skeleton = s
        paletteCount = s.boneCount
        var parent = s.parent.map { Int32($0) }
        var desc = sge_skeleton_desc()
        desc.boneCount = Int32(s.boneCount)
        desc.unitScale = s.unitScale
        desc.pelvisIndex = Int32(s.semantic(.pelvis) ?? -1)
        desc.leanIndex = Int32(s.semantic(.chest) ?? s.semantic(.spine3) ?? s.semantic(.spine2) ?? s.semantic(.spine1) ?? -1)  // ProceduralPoseSystem.swift:371-374
        withUnsafeBytes(of: s.rootRotationFix) { src in withUnsafeMutableBytes(of: &desc.rootRotationFix) { $0.copyMemory(from: src) } }
        let rest = packed(s.restTranslation), raw = packed(s.rawRestTranslation), pre = packed(s.preRotationDegrees)
        s.bindLocal.withUnsafeBytes { bl in s.invBindModel.withUnsafeBytes { ib in
        rest.withUnsafeBufferPointer { rt in raw.withUnsafeBufferPointer { rr in pre.withUnsafeBufferPointer { pr in
        parent.withUnsafeMutableBufferPointer { pa in
            desc.parent = UnsafePointer(pa.baseAddress)
            desc.bindLocal = bl.bindMemory(to: Float.self).baseAddress
            desc.invBindModel = ib.bindMemory(to: Float.self).baseAddress
            desc.restTranslation = rt.baseAddress
            desc.rawRestTranslation = rr.baseAddress
            desc.preRotationDegrees = pr.baseAddress
            check(sge_skeleton_upload(ctx, &desc))
        }}}}}}
    }

    /// Animation.swift:11-53 flattened per skeleton bone. Call with every MotionProfile a character may reference.
    public func upload(profiles: [MotionProfile]) {
        guard let s = skeleton else { preconditionFailure("upload(skeleton:) first") }
        profileTable = profiles
        profileIndex.removeAll()
        let B = s.boneCount, C = Int(SGE_MAX_COEFFS)
        var present = [[UInt8]](), counts = [[UInt8]](), coeffs = [[Float]]()
        for (k, p) in profiles.enumerated() {
            profileIndex[p.name] = Int32(k)
            var pr = [UInt8](repeating: 0, count: B)
            var cn = [UInt8](repeating: UInt8(SGE_AXIS_ABSENT), count: B * 6)
            var co = [Float](repeating: 0, count: B * 6 * C)
            for i in 0..<B {
                guard let bone = p.bones[s.names[i]] else { continue }          // String-keyed lookup done ONCE here, not per frame
                pr[i] = 1
                let axes: [[Float]?] = [bone.translation.x, bone.translation.y, bone.translation.z,
                                        bone.rotation.x, bone.rotation.y, bone.rotation.z]
                for (a, values) in axes.enumerated() {
                    guard let v = values else { continue }                       // nil axis stays SGE_AXIS_ABSENT
                    cn[i * 6 + a] = UInt8(min(v.count, C))
                    for (j, x) in v.prefix(C).enumerated() { co[(i * 6 + a) * C + j] = x }
                }
            }
            present.append(pr); counts.append(cn); coeffs.append(co)
        }
        var descs = [sge_motion_profile_desc](repeating: sge_motion_profile_desc(), count: profiles.count)
        // keep the arrays alive across the call
        withExtendedLifetime((present, counts, coeffs)) {
            for k in profiles.indices {
                descs[k].order = Int32(profiles[k].order)
                descs[k].cycleDuration = profiles[k].phase?.cycleDuration ?? profiles[k].duration
                present[k].withUnsafeBufferPointer { descs[k].bonePresent = $0.baseAddress }
                counts[k].withUnsafeBufferPointer { descs[k].coeffCount = $0.baseAddress }
                coeffs[k].withUnsafeBufferPointer { descs[k].coeffs = $0.baseAddress }
            }
            check(sge_motion_profiles_upload(ctx, descs, Int32(descs.count)))
        }
    }

    /// SkinnedMeshDescriptor (ProceduralMeshAPI.swift:143-181): SoA streams once; tangents as RTGeometryCache.swift:514 computes them.
    public func upload(skinnedMesh d: SkinnedMeshDescriptor) {
        let st = d.streams
        let pos = packed(st.positions), nrm = packed(st.normals)
        let tan = MeshTangents.compute(positions: st.positions, normals: st.normals, uvs: st.uvs, indices16: d.indices16, indices32: d.indices32)
        var desc = sge_skinned_mesh_desc()
        desc.vertexCount = Int32(st.positions.count)
        pos.withUnsafeBufferPointer { p in nrm.withUnsafeBufferPointer { n in tan.withUnsafeBytes { t in
        st.boneIndices.withUnsafeBytes { bi in st.boneWeights.withUnsafeBytes { bw in
            desc.positions = p.baseAddress; desc.normals = n.baseAddress
            desc.tangents = t.bindMemory(to: Float.self).baseAddress
            desc.boneIndices = bi.bindMemory(to: UInt16.self).baseAddress          // SIMD4<UInt16>: 4 x u16, no padding
            desc.boneWeights = bw.bindMemory(to: Float.self).baseAddress
            if let inv = d.invBindModel {                                          // re-bind of Systems.swift:2519-2527 folded into the palette
                inv.withUnsafeBytes { ib in
                    desc.invBindModel = ib.bindMemory(to: Float.self).baseAddress
                    desc.invBindCount = Int32(inv.count)
                    check(sge_skinned_mesh_upload(ctx, &desc))
                }
            } else {
                check(sge_skinned_mesh_upload(ctx, &desc))
            }
        }}}}}
    }

    // MARK: characters -----------------------------------------------------------------------------------------------------

    /// (Re)builds the entity <-> index tables from the World and uploads every component. Call when characters are created or
    /// destroyed (structural change), not per step.
    public func rebuild(from world: World) {
        entities = world.query(PhysicsBodyComponent.self, CharacterControllerComponent.self).sorted { $0.id < $1.id }
        indexOf = Dictionary(uniqueKeysWithValues: entities.enumerated().map { ($1, Int32($0)) })
        let n = entities.count
        bodies = .init(repeating: sge_body_state(), count: n); params = .init(repeating: sge_controller_params(), count: n)
        controllers = .init(repeating: sge_controller_state(), count: n); intents = .init(repeating: sge_move_intent(), count: n)
        locomotion = .init(repeating: sge_locomotion_state(), count: n); actions = .init(repeating: sge_action_state(), count: n)
        check(sge_characters_resize(ctx, Int32(n)))
        for i in 0..<n { encode(entity: entities[i], at: i, world: world) }
        check(sge_characters_upload(ctx, 0, Int32(n), bodies, params, controllers, intents, locomotion, actions))
    }

    /// Everything other Swift systems may have written into the World since the last step (PhysicsIntentSystem's inputs, jump / dodge
    /// edits of the velocity, teleports): re-encode and upload. The bodies / controllers arrays are authoritative on the GPU, so only
    /// entities flagged dirty by their writers need a body upload; intents are small and go every step.
    public func pushDirtyState(from world: World, dirtyBodies: Set<Entity> = []) {
        let n = entities.count
        guard n > 0 else { return }
        let mStore = world.store(MoveIntentComponent.self), mvStore = world.store(MovementComponent.self), dStore = world.store(DodgeActionComponent.self)
        var staging = sge_state_view()
        check(sge_state_push_begin(ctx, UInt32(SGE_STATE_INTENTS), 0, Int32(n), &staging))        // pinned staging, filled in place
        for i in 0..<n { staging.intents[i] = encodeIntent(mStore[entities[i]], mvStore[entities[i]], dStore[entities[i]]) }
        check(sge_state_push_commit(ctx))                                                          // copies enqueued in front of the next tick
        for e in dirtyBodies {
            guard let i = indexOf[e] else { continue }
            encode(entity: e, at: Int(i), world: world)
            check(sge_state_push_begin(ctx, UInt32(SGE_STATE_BODIES) | UInt32(SGE_STATE_CONTROLLERS), i, 1, &staging))
            staging.bodies[0] = bodies[Int(i)]
            staging.controllers[0] = controllers[Int(i)]
            check(sge_state_push_commit(ctx))
        }
    }

    /// Right behind the step's sge_tick: the snapshot of this step starts its way to pinned host memory (taken on the device behind
    /// the kernels that wrote the arrays, so the next tick may be enqueued at once). Returns immediately.
    private var pendingPull: Int32 = -1
    public func beginPull(which: UInt32 = UInt32(SGE_STATE_WORLD)) {
        guard !entities.isEmpty else { return }
        check(sge_state_pull_async(ctx, which, 0, 0, &pendingPull))
    }

    /// After the step: what the reference's systems would have written into the World (KinematicMoveStopSystem.writeBack
    /// Systems.swift:1802-1821, LocomotionProfileSystem :279-407, ActionAnimationSystem :475-517, PhysicsWritebackSystem :2249-2267).
    /// Waits for the pull begun last — its copy only, not the skin launch, not later ticks — and decodes it from pinned memory.
    public func pullBack(into world: World, palettes: Bool = false) {
        guard pendingPull >= 0 else { return }
        var v = sge_state_view()
        check(sge_state_wait(ctx, pendingPull, &v))
        pendingPull = -1
        let n = Int(v.count), first = Int(v.first)
        let pStore = world.store(PhysicsBodyComponent.self), cStore = world.store(CharacterControllerComponent.self)
        let lStore = world.store(LocomotionProfileComponent.self), mStore = world.store(MotionProfileComponent.self)
        let aStore = world.store(ActionAnimationComponent.self), tStore = world.store(TransformComponent.self)
        for i in 0..<n {
            let e = entities[first + i]
            if let vb = v.bodies {
                let body = vb[i]
                if var b = pStore[e] { decode(body, into: &b); pStore[e] = b }
                if var t = tStore[e] {      // PhysicsWritebackSystem: TransformComponent from the body
                    t.translation = SIMD3<Float>(Float(body.position.0), Float(body.position.1), Float(body.position.2))
                    t.rotation = quat(body.transformRotation)
                    tStore[e] = t
                }
            }
            if let vc = v.controllers, var c = cStore[e] { decode(vc[i], into: &c); cStore[e] = c }
            if let vl = v.locomotion {
                if var l = lStore[e] { decode(vl[i], into: &l); lStore[e] = l }
                if var m = mStore[e] { m.time = vl[i].motionTime; mStore[e] = m }
            }
            if let va = v.actions, var a = aStore[e] { decode(va[i], into: &a); aStore[e] = a }
        }
        if palettes, let s = skeleton, let vl = v.locomotion {     // only if something on the CPU still wants PoseComponent.palette
            var flat = [Float](repeating: 0, count: entities.count * s.boneCount * 16)
            check(sge_palettes_download(ctx, 0, Int32(entities.count), &flat, nil, nil))         // (synchronises: debugging aid, not a per-step path)
            let poseStore = world.store(PoseComponent.self)
            for i in 0..<n {
                guard var pose = poseStore[entities[first + i]] else { continue }
                flat.withUnsafeBytes { raw in
                    let base = raw.baseAddress!.advanced(by: (first + i) * s.boneCount * 64).assumingMemoryBound(to: matrix_float4x4.self)
                    pose.palette = Array(UnsafeBufferPointer(start: base, count: s.boneCount))
                }
                pose.phase = vl[i].posePhase
                poseStore[entities[first + i]] = pose
            }
        }
    }

    public func index(of e: Entity) -> Int32? { indexOf[e] }

    // MARK: component <-> POD, field for field (Components.swift) --------------------------------------------------------------

    private func encode(entity e: Entity, at i: Int, world: World) {
        let body = world.store(PhysicsBodyComponent.self)[e]!, c = world.store(CharacterControllerComponent.self)[e]!
        let transform = world.store(TransformComponent.self)[e]
        // PhysicsBodyComponent :549-598 (position / velocity stay Double)
        var b = sge_body_state()
        b.position = (body.position.x, body.position.y, body.position.z)
        b.linearVelocity = (body.linearVelocity.x, body.linearVelocity.y, body.linearVelocity.z)
        b.rotation = tuple(body.rotation)
        b.transformRotation = tuple(transform?.rotation ?? body.rotation)         // what PoseStackSystem reads one step stale (:345)
        b.bodyType = body.bodyType == .static ? UInt32(SGE_BODY_STATIC) : (body.bodyType == .kinematic ? UInt32(SGE_BODY_KINEMATIC) : UInt32(SGE_BODY_DYNAMIC))
        bodies[i] = b
        // CharacterControllerComponent :353-431, constants
        var p = sge_controller_params()
        p.radius = c.radius; p.halfHeight = c.halfHeight; p.skinWidth = c.skinWidth; p.groundSnapSkin = c.groundSnapSkin
        p.snapDistance = c.snapDistance; p.fallProbeDistance = c.fallProbeDistance
        p.groundSnapMaxSpeed = c.groundSnapMaxSpeed; p.groundSnapMaxToi = c.groundSnapMaxToi
        p.groundSnapMaxStep = c.groundSnapMaxStep; p.groundSweepMaxStep = c.groundSweepMaxStep
        p.maxSlideIterations = Int32(c.maxSlideIterations); p.minGroundDot = c.minGroundDot; p.collisionMask = c.collisionMask
        if let agent = world.store(AgentCollisionComponent.self)[e] {              // AgentCollisionComponent :433-445
            p.agentFlags = UInt32(SGE_AGENT_PRESENT) | (agent.isSolid ? UInt32(SGE_AGENT_SOLID) : 0) | (agent.radiusOverride != nil ? UInt32(SGE_AGENT_RADIUS_OVERRIDE) : 0)
            p.agentRadiusOverride = agent.radiusOverride ?? 0
            p.agentMassWeight = agent.massWeight
        } else {
            p.agentFlags = 0; p.agentRadiusOverride = 0; p.agentMassWeight = 1
        }
        params[i] = p
        // CharacterControllerComponent, per-step state
        var s = sge_controller_state()
        s.groundNormal = (c.groundNormal.x, c.groundNormal.y, c.groundNormal.z)
        s.groundTriangleIndex = Int32(c.groundTriangleIndex)
        s.sideContactNormal = (c.sideContactNormal.x, c.sideContactNormal.y, c.sideContactNormal.z)
        s.sideContactFrames = Int32(c.sideContactFrames)
        let m = min(c.contactManifoldTriangles.count, min(c.contactManifoldNormals.count, Int(SGE_MANIFOLD_MAX)))
        withUnsafeMutableBytes(of: &s.manifoldTriangles) { raw in
            let t = raw.bindMemory(to: Int32.self); for k in 0..<m { t[k] = Int32(c.contactManifoldTriangles[k]) }
        }
        withUnsafeMutableBytes(of: &s.manifoldNormals) { raw in
            let f = raw.bindMemory(to: Float.self)
            for k in 0..<m { f[3 * k] = c.contactManifoldNormals[k].x; f[3 * k + 1] = c.contactManifoldNormals[k].y; f[3 * k + 2] = c.contactManifoldNormals[k].z }
        }
        s.manifoldCount = Int32(m); s.manifoldFrames = Int32(c.contactManifoldFrames)
        s.groundTransitionFrames = Int32(c.groundTransitionFrames)
        s.flags = (c.grounded ? UInt32(SGE_CTRL_GROUNDED) : 0) | (c.groundedNear ? UInt32(SGE_CTRL_GROUNDED_NEAR) : 0) | (c.groundSliding ? UInt32(SGE_CTRL_GROUND_SLIDING) : 0)
        s.groundDistance = c.groundDistance
        controllers[i] = s
        intents[i] = encodeIntent(world.store(MoveIntentComponent.self)[e], world.store(MovementComponent.self)[e], world.store(DodgeActionComponent.self)[e])
        // LocomotionProfileComponent + MotionProfileComponent :203-293
        var l = sge_locomotion_state()
        if let lc = world.store(LocomotionProfileComponent.self)[e] {
            l.profile = (row(lc.idleProfile), row(lc.walkProfile), row(lc.runProfile), row(lc.fallProfile))
            l.time = (lc.idleTime, lc.walkTime, lc.runTime, lc.fallTime)
            l.idleEnterSpeed = lc.idleEnterSpeed; l.idleExitSpeed = lc.idleExitSpeed; l.runEnterSpeed = lc.runEnterSpeed; l.runExitSpeed = lc.runExitSpeed
            l.fallMinDropHeight = lc.fallMinDropHeight; l.blendTime = lc.blendTime; l.blendT = lc.blendT
            l.idleInertiaHalfLife = lc.idleInertiaHalfLife; l.idleInertia = lc.idleInertia
            l.fromState = Int32(lc.fromState.rawValue); l.state = Int32(lc.state.rawValue)
            l.flags |= UInt32(SGE_LOCO_PRESENT) | (lc.isBlending ? UInt32(SGE_LOCO_IS_BLENDING) : 0)
        }
        if let mp = world.store(MotionProfileComponent.self)[e] {
            l.flags |= UInt32(SGE_MOTION_PRESENT) | (mp.loop ? UInt32(SGE_MOTION_LOOP) : 0) | (mp.inPlace ? UInt32(SGE_MOTION_IN_PLACE) : 0)
            l.motionTime = mp.time; l.playbackRate = mp.playbackRate; l.motionProfile = row(mp.profile)
        }
        locomotion[i] = l
        // ActionAnimationComponent :620-653 (+ the dodge window of Systems.swift:488)
        var a = sge_action_state()
        if let ac = world.store(ActionAnimationComponent.self)[e] {
            a.profile = row(ac.profile); a.time = ac.time; a.playbackRate = ac.playbackRate; a.weight = ac.weight
            a.blendInTime = ac.blendInTime; a.blendOutHalfLife = ac.blendOutHalfLife
            a.flags = UInt32(SGE_ACTION_PRESENT) | (ac.active ? UInt32(SGE_ACTION_ACTIVE) : 0) | (ac.loop ? UInt32(SGE_ACTION_LOOP) : 0)
                | (ac.inPlace ? UInt32(SGE_ACTION_IN_PLACE) : 0) | (ac.exiting ? UInt32(SGE_ACTION_EXITING) : 0)
            if let dodge = world.store(DodgeActionComponent.self)[e] {
                a.flags |= UInt32(SGE_ACTION_HAS_DODGE)
                a.dodgeEnd = dodge.endTime > 0 ? dodge.endTime : dodge.duration
            }
        }
        actions[i] = a
    }

    /// MoveIntentComponent + MovementComponent (:600-618, :684-702) as PhysicsIntentSystem consumes them (Systems.swift:205-250)
    private func encodeIntent(_ intent: MoveIntentComponent?, _ movement: MovementComponent?, _ dodge: DodgeActionComponent?) -> sge_move_intent {
        var out = sge_move_intent()
        guard let intent = intent else { return out }
        out.desiredVelocity = (intent.desiredVelocity.x, intent.desiredVelocity.y, intent.desiredVelocity.z)
        out.desiredFacingYaw = intent.desiredFacingYaw
        out.flags = UInt32(SGE_INTENT_PRESENT) | (intent.hasFacingYaw ? UInt32(SGE_INTENT_HAS_FACING_YAW) : 0) | ((dodge?.active ?? false) ? UInt32(SGE_INTENT_DODGE_ACTIVE) : 0)
        out.maxAcceleration = movement?.maxAcceleration ?? 0
        out.maxDeceleration = movement?.maxDeceleration ?? 0
        return out
    }

    private func decode(_ b: sge_body_state, into body: inout PhysicsBodyComponent) {
        body.position = SIMD3<Double>(b.position.0, b.position.1, b.position.2)
        body.linearVelocity = SIMD3<Double>(b.linearVelocity.0, b.linearVelocity.1, b.linearVelocity.2)
        body.rotation = quat(b.rotation)
    }

    private func decode(_ s: sge_controller_state, into c: inout CharacterControllerComponent) {
        c.groundNormal = SIMD3<Float>(s.groundNormal.0, s.groundNormal.1, s.groundNormal.2)
        c.groundTriangleIndex = Int(s.groundTriangleIndex)
        c.sideContactNormal = SIMD3<Float>(s.sideContactNormal.0, s.sideContactNormal.1, s.sideContactNormal.2)
        c.sideContactFrames = Int(s.sideContactFrames)
        let m = Int(s.manifoldCount)
        var tris = [Int](), nrms = [SIMD3<Float>]()
        withUnsafeBytes(of: s.manifoldTriangles) { raw in let t = raw.bindMemory(to: Int32.self); for k in 0..<m { tris.append(Int(t[k])) } }
        withUnsafeBytes(of: s.manifoldNormals) { raw in let f = raw.bindMemory(to: Float.self); for k in 0..<m { nrms.append(SIMD3<Float>(f[3 * k], f[3 * k + 1], f[3 * k + 2])) } }
        c.contactManifoldTriangles = tris; c.contactManifoldNormals = nrms; c.contactManifoldFrames = Int(s.manifoldFrames)
        c.groundTransitionFrames = Int(s.groundTransitionFrames)
        c.grounded = s.flags & UInt32(SGE_CTRL_GROUNDED) != 0
        c.groundedNear = s.flags & UInt32(SGE_CTRL_GROUNDED_NEAR) != 0
        c.groundSliding = s.flags & UInt32(SGE_CTRL_GROUND_SLIDING) != 0
        c.groundDistance = s.groundDistance
    }

    private func decode(_ l: sge_locomotion_state, into c: inout LocomotionProfileComponent) {
        c.idleTime = l.time.0; c.walkTime = l.time.1; c.runTime = l.time.2; c.fallTime = l.time.3
        c.blendT = l.blendT; c.idleInertia = l.idleInertia
        c.fromState = LocomotionState(rawValue: Int(l.fromState)) ?? .idle
        c.state = LocomotionState(rawValue: Int(l.state)) ?? .idle
        c.isBlending = l.flags & UInt32(SGE_LOCO_IS_BLENDING) != 0
    }

    private func decode(_ a: sge_action_state, into c: inout ActionAnimationComponent) {
        c.time = a.time; c.weight = a.weight
        c.active = a.flags & UInt32(SGE_ACTION_ACTIVE) != 0
        c.exiting = a.flags & UInt32(SGE_ACTION_EXITING) != 0
    }

    private func row(_ p: MotionProfile) -> Int32 { profileIndex[p.name] ?? 0 }   // MotionProfile is a value type: matched by name
    // SIMD3<Float> has a 16-byte stride in Swift; the ABI takes packed xyz
    func packed(_ v: [SIMD3<Float>]) -> [Float] { var out = [Float](); out.reserveCapacity(v.count * 3); for x in v { out.append(x.x); out.append(x.y); out.append(x.z) }; return out }
    private func tuple(_ q: simd_quatf) -> (Float, Float, Float, Float) { (q.imag.x, q.imag.y, q.imag.z, q.real) }
    private func quat(_ t: (Float, Float, Float, Float)) -> simd_quatf { simd_quatf(ix: t.0, iy: t.1, iz: t.2, r: t.3) }
}
