//  KinematicMoveStopSystem.swift, GravitySystem / PhysicsIntentSystem / LocomotionProfileSystem / ActionAnimationSystem /
//  AgentSeparationSystem stand-ins — same class names, same FixedStepSystem protocol (Systems.swift:15-17), so the fixed lists of
//  DemoScene.swift:56-75 stay as they are. NOT COMPILED HERE (see GPUCrowd.swift); C++ twins in ../sge_host.hpp.
//
//  Every system here is one sge_tick with its own stage bit. A scene that swaps in all of them may instead put ONE
//  `GPUCharacterStepSystem` (bottom of this file) where the first of them stood and drop the others: one tick with all stages is what
//  bench.py times.

import simd
import CSGE

func sgeTick(_ crowd: GPUCrowd, dt: Float, gravity: SIMD3<Float> = SIMD3<Float>(0, -98.0, 0), stages: UInt32) {
    var d = sge_tick_desc()
    d.dt = dt
    d.gravity = (gravity.x, gravity.y, gravity.z)
    d.stages = stages
    d.first = 0; d.count = 0                                   // the whole crowd
    crowd.check(sge_tick(crowd.ctx, &d))
}

/// Systems.swift:1136-1157 — `private` in the reference's file; a host that swaps this binding in names it here instead
public struct SideContactOnlyCachePolicy: ContactCachePolicy {
    public init() {}
    public mutating func decay(controller: inout CharacterControllerComponent) { var p = DefaultContactCachePolicy(); p.decay(controller: &controller) }
    public func cachedNormal(controller: CharacterControllerComponent, triangleIndex: Int) -> SIMD3<Float>? {
        DefaultContactCachePolicy().cachedNormal(controller: controller, triangleIndex: triangleIndex)
    }
    public mutating func record(controller: inout CharacterControllerComponent, triangleIndex: Int, normal: SIMD3<Float>, isSideContact: Bool) {
        guard isSideContact else { return }
        var p = DefaultContactCachePolicy(); p.record(controller: &controller, triangleIndex: triangleIndex, normal: normal, isSideContact: true)
    }
}

/// Systems.swift:1402-1415, :1823-1902
public final class KinematicMoveStopSystem: FixedStepSystem {
    private let crowd: GPUCrowd
    private let gravity: SIMD3<Float>
    private var query: CollisionQuery?
    /// kinematic platforms of this step (Systems.swift:1832-1835): entity, mesh positions, previous positionF
    public var platformEntities: [Entity] = []

    /// The policy runs inside the kernels, so it must be one of the two the reference defines (:1102-1157); any other
    /// ContactCachePolicy is host code that cannot be called per contact from the GPU and is refused, not silently dropped.
    public init(crowd: GPUCrowd, gravity: SIMD3<Float> = SIMD3<Float>(0, -98.0, 0), contactCachePolicy: ContactCachePolicy = DefaultContactCachePolicy()) {
        self.crowd = crowd; self.gravity = gravity
        if contactCachePolicy is DefaultContactCachePolicy { policyBit = 0 }
        else if contactCachePolicy is SideContactOnlyCachePolicy { policyBit = UInt32(SGE_STAGE_SIDE_CONTACT_CACHE) }
        else { preconditionFailure("KinematicMoveStopSystem on the GPU supports DefaultContactCachePolicy and SideContactOnlyCachePolicy only") }
    }
    private let policyBit: UInt32
    public func setQuery(_ query: CollisionQuery?) { self.query = query }   // kept for CollisionQueryRefreshSystem (:157-180); the world lives on the GPU

    public func fixedUpdate(world: World, dt: Float) {
        crowd.pushDirtyState(from: world)
        uploadPlatforms(world: world)
        sgeTick(crowd, dt: dt, gravity: gravity, stages: UInt32(SGE_STAGE_MOVE) | policyBit)
        crowd.beginPull(which: UInt32(SGE_STATE_BODIES) | UInt32(SGE_STATE_CONTROLLERS))   // writeBack, Systems.swift:1802-1821
        crowd.pullBack(into: world)
    }

    /// PlatformCarry.computeDelta's inputs (:644-732): per kinematic platform its world AABB (meshWorldAABB :627-642), its positionF -
    /// prevPositionF of this step, and whether the body is kinematic
    private func uploadPlatforms(world: World) {
        let pStore = world.store(PhysicsBodyComponent.self), mStore = world.store(StaticMeshComponent.self), tStore = world.store(TransformComponent.self)
        var out = [sge_platform_state]()
        for e in platformEntities.prefix(Int(SGE_MAX_PLATFORMS)) {
            guard let body = pStore[e], let mesh = mStore[e], let t = tStore[e] else { continue }
            var p = sge_platform_state()
            let positions = crowd.packed((mesh.collisionMesh ?? mesh.mesh).streams.positions)
            var m = t.modelMatrix
            var mn = (Float(0), Float(0), Float(0)), mx = mn
            let ok = withUnsafeBytes(of: &m) { mm in
                withUnsafeMutableBytes(of: &mn) { a in withUnsafeMutableBytes(of: &mx) { b in
                    sge_mesh_world_aabb(positions, Int32(positions.count / 3), mm.bindMemory(to: Float.self).baseAddress,
                                        a.bindMemory(to: Float.self).baseAddress, b.bindMemory(to: Float.self).baseAddress) == SGE_OK
                }}
            }
            p.aabbMin = mn; p.aabbMax = mx
            p.hasAABB = ok && !positions.isEmpty ? 1 : 0
            let d = body.positionF - body.prevPositionF
            p.delta = (d.x, d.y, d.z)
            p.kinematic = body.bodyType == .kinematic ? 1 : 0
            out.append(p)
        }
        crowd.check(sge_platforms_upload(crowd.ctx, out, Int32(out.count)))
    }
}

/// Systems.swift:596-620
public final class GravitySystem: FixedStepSystem {
    private let crowd: GPUCrowd
    public var gravity: SIMD3<Float>
    public init(crowd: GPUCrowd, gravity: SIMD3<Float> = SIMD3<Float>(0, -98.0, 0)) { self.crowd = crowd; self.gravity = gravity }
    public func fixedUpdate(world: World, dt: Float) { sgeTick(crowd, dt: dt, gravity: gravity, stages: UInt32(SGE_STAGE_GRAVITY)) }
}

/// Systems.swift:205-250 (controller branch)
public final class PhysicsIntentSystem: FixedStepSystem {
    private let crowd: GPUCrowd
    public init(crowd: GPUCrowd) { self.crowd = crowd }
    public func fixedUpdate(world: World, dt: Float) {
        crowd.pushDirtyState(from: world)
        sgeTick(crowd, dt: dt, stages: UInt32(SGE_STAGE_INTENT))
    }
}

/// Systems.swift:279-407
public final class LocomotionProfileSystem: FixedStepSystem {
    private let crowd: GPUCrowd
    public init(crowd: GPUCrowd) { self.crowd = crowd }
    public func fixedUpdate(world: World, dt: Float) { sgeTick(crowd, dt: dt, stages: UInt32(SGE_STAGE_LOCOMOTION)) }
}

/// Systems.swift:475-517
public final class ActionAnimationSystem: FixedStepSystem {
    private let crowd: GPUCrowd
    public init(crowd: GPUCrowd) { self.crowd = crowd }
    public func fixedUpdate(world: World, dt: Float) { sgeTick(crowd, dt: dt, stages: UInt32(SGE_STAGE_ACTION)) }
}

/// Systems.swift:1906-2210 — resolved in entity-id order (the order of GPUCrowd.entities); any crowd size (sge_amd.h, SGE_STAGE_SEPARATION)
public final class AgentSeparationSystem: FixedStepSystem {
    private let crowd: GPUCrowd
    public var iterations: Int { didSet { push() } }
    public var separationMargin: Float { didSet { push() } }
    public var heightMargin: Float { didSet { push() } }
    public init(crowd: GPUCrowd, iterations: Int = 2, separationMargin: Float = 0.2, heightMargin: Float = 0.1) {
        self.crowd = crowd; self.iterations = max(1, iterations); self.separationMargin = separationMargin; self.heightMargin = heightMargin
        push()
    }
    private func push() { crowd.check(sge_separation_params(crowd.ctx, Int32(max(1, iterations)), separationMargin, heightMargin)) }
    public func setQuery(_ query: CollisionQuery?) {}
    public func fixedUpdate(world: World, dt: Float) {
        sgeTick(crowd, dt: dt, stages: UInt32(SGE_STAGE_SEPARATION))
        crowd.beginPull(which: UInt32(SGE_STATE_BODIES))
        crowd.pullBack(into: world)
    }
}

/// All character stages of one fixed step as ONE tick, in the order of DemoScene.swift:57-75 (intent, gravity, move, [separation],
/// locomotion, action, pose, write-back, skin): put it where physicsIntentSystem stands and remove the systems above.
public final class GPUCharacterStepSystem: FixedStepSystem {
    private let crowd: GPUCrowd
    public var gravity = SIMD3<Float>(0, -98.0, 0)
    public var separation = false
    public var skin = true                     // RTSkinningEncoder work of the frame folded into the step (SGE_STAGE_SKIN)
    /// false: the World holds step n when fixedUpdate returns (the host waits for move(n) + pose(n) + a 352 B/character copy, never for
    /// skin(n)). true: the World runs one step behind and nothing is waited for — tick n + 1 is enqueued while pull n is on its way.
    public var lagged = false
    public init(crowd: GPUCrowd) {
        self.crowd = crowd
        crowd.check(sge_context_set_option(crowd.ctx, Int32(SGE_OPT_OVERLAP_SKIN), 1))     // skin(n) beside move(n+1) + pose(n+1)
    }
    public func fixedUpdate(world: World, dt: Float) {
        crowd.pushDirtyState(from: world)
        var stages = UInt32(SGE_STAGE_ALL_FIXED)
        if skin { stages |= UInt32(SGE_STAGE_SKIN) }
        if separation { stages |= UInt32(SGE_STAGE_SEPARATION) }
        sgeTick(crowd, dt: dt, gravity: gravity, stages: stages)
        if lagged { crowd.pullBack(into: world) }     // the previous step's pull, long landed
        crowd.beginPull()
        if !lagged { crowd.pullBack(into: world) }
    }
}
