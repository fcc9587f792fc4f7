//  CollisionQuery.swift — the reference's CollisionQuery facade (Game/CollisionQuery.swift:54-160) over the GPU collision world.
//  NOT COMPILED HERE (see GPUCrowd.swift). Same initialiser, same seven query methods, same hit structs, `nil` / empty array for
//  "no hit". C++ twin: sge::CollisionQuery in ../sge_host.hpp (run on the GPU by tests/cpp/host_mirror_smoke.cpp).
//
//  The facade answers single queries synchronously through the batch entry points (one query = one small launch + a wait): fine for
//  gameplay probes, and what the parity tests use. The hot callers — KinematicMoveStopSystem and AgentSeparationSystem — do not go
//  through it: their queries run inside sge_tick.

import simd
import CSGE

public struct RaycastHit { public var distance: Float; public var position, normal: SIMD3<Float>; public var triangleIndex: Int; public var material: SurfaceMaterial }
public struct CapsuleCastHit { public var toi: Float; public var position, normal, triangleNormal: SIMD3<Float>; public var triangleIndex: Int; public var material: SurfaceMaterial }
public struct CapsuleOverlapHit { public var depth: Float; public var position, normal, triangleNormal: SIMD3<Float>; public var triangleIndex: Int; public var material: SurfaceMaterial }

public final class CollisionQuery {
    private let crowd: GPUCrowd
    private var staticEntities: [Entity] = []      // order of the last rebuild = entity index of sge_collision_update_transforms
    private var dynamicEntities: [Entity] = []
    public private(set) var stats = CollisionQueryStats()

    /// CollisionQuery.init(world:activeEntityIDs:) :57-59 -> TriangleMeshSet.rebuild for both sets (:331-417, partitionEntities :886-900)
    public init(crowd: GPUCrowd, world: World, activeEntityIDs: Set<UInt32>? = nil) {
        self.crowd = crowd
        rebuild(world: world, activeEntityIDs: activeEntityIDs)
    }

    public func resetStats() { stats = CollisionQueryStats(); var s = sge_move_stats(); _ = sge_move_stats_read(crowd.ctx, &s, 1) }

    /// filteredEntities (:10-26) + partition by body type: entities whose PhysicsBodyComponent is not .static go to the dynamic set
    public func rebuild(world: World, activeEntityIDs: Set<UInt32>? = nil) {
        let mStore = world.store(StaticMeshComponent.self), pStore = world.store(PhysicsBodyComponent.self)
        let all = world.query(TransformComponent.self, StaticMeshComponent.self)
            .filter { e in (activeEntityIDs.map { $0.contains(e.id) } ?? true) && (mStore[e]?.collides ?? false) }
            .sorted { $0.id < $1.id }
        staticEntities = all.filter { (pStore[$0]?.bodyType ?? .static) == .static }
        dynamicEntities = all.filter { (pStore[$0]?.bodyType ?? .static) != .static }
        upload(set: staticEntities, world: world, rebuild: sge_collision_rebuild_static)
        upload(set: dynamicEntities, world: world, rebuild: sge_collision_rebuild_dynamic)
    }

    private func upload(set: [Entity], world: World, rebuild: (OpaquePointer?, UnsafePointer<sge_static_mesh_entity>?, Int32) -> Int32) {
        let mStore = world.store(StaticMeshComponent.self), tStore = world.store(TransformComponent.self)
        var descs = [sge_static_mesh_entity](repeating: sge_static_mesh_entity(), count: set.count)
        var keepPositions = [[Float]](), keepIndices = [[UInt32]](), keepMaterials = [[sge_surface_material]]()
        for e in set {
            let c = mStore[e]!, mesh = c.collisionMesh ?? c.mesh                 // `collisionMesh ?? mesh`, :344
            keepPositions.append(crowd.packed(mesh.streams.positions))
            keepIndices.append(mesh.indices32 ?? (mesh.indices16 ?? []).map { UInt32($0) })
            keepMaterials.append((c.triangleMaterials ?? []).map { sge_surface_material(muS: $0.muS, muK: $0.muK, flattenGround: $0.flattenGround ? 1 : 0) })
        }
        withExtendedLifetime((keepPositions, keepIndices, keepMaterials)) {
            for (k, e) in set.enumerated() {
                let c = mStore[e]!
                keepPositions[k].withUnsafeBufferPointer { descs[k].positions = $0.baseAddress }
                descs[k].vertexCount = Int32(keepPositions[k].count / 3)
                keepIndices[k].withUnsafeBufferPointer { descs[k].indices = $0.baseAddress }
                descs[k].indexCount = Int32(keepIndices[k].count)
                var m = tStore[e]!.modelMatrix
                withUnsafeBytes(of: &m) { src in withUnsafeMutableBytes(of: &descs[k].modelMatrix) { $0.copyMemory(from: src) } }
                descs[k].material = sge_surface_material(muS: c.material.muS, muK: c.material.muK, flattenGround: c.material.flattenGround ? 1 : 0)
                if !keepMaterials[k].isEmpty {
                    keepMaterials[k].withUnsafeBufferPointer { descs[k].triangleMaterials = $0.baseAddress }
                    descs[k].triangleMaterialCount = Int32(keepMaterials[k].count)
                }
                descs[k].collisionLayer = c.collisionLayer
            }
            crowd.check(rebuild(crowd.ctx, descs, Int32(descs.count)))
        }
    }

    /// updateStaticTransforms / updateDynamicTransforms (:67-83): new model matrices for some entities of the last rebuild -> BVH.refit
    public func updateStaticTransforms(world: World, entities: [Entity], activeEntityIDs: Set<UInt32>? = nil) {
        update(set: Int32(SGE_SET_STATIC), table: staticEntities, world: world, entities: entities, activeEntityIDs: activeEntityIDs)
    }
    public func updateDynamicTransforms(world: World, entities: [Entity], activeEntityIDs: Set<UInt32>? = nil) {
        update(set: Int32(SGE_SET_DYNAMIC), table: dynamicEntities, world: world, entities: entities, activeEntityIDs: activeEntityIDs)
    }
    private func update(set: Int32, table: [Entity], world: World, entities: [Entity], activeEntityIDs: Set<UInt32>?) {
        let tStore = world.store(TransformComponent.self)
        var indices = [Int32](), matrices = [Float]()
        for e in entities where activeEntityIDs.map({ $0.contains(e.id) }) ?? true {
            guard let k = table.firstIndex(of: e), let t = tStore[e] else { continue }
            indices.append(Int32(k))
            var m = t.modelMatrix
            withUnsafeBytes(of: &m) { matrices.append(contentsOf: $0.bindMemory(to: Float.self)) }
        }
        guard !indices.isEmpty else { return }
        crowd.check(sge_collision_update_transforms(crowd.ctx, set, indices, matrices, Int32(indices.count)))
    }

    // MARK: the seven queries (:85-159) ----------------------------------------------------------------------------------------

    public func raycast(origin: SIMD3<Float>, direction: SIMD3<Float>, maxDistance: Float, mask: UInt32 = CollisionLayer.all) -> RaycastHit? {
        var q = sge_ray_query(origin: (origin.x, origin.y, origin.z), direction: (direction.x, direction.y, direction.z), maxDistance: maxDistance, mask: mask)
        var h = sge_raycast_hit()
        crowd.check(sge_raycast_batch(crowd.ctx, &q, 1, &h))
        guard h.hit != 0 else { return nil }
        return RaycastHit(distance: h.distance, position: v(h.position), normal: v(h.normal), triangleIndex: Int(h.triangleIndex), material: material(h.material))
    }

    public func capsuleCast(from: SIMD3<Float>, delta: SIMD3<Float>, radius: Float, halfHeight: Float, mask: UInt32 = CollisionLayer.all) -> CapsuleCastHit? {
        cast(from, delta, radius, halfHeight, 0, mask, UInt32(SGE_CAST))
    }
    public func capsuleCastBlocking(from: SIMD3<Float>, delta: SIMD3<Float>, radius: Float, halfHeight: Float, mask: UInt32 = CollisionLayer.all) -> CapsuleCastHit? {
        cast(from, delta, radius, halfHeight, 0, mask, UInt32(SGE_CAST_BLOCKING))
    }
    public func capsuleCastGround(from: SIMD3<Float>, delta: SIMD3<Float>, radius: Float, halfHeight: Float, minNormalY: Float, mask: UInt32 = CollisionLayer.all) -> CapsuleCastHit? {
        cast(from, delta, radius, halfHeight, minNormalY, mask, UInt32(SGE_CAST_GROUND))
    }

    public func capsuleOverlap(from: SIMD3<Float>, radius: Float, halfHeight: Float, mask: UInt32 = CollisionLayer.all) -> CapsuleOverlapHit? {
        var q = query(from, .zero, radius, halfHeight, 0, mask, UInt32(SGE_CAST))
        var h = sge_capsule_overlap_hit()
        var found: Int32 = 0
        crowd.check(sge_capsule_overlap_batch(crowd.ctx, &q, 1, &h, &found))
        return found != 0 ? overlapHit(h) : nil
    }

    public func capsuleOverlapAll(from: SIMD3<Float>, radius: Float, halfHeight: Float, maxHits: Int = 8, mask: UInt32 = CollisionLayer.all) -> [CapsuleOverlapHit] {
        let cap = Int32(min(max(1, maxHits), Int(SGE_MAX_OVERLAP_HITS)))             // max(1, maxHits), :157
        var q = query(from, .zero, radius, halfHeight, 0, mask, UInt32(SGE_CAST))
        var hits = [sge_capsule_overlap_hit](repeating: sge_capsule_overlap_hit(), count: Int(cap))
        var count: Int32 = 0
        crowd.check(sge_capsule_overlap_all_batch(crowd.ctx, &q, 1, cap, &hits, &count))
        return hits.prefix(Int(count)).map(overlapHit)
    }

    // MARK: helpers ------------------------------------------------------------------------------------------------------------

    private func cast(_ from: SIMD3<Float>, _ delta: SIMD3<Float>, _ r: Float, _ hh: Float, _ minY: Float, _ mask: UInt32, _ mode: UInt32) -> CapsuleCastHit? {
        var q = query(from, delta, r, hh, minY, mask, mode)
        var h = sge_capsule_cast_hit()
        crowd.check(sge_capsule_cast_batch(crowd.ctx, &q, 1, &h))
        guard h.hit != 0 else { return nil }
        return CapsuleCastHit(toi: h.toi, position: v(h.position), normal: v(h.normal), triangleNormal: v(h.triangleNormal),
                              triangleIndex: Int(h.triangleIndex), material: material(h.material))
    }
    private func query(_ from: SIMD3<Float>, _ delta: SIMD3<Float>, _ r: Float, _ hh: Float, _ minY: Float, _ mask: UInt32, _ mode: UInt32) -> sge_capsule_query {
        sge_capsule_query(from: (from.x, from.y, from.z), delta: (delta.x, delta.y, delta.z), radius: r, halfHeight: hh, minNormalY: minY, mask: mask, mode: mode)
    }
    private func overlapHit(_ h: sge_capsule_overlap_hit) -> CapsuleOverlapHit {
        CapsuleOverlapHit(depth: h.depth, position: v(h.position), normal: v(h.normal), triangleNormal: v(h.triangleNormal),
                          triangleIndex: Int(h.triangleIndex), material: material(h.material))
    }
    private func v(_ t: (Float, Float, Float)) -> SIMD3<Float> { SIMD3<Float>(t.0, t.1, t.2) }
    private func material(_ m: sge_surface_material) -> SurfaceMaterial { SurfaceMaterial(muS: m.muS, muK: m.muK, flattenGround: m.flattenGround != 0) }
}
