//  PoseStackSystem.swift — Game/ProceduralPoseSystem.swift:10-406 as one stage of sge_tick. NOT COMPILED HERE (see GPUCrowd.swift).
//  C++ twin: sge::PoseStackSystem in ../sge_host.hpp.

import simd
import CSGE

public final class PoseStackSystem: FixedStepSystem {
    private let crowd: GPUCrowd
    /// true: PoseComponent.palette / .phase are copied back into the World after the step (only if something on the CPU reads them;
    /// the skinning stage reads the palettes where they are, in HBM)
    public var writePalettesBack = false
    public init(crowd: GPUCrowd) { self.crowd = crowd }       // the reference's `public init()` plus the crowd handle

    public func fixedUpdate(world: World, dt: Float) {
        // the palette is model[i] * mesh.invBindModel[i] when the mesh carries invBindModel of matching count (Systems.swift:2519-2527):
        // that re-bind is part of sge_skinned_mesh_upload, so the palettes written here are the ones the skinning kernel uses
        sgeTick(crowd, dt: dt, stages: UInt32(SGE_STAGE_POSE) | UInt32(SGE_STAGE_WRITEBACK))
        if writePalettesBack { crowd.pullBack(into: world, palettes: true) }
    }
}
