//  RTSkinningEncoder.swift — Game/RTSkinningEncoder.swift:14-56 with device pointers where Metal had MTLBuffers.
//  NOT COMPILED HERE (see GPUCrowd.swift). C++ twin: sge::RTSkinningEncoder in ../sge_host.hpp.
//
//  `encode` enqueues on the context's stream and returns (the command-buffer semantics of the original: work completes before
//  whatever the caller enqueues on the same stream afterwards, e.g. the acceleration-structure refit, RayTracingScene.swift:35-43).
//  The whole job list is ONE launch; a crowd of clones does not need jobs at all (SGE_STAGE_SKIN of sge_tick).

import CSGE

/// RTGeometryCache.swift:43-52, buffers as device addresses (hipMalloc'ed by the caller, or the context's own: see below)
public struct RTSkinningJob {
    public let sourcePositions, sourceNormals, sourceTangents, sourceBoneIndices, sourceBoneWeights, paletteBuffer: UnsafeRawPointer
    public let paletteCount: Int          // bones in paletteBuffer (skeleton.boneCount; the Metal kernel indexes it unchecked)
    public let vertexCount: Int
    public let dstBaseVertex: Int
    public var sourceStride16 = false     // true: positions / normals are 16-byte `float3` as Metal lays them out
}

public final class RTSkinningEncoder {
    private let crowd: GPUCrowd
    /// failable like `init?(device: MTLDevice)`: nil when the library has no usable GPU behind the context
    public init?(crowd: GPUCrowd?) {
        guard let crowd = crowd else { return nil }
        self.crowd = crowd
    }

    public func encode(outputBuffer: UnsafeMutableRawPointer, outputNormalBuffer: UnsafeMutableRawPointer, outputTangentBuffer: UnsafeMutableRawPointer,
                       outputStride16: Bool = false, jobs: [RTSkinningJob]) {
        guard !jobs.isEmpty else { return }                        // :32-35
        var descs = jobs.map { j -> sge_skinning_job in
            sge_skinning_job(d_sourcePositions: j.sourcePositions, d_sourceNormals: j.sourceNormals, d_sourceTangents: j.sourceTangents,
                             d_sourceBoneIndices: j.sourceBoneIndices, d_sourceBoneWeights: j.sourceBoneWeights, d_palette: j.paletteBuffer,
                             paletteCount: Int32(j.paletteCount), vertexCount: Int32(j.vertexCount), dstBaseVertex: Int32(j.dstBaseVertex),
                             sourceLayout: j.sourceStride16 ? Int32(SGE_LAYOUT_PADDED16) : Int32(SGE_LAYOUT_PACKED))
        }
        crowd.check(sge_skinning_encode(crowd.ctx, outputBuffer, outputNormalBuffer, outputTangentBuffer,
                                        outputStride16 ? Int32(SGE_LAYOUT_PADDED16) : Int32(SGE_LAYOUT_PACKED), &descs, Int32(descs.count)))
    }

    /// What enqueueing behind `encode` on the same MTLCommandBuffer gives the reference (RayTracingScene.swift:35-43): `consumer` (a
    /// hipStream_t, nil = the context's stream) is ordered behind every skin launch so far. Needed under SGE_OPT_OVERLAP_SKIN.
    public func waitForSkinning(consumer: UnsafeMutableRawPointer? = nil) {
        crowd.check(sge_skin_wait(crowd.ctx, consumer))
    }

    /// ... and the reverse: what `consumer` holds so far completes before the next skin launch overwrites the streams.
    public func skinningConsumed(consumer: UnsafeMutableRawPointer? = nil) {
        crowd.check(sge_skin_consumed(crowd.ctx, consumer))
    }

    /// RTGeometryCache.makeSkinningJob (:492-576) for character `index` of the crowd over the context's own buffers: the source streams
    /// uploaded once, the palette the pose stage wrote (no per-frame makeBuffer + memcpy, :556-566), destination = running vertex offset.
    /// Build the jobs every frame, as the reference does: under SGE_OPT_OVERLAP_SKIN the palettes alternate between two buffers.
    public func makeSkinningJob(characterIndex index: Int, vertexCount: Int) -> RTSkinningJob? {
        var pal: UnsafeMutableRawPointer?, op: UnsafeMutableRawPointer?, on: UnsafeMutableRawPointer?, ot: UnsafeMutableRawPointer?
        var sp: UnsafeMutableRawPointer?, sn: UnsafeMutableRawPointer?, st: UnsafeMutableRawPointer?, si: UnsafeMutableRawPointer?, sw: UnsafeMutableRawPointer?
        guard sge_crowd_buffers(crowd.ctx, &pal, &op, &on, &ot) == SGE_OK, sge_skinned_mesh_buffers(crowd.ctx, &sp, &sn, &st, &si, &sw) == SGE_OK,
              let p = pal, let a = sp, let b = sn, let c = st, let d = si, let e = sw else { return nil }
        let bones = crowd.paletteCount
        return RTSkinningJob(sourcePositions: a, sourceNormals: b, sourceTangents: c, sourceBoneIndices: d, sourceBoneWeights: e,
                             paletteBuffer: UnsafeRawPointer(p.advanced(by: index * bones * 64)), paletteCount: bones,
                             vertexCount: vertexCount, dstBaseVertex: index * vertexCount)
    }
}
