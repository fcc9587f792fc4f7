// ORACLE — TEST INFRASTRUCTURE ONLY (see sge_oracle_math.h).
//
// CPU float32 restatement of the skinning path:
//   skinningKernel                 Game/RayTracing.metalinc:737-776
//   RTSkinningEncoder.encode       Game/RTSkinningEncoder.swift:27-56 (one "dispatch" per character)
//   MeshTangents.compute           Game/MeshTangents.swift:10-83
//   SkeletonLoader.buildSkeleton   Game/SkeletonLoader.swift:28-87
// The reference has no CPU skinning path (RTGeometryCache.swift:281,425-429
// uploads zeros without a GPU); this is the Metal kernel body as a loop.
// Metal is built with MTL_FAST_MATH, so its normalize() is not IEEE-exact;
// here normalize(x) = x * (1/sqrt(dot(x,x))).
#include "sge_oracle.h"

namespace sgeo {

void skinning_kernel(int vertexCount, const float* pos, const float* nrm, const float* tan,
                     const uint16_t* idx, const float* wts, const M4* palette,
                     float* outPos, float* outNrm, float* outTan, int dstBaseVertex) {
    for (int gid = 0; gid < vertexCount; ++gid) {
        V4 p = V4{pos[gid * 3], pos[gid * 3 + 1], pos[gid * 3 + 2], 1.0f};
        V4 n = V4{nrm[gid * 3], nrm[gid * 3 + 1], nrm[gid * 3 + 2], 0.0f};
        V4 t = V4{tan[gid * 4], tan[gid * 4 + 1], tan[gid * 4 + 2], 0.0f};
        float tw = tan[gid * 4 + 3];
        const uint16_t* ix = &idx[gid * 4];
        const float* w = &wts[gid * 4];
        V3 acc = V3{0, 0, 0}, nAcc = V3{0, 0, 0}, tAcc = V3{0, 0, 0};
        for (int j = 0; j < 4; ++j) {
            if (w[j] > 0.0f) { V4 r = mul(palette[ix[j]], p); acc += V3{r.x, r.y, r.z} * w[j]; }
        }
        for (int j = 0; j < 4; ++j) {
            if (w[j] > 0.0f) { V4 r = mul(palette[ix[j]], n); nAcc += V3{r.x, r.y, r.z} * w[j]; }
        }
        for (int j = 0; j < 4; ++j) {
            if (w[j] > 0.0f) { V4 r = mul(palette[ix[j]], t); tAcc += V3{r.x, r.y, r.z} * w[j]; }
        }
        size_t o = (size_t)dstBaseVertex + gid;
        outPos[o * 3] = acc.x; outPos[o * 3 + 1] = acc.y; outPos[o * 3 + 2] = acc.z;
        V3 nn = normalize(nAcc);
        outNrm[o * 3] = nn.x; outNrm[o * 3 + 1] = nn.y; outNrm[o * 3 + 2] = nn.z;
        V3 tn = normalize(tAcc);
        outTan[o * 4] = tn.x; outTan[o * 4 + 1] = tn.y; outTan[o * 4 + 2] = tn.z; outTan[o * 4 + 3] = tw;
    }
}

// dstBaseVertex = running vertex offset across skinned items (RTGeometryCache.swift:266-315)
void skin_characters(World& w, int first, int count) {
    const int V = w.mesh.vertexCount, B = w.skeleton.boneCount;
    for (int e = first; e < first + count; ++e) {
        skinning_kernel(V, w.mesh.positions.data(), w.mesh.normals.data(), w.mesh.tangents.data(),
                        w.mesh.indices.data(), w.mesh.weights.data(), &w.palette[(size_t)e * B],
                        w.outPositions.data(), w.outNormals.data(), w.outTangents.data(), e * V);
    }
}

} // namespace sgeo

using namespace sgeo;

extern "C" {

// MeshTangents.swift:10-83
int sgeo_mesh_tangents_compute(int32_t vCount, const float* positions, const float* normals, const float* uvs,
                               const uint16_t* indices16, const uint32_t* indices32, int32_t indexCount,
                               float* tangents) {
    if (vCount <= 0) return 0;
    std::vector<V3> tan1(vCount, V3{0, 0, 0}), tan2(vCount, V3{0, 0, 0});
    auto P = [&](int i) { return V3{positions[i * 3], positions[i * 3 + 1], positions[i * 3 + 2]}; };
    auto addTriangle = [&](int i0, int i1, int i2) {
        V3 p0 = P(i0), p1 = P(i1), p2 = P(i2);
        float u0x = uvs[i0 * 2], u0y = uvs[i0 * 2 + 1];
        float d1x = uvs[i1 * 2] - u0x, d1y = uvs[i1 * 2 + 1] - u0y;
        float d2x = uvs[i2 * 2] - u0x, d2y = uvs[i2 * 2 + 1] - u0y;
        V3 dp1 = p1 - p0, dp2 = p2 - p0;
        float denom = d1x * d2y - d1y * d2x;
        if (fabsf(denom) < 1e-6f) return;
        float r = 1.0f / denom;
        V3 t = (dp1 * d2y - dp2 * d1y) * r;
        V3 b = (dp2 * d1x - dp1 * d2x) * r;
        tan1[i0] += t; tan1[i1] += t; tan1[i2] += t;
        tan2[i0] += b; tan2[i1] += b; tan2[i2] += b;
    };
    for (int idx = 0; idx + 2 < indexCount; idx += 3) {
        if (indices16) addTriangle(indices16[idx], indices16[idx + 1], indices16[idx + 2]);
        else if (indices32) addTriangle((int)indices32[idx], (int)indices32[idx + 1], (int)indices32[idx + 2]);
    }
    for (int i = 0; i < vCount; ++i) {
        V3 n = normalize(V3{normals[i * 3], normals[i * 3 + 1], normals[i * 3 + 2]});
        V3 t = tan1[i];
        if (length_squared(t) < 1e-8f) {
            tangents[i * 4] = 1; tangents[i * 4 + 1] = 0; tangents[i * 4 + 2] = 0; tangents[i * 4 + 3] = 1;
            continue;
        }
        t = normalize(t - n * dot(n, t));
        V3 b = tan2[i];
        float w = dot(cross(n, t), b) < 0.0f ? -1.0f : 1.0f;
        tangents[i * 4] = t.x; tangents[i * 4 + 1] = t.y; tangents[i * 4 + 2] = t.z; tangents[i * 4 + 3] = w;
    }
    return 0;
}

// SkeletonLoader.swift:28-87 + Skeleton.swift:153-156
int sgeo_skeleton_build(int32_t boneCount, const int32_t* parent, const float* rawTranslations,
                        const float* preRotationDegrees, const float rootFixDegrees[3], float unitScale,
                        int zeroRoot, float* restTranslation, float* bindLocal, float* invBindModel,
                        float* rootRotationFix) {
    M4 rootFix = rotationXYZDegrees(V3{rootFixDegrees[0], rootFixDegrees[1], rootFixDegrees[2]});
    std::vector<M4> local(boneCount), model(boneCount);
    for (int i = 0; i < boneCount; ++i) {
        V3 raw = (zeroRoot && i == 0) ? V3{0, 0, 0}
                                      : V3{rawTranslations[i * 3], rawTranslations[i * 3 + 1], rawTranslations[i * 3 + 2]};
        V3 t = raw * unitScale;
        restTranslation[i * 3] = t.x; restTranslation[i * 3 + 1] = t.y; restTranslation[i * 3 + 2] = t.z;
        V3 pre = V3{preRotationDegrees[i * 3], preRotationDegrees[i * 3 + 1], preRotationDegrees[i * 3 + 2]};
        M4 rot = mul(rotationXYZDegrees(pre), rotationXYZDegrees(V3{0, 0, 0}));
        if (i == 0) rot = mul(rootFix, rot);
        local[i] = mul(matrix4x4_translation(t.x, t.y, t.z), rot);
    }
    for (int i = 0; i < boneCount; ++i) {
        int p = parent[i];
        model[i] = p < 0 ? local[i] : mul(model[p], local[i]);
    }
    for (int i = 0; i < boneCount; ++i) {
        M4 inv = inverse(model[i]);
        for (int k = 0; k < 16; ++k) {
            bindLocal[i * 16 + k] = (&local[i].c[0].x)[k];
            invBindModel[i * 16 + k] = (&inv.c[0].x)[k];
        }
    }
    for (int k = 0; k < 16; ++k) rootRotationFix[k] = (&rootFix.c[0].x)[k];
    return 0;
}

} // extern "C"
