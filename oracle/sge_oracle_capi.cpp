// ORACLE — TEST INFRASTRUCTURE ONLY (see sge_oracle_math.h).
//
// extern "C" surface of the CPU restatement, shaped like the product ABI
// (include/sge_amd.h) with an `sgeo_` prefix and host pointers everywhere, so a
// parity test drives both with the same arrays. Loaded with ctypes by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product.
#include <cstring>
#include <thread>
#include <mutex>
#include "sge_oracle.h"

using namespace sgeo;

struct sgeo_world {
    World w;
    QueryStats stats;
};

static M4 loadM4(const float* p) { M4 m; std::memcpy(&m, p, 64); return m; }

extern "C" {

sgeo_world* sgeo_world_create(void) { return new sgeo_world(); }
void sgeo_world_destroy(sgeo_world* h) { delete h; }

int sgeo_skeleton_upload(sgeo_world* h, const sge_skeleton_desc* d) {
    if (!h || !d || d->boneCount <= 0 || d->boneCount > SGE_MAX_BONES) return SGE_ERR_INVALID;
    Skeleton& s = h->w.skeleton;
    int B = d->boneCount;
    s.boneCount = B;
    s.parent.assign(d->parent, d->parent + B);
    s.bindLocal.resize(B); s.invBindModel.resize(B);
    s.restTranslation.resize(B); s.rawRestTranslation.resize(B); s.preRotationDegrees.resize(B);
    for (int i = 0; i < B; ++i) {
        s.bindLocal[i] = loadM4(d->bindLocal + i * 16);
        s.invBindModel[i] = loadM4(d->invBindModel + i * 16);
        s.restTranslation[i] = V3{d->restTranslation[i * 3], d->restTranslation[i * 3 + 1], d->restTranslation[i * 3 + 2]};
        s.rawRestTranslation[i] = V3{d->rawRestTranslation[i * 3], d->rawRestTranslation[i * 3 + 1], d->rawRestTranslation[i * 3 + 2]};
        s.preRotationDegrees[i] = V3{d->preRotationDegrees[i * 3], d->preRotationDegrees[i * 3 + 1], d->preRotationDegrees[i * 3 + 2]};
    }
    s.rootRotationFix = loadM4(d->rootRotationFix);
    s.unitScale = d->unitScale;
    s.pelvisIndex = d->pelvisIndex;
    s.leanIndex = d->leanIndex;
    return SGE_OK;
}

int sgeo_motion_profiles_upload(sgeo_world* h, const sge_motion_profile_desc* p, int32_t count) {
    if (!h || !p || count <= 0 || count > SGE_MAX_PROFILES || h->w.skeleton.boneCount == 0) return SGE_ERR_INVALID;
    int B = h->w.skeleton.boneCount;
    h->w.profiles.resize(count);
    for (int k = 0; k < count; ++k) {
        MotionProfile& m = h->w.profiles[k];
        m.order = p[k].order;
        m.cycleDurationRaw = p[k].cycleDuration;
        m.bonePresent.assign(p[k].bonePresent, p[k].bonePresent + B);
        m.coeffCount.assign(p[k].coeffCount, p[k].coeffCount + B * 6);
        m.coeffs.assign(p[k].coeffs, p[k].coeffs + (size_t)B * 6 * SGE_MAX_COEFFS);
    }
    return SGE_OK;
}

int sgeo_skinned_mesh_upload(sgeo_world* h, const sge_skinned_mesh_desc* d) {
    if (!h || !d || d->vertexCount <= 0) return SGE_ERR_INVALID;
    SkinnedMesh& m = h->w.mesh;
    int V = d->vertexCount;
    m.vertexCount = V;
    m.positions.assign(d->positions, d->positions + V * 3);
    m.normals.assign(d->normals, d->normals + V * 3);
    m.tangents.assign(d->tangents, d->tangents + V * 4);
    m.indices.assign(d->boneIndices, d->boneIndices + V * 4);
    m.weights.assign(d->boneWeights, d->boneWeights + V * 4);
    m.invBindModel.clear();
    if (d->invBindModel && d->invBindCount > 0)
        for (int i = 0; i < d->invBindCount; ++i) m.invBindModel.push_back(loadM4(d->invBindModel + i * 16));
    return SGE_OK;
}

int sgeo_collision_rebuild_static(sgeo_world* h, const sge_static_mesh_entity* ents, int32_t count) {
    if (!h || (count > 0 && !ents)) return SGE_ERR_INVALID;
    h->w.query.staticSet.rebuild(ents, count);
    return SGE_OK;
}

int sgeo_collision_rebuild_dynamic(sgeo_world* h, const sge_static_mesh_entity* ents, int32_t count) {
    if (!h || (count > 0 && !ents)) return SGE_ERR_INVALID;
    h->w.query.dynamicSet.rebuild(ents, count);
    return SGE_OK;
}

int sgeo_collision_update_transforms(sgeo_world* h, int32_t set, const int32_t* entities, const float* matrices, int32_t n) {
    if (!h || (set != SGE_SET_STATIC && set != SGE_SET_DYNAMIC) || n < 0 || (n > 0 && (!entities || !matrices))) return SGE_ERR_INVALID;
    (set == SGE_SET_STATIC ? h->w.query.staticSet : h->w.query.dynamicSet).updateTransforms(entities, matrices, n);
    return SGE_OK;
}

int sgeo_collision_counts_set(sgeo_world* h, int32_t set, int32_t* v, int32_t* t, int32_t* n) {
    const TriangleMeshSet& s = set == SGE_SET_DYNAMIC ? h->w.query.dynamicSet : h->w.query.staticSet;
    if (v) *v = (int32_t)s.positions.size();
    if (t) *t = (int32_t)s.triangleAABBs.size();
    if (n) *n = (int32_t)s.bvh.nodes.size();
    return SGE_OK;
}

int sgeo_collision_counts(sgeo_world* h, int32_t* v, int32_t* t, int32_t* n) { return sgeo_collision_counts_set(h, SGE_SET_STATIC, v, t, n); }

int sgeo_collision_copy_set(sgeo_world* h, int32_t set, float* positions, uint32_t* indices, float* aabbs, sge_bvh_node* nodes,
                            int32_t* triOrder, int32_t* triLeaf);
int sgeo_collision_copy(sgeo_world* h, float* positions, uint32_t* indices, float* aabbs, sge_bvh_node* nodes,
                        int32_t* triOrder, int32_t* triLeaf) {
    return sgeo_collision_copy_set(h, SGE_SET_STATIC, positions, indices, aabbs, nodes, triOrder, triLeaf);
}

int sgeo_raycast_batch(sgeo_world* h, const sge_ray_query* q, int32_t count, sge_raycast_hit* out) {
    for (int i = 0; i < count; ++i) {
        RaycastHit hit;
        bool got = h->w.query.raycast(V3{q[i].origin[0], q[i].origin[1], q[i].origin[2]},
                                      V3{q[i].direction[0], q[i].direction[1], q[i].direction[2]}, q[i].maxDistance, q[i].mask, hit);
        std::memset(&out[i], 0, sizeof(out[i]));
        out[i].hit = got ? 1 : 0;
        out[i].triangleIndex = -1;
        if (got) {
            out[i].distance = hit.distance;
            out[i].position[0] = hit.position.x; out[i].position[1] = hit.position.y; out[i].position[2] = hit.position.z;
            out[i].normal[0] = hit.normal.x; out[i].normal[1] = hit.normal.y; out[i].normal[2] = hit.normal.z;
            out[i].triangleIndex = hit.triangleIndex;
            out[i].material = hit.material;
        }
    }
    return SGE_OK;
}

// meshWorldAABB, Systems.swift:627-642
int sgeo_mesh_world_aabb(const float* positions, int32_t count, const float* m, float* outMin, float* outMax) {
    if (!positions || count <= 0 || !m || !outMin || !outMax) return SGE_ERR_INVALID;
    M4 model;
    for (int k = 0; k < 16; ++k) (&model.c[0].x)[k] = m[k];
    const float big = 3.402823466e+38f;
    V3 mn = V3{big, big, big}, mx = V3{-big, -big, -big};
    for (int i = 0; i < count; ++i) {
        V4 wp = mul(model, V4{positions[i * 3], positions[i * 3 + 1], positions[i * 3 + 2], 1});
        V3 v = V3{wp.x, wp.y, wp.z};
        mn = vmin(mn, v); mx = vmax(mx, v);
    }
    outMin[0] = mn.x; outMin[1] = mn.y; outMin[2] = mn.z;
    outMax[0] = mx.x; outMax[1] = mx.y; outMax[2] = mx.z;
    return SGE_OK;
}

int sgeo_platforms_upload(sgeo_world* h, const sge_platform_state* p, int32_t count) {
    if (!h || count < 0 || (count > 0 && !p)) return SGE_ERR_INVALID;
    h->w.platforms.assign(p, p + count);
    return SGE_OK;
}

int sgeo_collision_copy_set(sgeo_world* h, int32_t set, float* positions, uint32_t* indices, float* aabbs, sge_bvh_node* nodes,
                            int32_t* triOrder, int32_t* triLeaf) {
    const TriangleMeshSet& s = set == SGE_SET_DYNAMIC ? h->w.query.dynamicSet : h->w.query.staticSet;
    if (positions) std::memcpy(positions, s.positions.data(), s.positions.size() * 12);
    if (indices) std::memcpy(indices, s.indices.data(), s.indices.size() * 4);
    if (aabbs) std::memcpy(aabbs, s.triangleAABBs.data(), s.triangleAABBs.size() * 24);
    if (nodes)
        for (size_t i = 0; i < s.bvh.nodes.size(); ++i) {
            const BVHNode& b = s.bvh.nodes[i];
            nodes[i] = sge_bvh_node{{b.bounds.min.x, b.bounds.min.y, b.bounds.min.z},
                                    {b.bounds.max.x, b.bounds.max.y, b.bounds.max.z},
                                    b.left, b.right, b.start, b.count, b.parent};
        }
    if (triOrder) for (size_t i = 0; i < s.bvh.triOrder.size(); ++i) triOrder[i] = s.bvh.triOrder[i];
    if (triLeaf) for (size_t i = 0; i < s.bvh.triLeaf.size(); ++i) triLeaf[i] = s.bvh.triLeaf[i];
    return SGE_OK;
}

int sgeo_capsule_cast_batch(sgeo_world* h, const sge_capsule_query* q, int32_t count, sge_capsule_cast_hit* out) {
    for (int i = 0; i < count; ++i) {
        CapsuleCastHit hit;
        bool got = h->w.query.capsuleCastCombined(V3{q[i].from[0], q[i].from[1], q[i].from[2]},
                                                  V3{q[i].delta[0], q[i].delta[1], q[i].delta[2]}, q[i].radius,
                                                  q[i].halfHeight, q[i].mode == SGE_CAST_BLOCKING,
                                                  q[i].mode == SGE_CAST_GROUND, q[i].minNormalY, q[i].mask, hit);
        std::memset(&out[i], 0, sizeof(out[i]));
        out[i].hit = got ? 1 : 0;
        if (got) {
            out[i].toi = hit.toi;
            out[i].position[0] = hit.position.x; out[i].position[1] = hit.position.y; out[i].position[2] = hit.position.z;
            out[i].normal[0] = hit.normal.x; out[i].normal[1] = hit.normal.y; out[i].normal[2] = hit.normal.z;
            out[i].triangleNormal[0] = hit.triangleNormal.x; out[i].triangleNormal[1] = hit.triangleNormal.y; out[i].triangleNormal[2] = hit.triangleNormal.z;
            out[i].triangleIndex = hit.triangleIndex;
            out[i].material = hit.material;
        } else {
            out[i].triangleIndex = -1;
        }
    }
    QueryStats s = take_thread_stats();
    h->stats.candidates += s.candidates; h->stats.sweeps += s.sweeps; h->stats.iterations += s.iterations; h->stats.queries += s.queries;
    return SGE_OK;
}

int sgeo_capsule_overlap_all_batch(sgeo_world* h, const sge_capsule_query* q, int32_t count, int32_t maxHits,
                                   sge_capsule_overlap_hit* out, int32_t* outCounts) {
    if (maxHits < 1 || maxHits > SGE_MAX_OVERLAP_HITS) return SGE_ERR_INVALID;
    for (int i = 0; i < count; ++i) {
        CapsuleOverlapHit hits[SGE_MAX_OVERLAP_HITS];
        int n = h->w.query.capsuleOverlapAll(V3{q[i].from[0], q[i].from[1], q[i].from[2]}, q[i].radius, q[i].halfHeight,
                                             maxHits, q[i].mask, hits);
        outCounts[i] = n;
        for (int k = 0; k < maxHits; ++k) {
            sge_capsule_overlap_hit& o = out[(size_t)i * maxHits + k];
            std::memset(&o, 0, sizeof(o));
            if (k >= n) { o.triangleIndex = -1; continue; }
            o.depth = hits[k].depth;
            o.position[0] = hits[k].position.x; o.position[1] = hits[k].position.y; o.position[2] = hits[k].position.z;
            o.normal[0] = hits[k].normal.x; o.normal[1] = hits[k].normal.y; o.normal[2] = hits[k].normal.z;
            o.triangleNormal[0] = hits[k].triangleNormal.x; o.triangleNormal[1] = hits[k].triangleNormal.y; o.triangleNormal[2] = hits[k].triangleNormal.z;
            o.triangleIndex = hits[k].triangleIndex;
            o.material = hits[k].material;
        }
    }
    take_thread_stats();
    return SGE_OK;
}

int sgeo_capsule_overlap_batch(sgeo_world* h, const sge_capsule_query* q, int32_t count, sge_capsule_overlap_hit* out,
                               int32_t* outFound) {
    for (int i = 0; i < count; ++i) {
        CapsuleOverlapHit hit;
        bool got = h->w.query.capsuleOverlap(V3{q[i].from[0], q[i].from[1], q[i].from[2]}, q[i].radius, q[i].halfHeight, q[i].mask, hit);
        sge_capsule_overlap_hit& o = out[i];
        std::memset(&o, 0, sizeof(o));
        outFound[i] = got ? 1 : 0;
        if (!got) { o.triangleIndex = -1; continue; }
        o.depth = hit.depth;
        o.position[0] = hit.position.x; o.position[1] = hit.position.y; o.position[2] = hit.position.z;
        o.normal[0] = hit.normal.x; o.normal[1] = hit.normal.y; o.normal[2] = hit.normal.z;
        o.triangleNormal[0] = hit.triangleNormal.x; o.triangleNormal[1] = hit.triangleNormal.y; o.triangleNormal[2] = hit.triangleNormal.z;
        o.triangleIndex = hit.triangleIndex;
        o.material = hit.material;
    }
    return SGE_OK;
}

int sgeo_characters_resize(sgeo_world* h, int32_t n) {
    World& w = h->w;
    int B = w.skeleton.boneCount, V = w.mesh.vertexCount;
    w.bodies.assign(n, sge_body_state{});
    w.params.assign(n, sge_controller_params{});
    w.controllers.assign(n, sge_controller_state{});
    w.intents.assign(n, sge_move_intent{});
    w.locomotion.assign(n, sge_locomotion_state{});
    w.actions.assign(n, sge_action_state{});
    w.local.assign((size_t)n * B, m4_identity());
    w.model.assign((size_t)n * B, m4_identity());
    w.palette.assign((size_t)n * B, m4_identity());
    w.outPositions.assign((size_t)n * V * 3, 0.f);
    w.outNormals.assign((size_t)n * V * 3, 0.f);
    w.outTangents.assign((size_t)n * V * 4, 0.f);
    return SGE_OK;
}

#define COPY_IN(vec, ptr) if (ptr) std::memcpy(&h->w.vec[first], ptr, sizeof(h->w.vec[0]) * count)
#define COPY_OUT(vec, ptr) if (ptr) std::memcpy(ptr, &h->w.vec[first], sizeof(h->w.vec[0]) * count)

int sgeo_characters_upload(sgeo_world* h, int32_t first, int32_t count, const sge_body_state* b,
                           const sge_controller_params* p, const sge_controller_state* c, const sge_move_intent* i,
                           const sge_locomotion_state* l, const sge_action_state* a) {
    if (first < 0 || count < 0 || (size_t)(first + count) > h->w.bodies.size()) return SGE_ERR_INVALID;
    COPY_IN(bodies, b); COPY_IN(params, p); COPY_IN(controllers, c); COPY_IN(intents, i); COPY_IN(locomotion, l); COPY_IN(actions, a);
    return SGE_OK;
}
int sgeo_characters_download(sgeo_world* h, int32_t first, int32_t count, sge_body_state* b, sge_controller_params* p,
                             sge_controller_state* c, sge_move_intent* i, sge_locomotion_state* l, sge_action_state* a) {
    if (first < 0 || count < 0 || (size_t)(first + count) > h->w.bodies.size()) return SGE_ERR_INVALID;
    COPY_OUT(bodies, b); COPY_OUT(params, p); COPY_OUT(controllers, c); COPY_OUT(intents, i); COPY_OUT(locomotion, l); COPY_OUT(actions, a);
    return SGE_OK;
}

int sgeo_palettes_download(sgeo_world* h, int32_t first, int32_t count, float* palette, float* model, float* local) {
    int B = h->w.skeleton.boneCount;
    size_t off = (size_t)first * B, n = (size_t)count * B * 64;
    if (palette) std::memcpy(palette, &h->w.palette[off], n);
    if (model) std::memcpy(model, &h->w.model[off], n);
    if (local) std::memcpy(local, &h->w.local[off], n);
    return SGE_OK;
}

int sgeo_skinned_download(sgeo_world* h, int64_t firstVertex, int64_t count, float* pos, float* nrm, float* tan) {
    if (pos) std::memcpy(pos, &h->w.outPositions[firstVertex * 3], count * 12);
    if (nrm) std::memcpy(nrm, &h->w.outNormals[firstVertex * 3], count * 12);
    if (tan) std::memcpy(tan, &h->w.outTangents[firstVertex * 4], count * 16);
    return SGE_OK;
}

// Oracle-only: overwrite the skinned streams (tests feed both sides the same vertices before comparing what a ray sees).
int sgeo_skinned_upload(sgeo_world* h, int64_t firstVertex, int64_t count, const float* pos, const float* nrm, const float* tan) {
    if (!h || firstVertex < 0 || count < 0) return SGE_ERR_INVALID;
    const size_t total = h->w.bodies.size() * (size_t)h->w.mesh.vertexCount;
    if ((size_t)(firstVertex + count) > total) return SGE_ERR_INVALID;
    if (h->w.outPositions.size() < total * 3) { h->w.outPositions.resize(total * 3); h->w.outNormals.resize(total * 3); h->w.outTangents.resize(total * 4); }
    if (pos) std::memcpy(&h->w.outPositions[firstVertex * 3], pos, count * 12);
    if (nrm) std::memcpy(&h->w.outNormals[firstVertex * 3], nrm, count * 12);
    if (tan) std::memcpy(&h->w.outTangents[firstVertex * 4], tan, count * 16);
    return SGE_OK;
}

// RTSkinningEncoder.encode with host pointers (packed layouts)
int sgeo_skinning_encode(float* outPos, float* outNrm, float* outTan, const sge_skinning_job* jobs, int32_t jobCount) {
    for (int j = 0; j < jobCount; ++j) {
        const sge_skinning_job& J = jobs[j];
        skinning_kernel(J.vertexCount, (const float*)J.d_sourcePositions, (const float*)J.d_sourceNormals,
                        (const float*)J.d_sourceTangents, (const uint16_t*)J.d_sourceBoneIndices,
                        (const float*)J.d_sourceBoneWeights, (const M4*)J.d_palette, outPos, outNrm, outTan, J.dstBaseVertex);
    }
    return SGE_OK;
}

static void run_range(sgeo_world* h, const sge_tick_desc& d, int first, int count,
                      const std::vector<AgentSweepState>* agents, int selfOffset) {
    World& w = h->w;
    V3 g = V3{d.gravity[0], d.gravity[1], d.gravity[2]};
    // order of the fixed lists in DemoScene.swift:57-75
    if (d.stages & SGE_STAGE_INTENT) intent_fixed_update(w, first, count, d.dt);
    if (d.stages & SGE_STAGE_GRAVITY) gravity_fixed_update(w, first, count, d.dt, g);
    w.sideContactCacheOnly = (d.stages & SGE_STAGE_SIDE_CONTACT_CACHE) != 0;
    if (d.stages & SGE_STAGE_MOVE) kinematic_move_fixed_update(w, first, count, d.dt, g, agents, selfOffset);
    if (d.stages & SGE_STAGE_LOCOMOTION) locomotion_fixed_update(w, first, count);
    if (d.stages & SGE_STAGE_ACTION) action_fixed_update(w, first, count, d.dt);
    if (d.stages & SGE_STAGE_POSE) pose_fixed_update(w, first, count, d.dt);
    if (d.stages & SGE_STAGE_WRITEBACK) writeback_fixed_update(w, first, count);
    if (d.stages & SGE_STAGE_SKIN) skin_characters(w, first, count);
}

static int sgeo_tick_mt_stages(sgeo_world* h, const sge_tick_desc& dd, int first, int count, int threads,
                               const std::vector<AgentSweepState>* agentsPtr, int selfOffset);

// One fixed step on `threads` host threads (1 = the reference's main-actor execution).
int sgeo_tick_mt(sgeo_world* h, const sge_tick_desc* d, int32_t threads) {
    World& w = h->w;
    int first = d->first, count = d->count;
    if (count == 0) { first = 0; count = (int)w.bodies.size(); }
    if (first < 0 || (size_t)(first + count) > w.bodies.size()) return SGE_ERR_INVALID;
    // the same state errors as the product (sge_tick): stages that need assets which were never uploaded
    if (count > 0 && (d->stages & SGE_STAGE_POSE) && (w.skeleton.boneCount == 0 || w.profiles.empty())) return SGE_ERR_STATE;
    if (count > 0 && (d->stages & SGE_STAGE_SKIN) && w.mesh.vertexCount == 0) return SGE_ERR_STATE;
    std::vector<AgentSweepState> agents;
    int selfOffset = 0;
    const bool useAgents = (d->stages & SGE_STAGE_AGENTS) != 0;
    if (useAgents) {
        // the snapshot precedes gravity in neither order: collectAgentStates runs inside
        // KinematicMoveStopSystem, i.e. AFTER intent and gravity of this step.
        if (!w.agentsImported) {
            V3 g = V3{d->gravity[0], d->gravity[1], d->gravity[2]};
            sge_tick_desc pre = *d;
            if (d->stages & SGE_STAGE_INTENT) intent_fixed_update(w, 0, (int)w.bodies.size(), d->dt);
            if (d->stages & SGE_STAGE_GRAVITY) gravity_fixed_update(w, 0, (int)w.bodies.size(), d->dt, g);
            (void)pre;
        }
        collect_agent_states(w, agents, selfOffset);
    }
    sge_tick_desc dd = *d;
    if (useAgents && !w.agentsImported) dd.stages &= ~(SGE_STAGE_INTENT | SGE_STAGE_GRAVITY);
    if (d->stages & SGE_STAGE_SEPARATION) {
        // AgentSeparationSystem sits between KinematicMoveStopSystem and the animation systems (DemoScene.swift:66-71) and sees the
        // whole crowd: run the stages before it for everybody, then it, then the rest
        if (first != 0 || (size_t)count != w.bodies.size()) return SGE_ERR_INVALID;
        const uint32_t before = SGE_STAGE_INTENT | SGE_STAGE_GRAVITY | SGE_STAGE_MOVE | SGE_STAGE_AGENTS;
        sge_tick_desc a = dd, b = dd;
        a.stages = dd.stages & before;
        b.stages = dd.stages & ~(before | SGE_STAGE_SEPARATION);
        if (a.stages & ~(uint32_t)SGE_STAGE_AGENTS) {
            int rc = sgeo_tick_mt_stages(h, a, first, count, threads, useAgents ? &agents : nullptr, selfOffset);
            if (rc != SGE_OK) return rc;
        }
        agent_separation_fixed_update(w, w.separationIterations, w.separationMargin, w.separationHeightMargin);
        QueryStats s = take_thread_stats();
        h->stats.candidates += s.candidates; h->stats.sweeps += s.sweeps; h->stats.iterations += s.iterations; h->stats.queries += s.queries;
        if (b.stages) return sgeo_tick_mt_stages(h, b, first, count, threads, nullptr, 0);
        return SGE_OK;
    }
    return sgeo_tick_mt_stages(h, dd, first, count, threads, useAgents ? &agents : nullptr, selfOffset);
}

// the stages of `dd` over [first, first + count) on `threads` host threads
static int sgeo_tick_mt_stages(sgeo_world* h, const sge_tick_desc& dd, int first, int count, int threads,
                               const std::vector<AgentSweepState>* agentsPtr, int selfOffset) {
    if (threads <= 1) {
        run_range(h, dd, first, count, agentsPtr, selfOffset);
        QueryStats s = take_thread_stats();
        h->stats.candidates += s.candidates; h->stats.sweeps += s.sweeps; h->stats.iterations += s.iterations; h->stats.queries += s.queries;
        return SGE_OK;
    }
    std::mutex mu;
    std::vector<std::thread> pool;
    int chunk = (count + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        int f = first + t * chunk, c = std::min(chunk, first + count - f);
        if (c <= 0) break;
        pool.emplace_back([&, f, c]() {
            run_range(h, dd, f, c, agentsPtr, selfOffset);
            QueryStats s = take_thread_stats();
            std::lock_guard<std::mutex> lk(mu);
            h->stats.candidates += s.candidates; h->stats.sweeps += s.sweeps; h->stats.iterations += s.iterations; h->stats.queries += s.queries;
        });
    }
    for (auto& th : pool) th.join();
    return SGE_OK;
}

int sgeo_tick(sgeo_world* h, const sge_tick_desc* d) { return sgeo_tick_mt(h, d, 1); }

// collectAgentStates for this world's characters (after the caller ran intent+gravity)
int sgeo_agents_export(sgeo_world* h, sge_agent_state* out) {
    World& w = h->w;
    for (size_t e = 0; e < w.bodies.size(); ++e) {
        const sge_controller_params& P = w.params[e];
        bool solid = (P.agentFlags & SGE_AGENT_PRESENT) && (P.agentFlags & SGE_AGENT_SOLID);
        float radius = (P.agentFlags & SGE_AGENT_RADIUS_OVERRIDE) ? P.agentRadiusOverride : P.radius;
        for (int k = 0; k < 3; ++k) {
            out[e].position[k] = (float)w.bodies[e].position[k];
            out[e].velocity[k] = (float)w.bodies[e].linearVelocity[k];
        }
        out[e].radius = solid ? radius : -1.0f;
        out[e].halfHeight = P.halfHeight;
    }
    return SGE_OK;
}

int sgeo_agents_import(sgeo_world* h, const sge_agent_state* all, int32_t total, int32_t selfOffset) {
    h->w.importedAgents.assign(all, all + total);
    h->w.agentSelfOffset = selfOffset;
    h->w.agentsImported = total > 0;
    return SGE_OK;
}

int sgeo_separation_params(sgeo_world* h, int32_t iterations, float separationMargin, float heightMargin) {
    if (!h) return SGE_ERR_INVALID;
    h->w.separationIterations = iterations < 1 ? 1 : iterations; // max(1, iterations) :2146
    h->w.separationMargin = separationMargin;
    h->w.separationHeightMargin = heightMargin;
    return SGE_OK;
}

int sgeo_move_stats_read(sgeo_world* h, sge_move_stats* out, int reset) {
    out->queries = (uint64_t)h->stats.queries;
    out->candidates = (uint64_t)h->stats.candidates;
    out->sweepIterations = (uint64_t)h->stats.iterations;
    out->overflow = 0;
    out->traversalSteps = 0;
    out->sweepTrips = 0;
    out->prunedPairs = 0;
    if (reset) h->stats = QueryStats();
    return SGE_OK;
}

// ---- the step after skinning: index slice + instance matrices + brute-force closest hit (sge_oracle_blas.cpp) ----
int sgeo_blas_build(sgeo_world* h, const uint32_t* indices, int32_t index_count) {
    if (!h || !indices || index_count <= 0 || index_count % 3 != 0) return SGE_ERR_INVALID;
    if (h->w.mesh.vertexCount == 0) return SGE_ERR_STATE;
    for (int i = 0; i < index_count; ++i)
        if (indices[i] >= (uint32_t)h->w.mesh.vertexCount) return SGE_ERR_INVALID;
    h->w.blasIndices.assign(indices, indices + index_count);
    h->w.blasUVs.clear();
    return SGE_OK;
}

int sgeo_blas_set_uvs(sgeo_world* h, const float* uvs, int32_t vertex_count) {
    if (!h || !uvs) return SGE_ERR_INVALID;
    if (h->w.blasIndices.empty()) return SGE_ERR_STATE;
    if (vertex_count != h->w.mesh.vertexCount) return SGE_ERR_INVALID;
    h->w.blasUVs.assign(uvs, uvs + (size_t)vertex_count * 2);
    return SGE_OK;
}

int sgeo_blas_instances_upload(sgeo_world* h, int32_t first, int32_t count, const float* m) {
    if (!h || !m || first < 0 || count < 0 || (size_t)(first + count) > h->w.bodies.size()) return SGE_ERR_INVALID;
    const size_t N = h->w.bodies.size();
    if (h->w.blasInstances.size() != N * 16) {
        h->w.blasInstances.assign(N * 16, 0.0f);
        for (size_t i = 0; i < N; ++i) for (int k = 0; k < 4; ++k) h->w.blasInstances[i * 16 + k * 5] = 1.0f;
    }
    std::memcpy(&h->w.blasInstances[(size_t)first * 16], m, (size_t)count * 64);
    return SGE_OK;
}

int sgeo_blas_intersect_batch(sgeo_world* h, const sge_blas_ray* rays, int32_t count, sge_blas_hit* hits) {
    if (!h || count < 0 || (count > 0 && (!rays || !hits))) return SGE_ERR_INVALID;
    if (h->w.blasIndices.empty() || h->w.bodies.empty()) return SGE_ERR_STATE;
    if (h->w.outPositions.size() < h->w.bodies.size() * (size_t)h->w.mesh.vertexCount * 3) return SGE_ERR_STATE;
    for (int i = 0; i < count; ++i) blas_intersect(h->w, rays[i], hits[i]);
    return SGE_OK;
}

} // extern "C"
