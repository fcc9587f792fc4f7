// C ABI of libsge_amd.so (include/sge_amd.h): context, uploads, the batched
// fixed step and read-backs. Device memory is context-owned; everything is
// enqueued on one HIP stream in the reference's system order
// (Game/DemoScene.swift:57-75): intent -> gravity -> move -> locomotion -> action
// -> pose -> write-back, then the skinning dispatch of the render frame
// (Game/RayTracingScene.swift:28-43).
#include <cstring>
#include <map>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include "sge_internal.hpp"

namespace sge {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
int hip_fail(hipError_t e, const char* what) {
    g_error = std::string(what) + ": " + hipGetErrorString(e);
    return SGE_ERR_DEVICE;
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n) {
        if (n <= bytes && p) return SGE_OK;
        release();
        if (n == 0) return SGE_OK;
        SGE_HIP(hipMalloc(&p, n));
        bytes = n;
        return SGE_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Events {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double ms = 0;
    long long launches = 0;
    std::vector<float>* each = nullptr; // when set: every launch's duration as well (bounded by the reader, sge_debug_skin_launch_times)
};

} // namespace sge

using namespace sge;

struct sge_context {
    int device = 0;
    hipStream_t ownStream = nullptr, stream = nullptr;
    // LBS of step n overlaps move/CCD of step n+1: skinning runs on its own stream, ordered by two events
    hipStream_t skinStream = nullptr;
    hipEvent_t evPoseDone = nullptr, evSkinDone[2] = {nullptr, nullptr}, evMainMark = nullptr, evConsumed = nullptr;
    int lastSkin = 0;            // palette buffer (= event slot) of the newest overlapped skin launch
    // overlap mode, whole-crowd ticks with move + pose + skin: pose(n) on a stream of its own, beside move(n+1) — it reads the move
    // stage's results from a copy the move kernels write with their write-back (PoseInput, two buffers), so the step's dependent chain
    // on the main stream is the move stage alone
    hipStream_t poseStream = nullptr;
    hipEvent_t evMoveDone = nullptr, evPosePiped[2] = {nullptr, nullptr};
    bool posePending[2] = {false, false}, pipelinePose = true;
    int poseSlot = 0;            // PoseInput buffer (= event slot) of the newest pipelined pose launch
    // two palette buffers: with the overlap option pose(n+1) writes one while skin(n) still reads the other, so that only
    // skin(n+1) -> skin(n) and skin(n+1) -> pose(n+1) remain as dependencies (a single buffer chains pose(n+1) behind skin(n))
    int palRead = 0;             // buffer holding the latest palettes
    hipStream_t heavyStream = nullptr; // part 1 of the move stage for the step's heavy characters
    hipEvent_t evClassified = nullptr, evHeavyDone = nullptr, evTablesCopied = nullptr;
    int placementProbes = 8;   // candidate placements of the skinned output streams compared at allocation time
    float placementMs = 0; int placementTried = 0;
    int overlapSkinWorkgroups = 3; // LBS workgroups per CU while it shares the chip with the next step's collision kernels
    // overlap mode, LBS as resident workgroups with a ticket counter (quarters of a workgroup per CU): -1 = by crowd size
    // (kResidentSkinQuarters from kResidentSkinCharacters characters on, see there), 0 = never, q > 0 = always (SGE_SKIN_PERSISTENT)
    int residentSkinQuarters = -1;
    int residentSkinCharsPerUnit = kResidentSkinCharsPerUnit; // characters per work unit of the resident form (SGE_SKIN_CPW: 1, 2, 4, 8)
    int lastMoveCount = 1;       // characters of the move stage whose cost sum sits in hHeavyDemand[1]
    int lastSkinQuarters = 0, lastSkinCharsPerUnit = 1; // the form the newest skin stage took (sge_debug_skin_form)
    int overlapFusedWorkgroups = 0; // cap for the persistent workgroups of the fused LBS + refit kernel (0: as many as the LDS holds; measured 1 / 2 / 3: 1.53 / 1.47 / 1.44 ms per step)
    int heavyThreshold = 4000; // distance evaluations in a character's last step above which it takes the multi-wave kernel; < 0: off
    int heavyCap = 2048;       // most characters the multi-wave launch takes per step (its grid is sized by demand, see sge_tick)
    int* hHeavyDemand = nullptr; // pinned host word the move stage copies its demand count to
    // separation stage on crowds: {an agent was pushed further than a cell, a pass was redone serially} of the newest step whose copy
    // has landed (pinned); after a "pushed" step the candidates of the following steps come from 7 x 7 cells instead of 5 x 5
    int* hSepFlags = nullptr;
    int sepWideSteps = 0;
    // the heavy / order lists of the next move stage, built behind the last one (launch_move): valid for exactly this range / threshold
    bool listsValid = false; int listsFirst = 0, listsCount = 0, listsThreshold = 0, listsCap = 0;
    hipEvent_t evListsReady = nullptr;
    bool skinPending[2] = {false, false}, overlapSkin = false, customStream = false;
    int overlapRequested = 0;    // SGE_OPT_OVERLAP_SKIN as set: 0 off, 1 on the context's own stream only, 2 on a caller's stream as well
                                 // (overlapSkin = what holds for the stream in use; see applyOverlapOption)
    // options
    bool storePoseDebug = false, profile = false;
    bool waveProfOn = false; // SGE_WAVE_PROF=1: in-kernel cycle stamps of the move and pose kernels (diagnostics, tools/wave_prof.py)
    int skinLayout = SGE_LAYOUT_PACKED;
    // skeleton
    int boneCount = 0;
    DevSkeleton sk{};
    std::vector<float> hostSkeletonInvBind; // [B][16] skeleton.invBindModel
    std::vector<int32_t> hostParent;        // [B]
    std::vector<uint8_t> hostAnimated;      // [B] 1: some uploaded profile has an entry for the bone
    DevBuf dParent, dDepth, dLeanChain, dPath, dBindLocal, dInvBind, dRestT, dRawRestT, dPreRot, dSlotBone;
    // profiles
    DevProfiles prof{};
    DevBuf dCoeffs, dCoeffCount, dBonePresent;
    // mesh
    DevMesh mesh{};
    DevBuf dMeshPos, dMeshNrm, dMeshTan, dMeshIdx, dMeshWgt;
    // collision
    HostCollision hostCol, hostDyn; // StaticTriMesh.staticSet / dynamicSet (CollisionQuery.swift:710-711)
    DevCollision col{};
    DevBuf dWide, dTris, dMaterials, dBinNodes[2], dSlotOfRank, dPlatforms, dRayQueries, dRayOut, dCost, dHint, dHeavyFlags, dLists, dListCounts, dJobTable, dBlockJob;
    int platformCount = 0;
    // crowd
    DevCrowd crowd{};
    DevBuf dBodies, dParams, dCtrl, dIntents, dLoco, dActions, dPalettes[2], dPoseModel, dPoseLocal, dMoveScratch, dPoseIn[2];
    DevBuf dOutPos, dOutNrm, dOutTan;
    int outLayoutAllocated = -1;
    // agents
    DevAgents agents{};
    DevBuf dCellStart, dCellItems, dCellCursor, dAgentMinMax, dAgentGrid, dAgentsAll;
    // scratch for batched queries
    DevBuf dQueries, dCastOut, dOverlapOut, dCounts;
    // skinned-geometry acceleration structure (RTAccelerationBuilder.swift:75-145)
    std::vector<float> hostMeshPos; // source positions as uploaded: the topology is built from them
    HostBlas hostBlas;
    DevBlas blas{};
    // a tick with both the skin and the refit stage: 1 = one kernel unless the skin stage overlaps the next step (its persistent
    // workgroups hold 448 of every SIMD's 512 VGPRs, no collision kernel (128-176) fits beside them and the two sides run one
    // after the other: 1.49 against 1.46 ms per step; serial order 1.85 -> 1.61 ms), 2 = always, 0 = never
    int fuseBlas = 1;
    int blasBoundsChars = 0;
    DevBuf dBlasEntryLink, dBlasWideFirst, dBlasWideParent, dBlasWideLevel, dBlasSlotIdx, dBlasSlotTri,
           dBlasIndices, dBlasBounds, dBlasInstances, dBlasRays, dBlasHits, dBlasTileStart, dBlasRoundLen, dBlasRoundCluster, dBlasRoundIds, dBlasWorldBoxes, dBlasUVs, dBlasQueue, dSkinQueue, dBlasInstPoints, dBlasInstMinMax, dBlasInstGrid, dBlasInstOrder, dBlasGroupBoxes;
    bool blasHasUVs = false;
    // stats / profiling
    DevBuf dStats, dWaveProf, dOrderHist, dSepAgents, dSepCounts, dSepFlow;
    int separationIterations = 2; float separationMargin = 0.2f, separationHeightMargin = 0.1f; // AgentSeparationSystem.init :2146-2152
    Events evMove, evPose, evSkin, evAgents, evBlas;
    std::vector<float> skinLaunchMs; // per-launch durations of the skin stage since the last profile reset (SGE_OPT_PROFILE)
    // asynchronous World synchronisation (sge_state_*): device-side snapshots behind the kernels that wrote the arrays, moved to pinned
    // host memory on a stream of their own; pinned staging the other way
    hipStream_t copyStream = nullptr;
    struct PullSlot {
        int ticket = -1, first = 0, count = 0; uint32_t which = 0; bool inFlight = false;
        size_t off[5] = {0, 0, 0, 0, 0};
        DevBuf stage; void* host = nullptr; size_t hostBytes = 0;
        hipEvent_t evMain = nullptr, evPose = nullptr, evLanded = nullptr;
    } pull[2];
    int pullTickets = 0;
    struct PushSlot { void* host = nullptr; size_t hostBytes = 0; hipEvent_t evCopied = nullptr; bool inFlight = false; } push[2];
    int pushSlot = 0, pushFirst = 0, pushCount = 0; uint32_t pushWhich = 0; bool pushOpen = false;
    size_t pushOff[5] = {0, 0, 0, 0, 0};
};

namespace {

int upload(DevBuf& b, const void* src, size_t bytes, hipStream_t s) {
    int rc = b.alloc(bytes);
    if (rc != SGE_OK) return rc;
    if (bytes) SGE_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s));
    return SGE_OK;
}

constexpr size_t kMaxPendingEvents = 2048;
int drainEvents(Events& ev) {
    for (auto& pr : ev.pending) {
        SGE_HIP(hipEventSynchronize(pr.second));
        float ms = 0;
        SGE_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
        ev.ms += ms;
        ev.launches += 1;
        if (ev.each && ev.each->size() < 65536) ev.each->push_back(ms);
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    ev.pending.clear();
    return SGE_OK;
}

struct Bracket {
    sge_context* c; Events* ev; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    Bracket(sge_context* ctx, Events* e, hipStream_t stream = nullptr) : c(ctx), ev(e), s(stream ? stream : ctx->stream) {
        if (c->profile) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, s); }
    }
    ~Bracket() {
        if (!c->profile) return;
        (void)hipEventRecord(b, s);
        ev->pending.emplace_back(a, b);
        // a host that never calls sge_profile_read must not accumulate events without bound: fold the finished ones in
        if (ev->pending.size() >= kMaxPendingEvents) (void)drainEvents(*ev);
    }
};

// Everything enqueued so far on either stream has completed.
int syncAll(sge_context* c) {
    SGE_HIP(hipStreamSynchronize(c->stream));
    if (c->skinStream) SGE_HIP(hipStreamSynchronize(c->skinStream));
    if (c->heavyStream) SGE_HIP(hipStreamSynchronize(c->heavyStream));
    if (c->poseStream) SGE_HIP(hipStreamSynchronize(c->poseStream));
    if (c->copyStream) SGE_HIP(hipStreamSynchronize(c->copyStream));
    c->skinPending[0] = c->skinPending[1] = false;
    c->posePending[0] = c->posePending[1] = false;
    return SGE_OK;
}
// The main stream must not touch what the animation stages own (locomotion / action states, transformRotation, palettes) while a
// pose launch on the pose stream is in flight.
int joinPose(sge_context* c) {
    for (int f = 0; f < 2; ++f)
        if (c->posePending[f]) { SGE_HIP(hipStreamWaitEvent(c->stream, c->evPosePiped[f], 0)); c->posePending[f] = false; }
    return SGE_OK;
}
// The main stream must not touch palettes / skinned outputs while a skin launch is in flight.
int joinSkin(sge_context* c) {
    { int rcp = joinPose(c); if (rcp != SGE_OK) return rcp; }
    for (int f = 0; f < 2; ++f)
        if (c->skinPending[f]) { SGE_HIP(hipStreamWaitEvent(c->stream, c->evSkinDone[f], 0)); c->skinPending[f] = false; }
    return SGE_OK;
}

// SGE_OPT_OVERLAP_SKIN = 1 keeps the contract of ABI version 1 on a caller-provided stream: there the option is ignored, what the
// caller enqueues behind sge_tick is ordered behind the skin launch, and the palette pointer does not move. Value 2 is the explicit
// opt-in for a caller that orders its consumers with sge_skin_wait / sge_skin_consumed. Callers have synchronised (syncAll) before.
void applyOverlapOption(sge_context* c) { c->overlapSkin = c->overlapRequested >= 2 || (c->overlapRequested == 1 && !c->customStream); }

int refreshInvBind(sge_context* c, const float* meshInvBind, int meshInvBindCount) {
    // Systems.swift:2523: re-bind only when the mesh carries invBindModel of matching count
    const float* src = (meshInvBind && meshInvBindCount == c->boneCount) ? meshInvBind : c->hostSkeletonInvBind.data();
    if (c->boneCount == 0) return SGE_OK;
    int rc = upload(c->dInvBind, src, (size_t)c->boneCount * 64, c->stream);
    c->sk.invBind = c->dInvBind.as<float>();
    return rc;
}

// The order in which pose_kernel's lanes take the bones (DevSkeleton::slotBone). A skeleton whose bone count is not a multiple of
// 64 leaves its last pass partly filled (the Y-Bot: 65 bones, one lane busy in pass two), and that pass costs the wavefront as much
// as a full one. The bones put there are therefore the cheapest the rig has: bones no uploaded profile has an entry for (their
// local matrix is pre-rotation + rest translation, nothing to evaluate), leaves first, and never a bone together with its parent
// (so that their model matrix is one product with the parent's, finished a pass earlier). Rebuilt after either upload.
int rebuildPoseSlots(sge_context* c) {
    const int B = c->boneCount;
    if (B == 0) return SGE_OK;
    const std::vector<int32_t>& parent = c->hostParent;
    std::vector<uint8_t> animated = c->hostAnimated;
    animated.resize(B, 0);
    const int passes = (B + 63) / 64, extra = B - 64 * (passes - 1);
    std::vector<int32_t> slots;
    std::vector<uint8_t> isExtra(B, 0);
    if (passes > 1 && extra < 64) {
        std::vector<uint8_t> hasChild(B, 0);
        for (int i = 0; i < B; ++i) if (parent[i] >= 0) hasChild[parent[i]] = 1;
        // preference: static leaves, static inner bones, animated leaves, the rest; inside a class the highest index first
        int picked = 0;
        for (int cls = 0; cls < 4 && picked < extra; ++cls)
            for (int i = B - 1; i >= 1 && picked < extra; --i) { // (bone 0 carries the root special cases: it stays in pass one)
                if ((animated[i] ? 2 : 0) + (hasChild[i] ? 1 : 0) != cls || isExtra[i]) continue;
                if (parent[i] >= 0 && isExtra[parent[i]]) continue;      // not together with its parent ...
                bool childPicked = false;
                for (int j = i + 1; j < B && !childPicked; ++j) childPicked = isExtra[j] && parent[j] == i; // ... or one of its children
                if (childPicked) continue;
                isExtra[i] = 1;
                picked += 1;
            }
        if (picked != extra) std::fill(isExtra.begin(), isExtra.end(), 0); // no such choice on this rig: index order
    }
    for (int i = 0; i < B; ++i) if (!isExtra[i]) slots.push_back(i);
    for (int i = 0; i < B; ++i) if (isExtra[i]) slots.push_back(i);
    std::vector<int> slotOf(B);
    for (int s = 0; s < B; ++s) slotOf[slots[s]] = s;
    int lastPassStatic = passes > 1 ? 1 : 0, parentReady = 1;
    for (int s = 64; s < B; ++s) {
        const int p = parent[slots[s]];
        if (p >= 0 && slotOf[p] / 64 >= s / 64) parentReady = 0; // a bone of a later pass needs its parent's model from an earlier one
    }
    for (int s = 64 * (passes - 1); s < B && passes > 1; ++s) if (animated[slots[s]]) lastPassStatic = 0;
    int rc = upload(c->dSlotBone, slots.data(), (size_t)B * 4, c->stream);
    if (rc != SGE_OK) return rc;
    SGE_HIP(hipStreamSynchronize(c->stream));
    c->sk.slotBone = c->dSlotBone.as<int32_t>();
    c->sk.lastPassStatic = lastPassStatic;
    c->sk.extraParentReady = parentReady;
    return SGE_OK;
}

// The three skinned output streams are the path's HBM traffic (40 B per vertex per step). On MI355X the rate of the three-stream
// store pattern depends on where the driver placed the buffers: 0.81 or 1.05-1.10 ms for 5.63 GB, bimodal and reproducible per
// allocation. What decides is the placement of ONE stream — in every exchange of single buffers between a fast and a slow set the
// 16-byte tangent stream carried the property with it (tools/alloc_probe.hip), and inside one 96 GiB arena the tangent stream is fast
// anywhere in the first 64 GiB and slow in the rest, with positions and normals fixed (tools/tan_scan.hip); each buffer by itself
// streams at the same rate wherever it lies. So the streams are probed one at a time, tangents first: candidates for one stream are
// allocated (and held, so that they land in different places) and timed in the real three-stream pattern with the other two fixed.
// A candidate costs 16 (or 12) bytes per vertex instead of 40: a crowd that fills half the memory can still be probed.
int allocCrowdOutputs(sge_context* c) {
    const size_t verts = (size_t)c->crowd.count * (size_t)c->mesh.vertexCount;
    const size_t stride = c->skinLayout == SGE_LAYOUT_PADDED16 ? 16 : 12;
    const size_t need[3] = {verts * stride, verts * stride, verts * 16};
    DevBuf* bufs[3] = {&c->dOutPos, &c->dOutNrm, &c->dOutTan};
    c->outLayoutAllocated = c->skinLayout;
    if (need[0] <= bufs[0]->bytes && need[1] <= bufs[1]->bytes && need[2] <= bufs[2]->bytes && bufs[0]->p) return SGE_OK;
    const size_t total = need[0] + need[1] + need[2];
    for (DevBuf* b : bufs) b->release();
    if (total == 0) return SGE_OK;
    int rc;
    for (int k = 0; k < 3; ++k) if ((rc = bufs[k]->alloc(need[k])) != SGE_OK) return rc;
    c->placementMs = 0;
    c->placementTried = 1;
    if (c->placementProbes <= 1 || total < ((size_t)256 << 20)) return SGE_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    SGE_HIP(hipEventCreate(&e0));
    SGE_HIP(hipEventCreate(&e1));
    void* cur[3] = {bufs[0]->p, bufs[1]->p, bufs[2]->p};
    auto probe = [&](void* const p[3], float& ms) -> int {
        launch_store_probe(p[0], p[1], p[2], c->crowd.count, c->mesh.vertexCount, c->skinLayout, c->stream); // warm: first touch
        SGE_HIP(hipEventRecord(e0, c->stream));
        for (int r = 0; r < 2; ++r) launch_store_probe(p[0], p[1], p[2], c->crowd.count, c->mesh.vertexCount, c->skinLayout, c->stream);
        SGE_HIP(hipEventRecord(e1, c->stream));
        SGE_HIP(hipEventSynchronize(e1));
        SGE_HIP(hipEventElapsedTime(&ms, e0, e1));
        ms *= 0.5f;
        return SGE_OK;
    };
    const float goodMs = (float)((double)verts * (stride == 16 ? 48.0 : 40.0) / 6.5e12 * 1e3); // 6.5 TB/s: a good placement
    const bool debug = getenv("SGE_DEBUG_PLACEMENT") != nullptr;
    float bestMs = 0;
    if ((rc = probe(cur, bestMs)) != SGE_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
    if (debug) fprintf(stderr, "[sge] output placement: first allocation %.3f ms (good <= %.3f)\n", bestMs, goodMs);
    const int order[3] = {2, 0, 1}; // tangents, positions, normals
    for (int oi = 0; oi < 3 && bestMs > goodMs; ++oi) {
        const int s = order[oi];
        size_t freeB = 0, totalB = 0;
        int attempts = c->placementProbes - 1;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) attempts = (int)std::min<size_t>((size_t)attempts, freeB / 2 / need[s]);
        std::vector<void*> held; // candidates of stream s, all alive until the stream is decided
        // fast and slow regions are tens of GiB wide (the candidate sets of 5.6 GB each that this loop used to compare met a fast
        // one every 6-8 sets): untouched spacers between the candidates make the search stride ~12 GiB whatever the stream's size
        std::vector<void*> spacers;
        const size_t stride12 = (size_t)12 << 30;
        size_t spacer = need[s] < stride12 ? stride12 - need[s] : 0;
        if (attempts > 0) spacer = std::min(spacer, freeB / 4 / (size_t)attempts);
        void* best = cur[s];
        for (int a = 0; a < attempts && bestMs > goodMs; ++a) {
            if (spacer >= ((size_t)64 << 20)) {
                void* sp = nullptr;
                if (hipMalloc(&sp, spacer) == hipSuccess) spacers.push_back(sp); else (void)hipGetLastError();
            }
            void* cand = nullptr;
            if (hipMalloc(&cand, need[s]) != hipSuccess) { (void)hipGetLastError(); break; }
            held.push_back(cand);
            void* trial[3] = {cur[0], cur[1], cur[2]};
            trial[s] = cand;
            float ms = 0;
            if ((rc = probe(trial, ms)) != SGE_OK) { // every candidate, the spacers and the events go back; the streams stay as they are
                for (void* q : held) (void)hipFree(q);
                for (void* q : spacers) (void)hipFree(q);
                (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
                return rc;
            }
            c->placementTried += 1;
            if (debug) fprintf(stderr, "[sge]   stream %d candidate %p: %.3f ms\n", s, cand, ms);
            if (ms < bestMs) { bestMs = ms; best = cand; }
        }
        if (best != cur[s]) { (void)hipFree(cur[s]); cur[s] = best; bufs[s]->p = best; } // (the DevBuf never holds a freed pointer)
        for (void* q : held) if (q != best) (void)hipFree(q);
        for (void* q : spacers) (void)hipFree(q);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    for (int k = 0; k < 3; ++k) { bufs[k]->p = cur[k]; bufs[k]->bytes = need[k]; }
    c->placementMs = bestMs;
    if (debug) fprintf(stderr, "[sge] output placement: %d timed, kept %.3f ms\n", c->placementTried, bestMs);
    return SGE_OK;
}

// Per-character buffers of the acceleration structure: boxes (zeroed until the first refit) and instance matrices (identity).
int ensureBlasBuffers(sge_context* c) {
    if (c->blas.entryCount == 0 || c->crowd.count == 0) return SGE_OK;
    const size_t N = (size_t)c->crowd.count, rowBytes = (size_t)(c->blas.entryCount + 1) * 24;
    int rc;
    if ((rc = c->dBlasBounds.alloc(N * rowBytes)) != SGE_OK) return rc;
    SGE_HIP(hipMemsetAsync(c->dBlasBounds.p, 0, N * rowBytes, c->stream));
    if (c->blasBoundsChars != c->crowd.count) {
        std::vector<float> ident(N * 16, 0.0f);
        for (size_t i = 0; i < N; ++i) ident[i * 16] = ident[i * 16 + 5] = ident[i * 16 + 10] = ident[i * 16 + 15] = 1.0f;
        if ((rc = upload(c->dBlasInstances, ident.data(), N * 64, c->stream)) != SGE_OK) return rc;
        SGE_HIP(hipStreamSynchronize(c->stream));
        c->blasBoundsChars = c->crowd.count;
    }
    return SGE_OK;
}

// ---- agent grid (uniform XZ cells over the gathered snapshot) ---------------
__global__ void agentBoundsKernel(const sge_agent_state* a, int n, float* mm /*minx,minz,maxx,maxz,maxr,maxv*/) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float minx = 3e38f, minz = 3e38f, maxx = -3e38f, maxz = -3e38f, maxr = 0.f, maxv = 0.f;
    if (i < n && a[i].radius >= 0) {
        minx = maxx = a[i].position[0];
        minz = maxz = a[i].position[2];
        maxr = a[i].radius;
        // per-axis bound is enough: the sweep reach uses |v| <= sqrt(3)*max|v_k|; keep the true length
        float vx = a[i].velocity[0], vy = a[i].velocity[1], vz = a[i].velocity[2];
        maxv = sqrtf(vx * vx + vy * vy + vz * vz);
    }
    for (int o = 32; o > 0; o >>= 1) {
        minx = fminf(minx, __shfl_xor(minx, o, 64)); minz = fminf(minz, __shfl_xor(minz, o, 64));
        maxx = fmaxf(maxx, __shfl_xor(maxx, o, 64)); maxz = fmaxf(maxz, __shfl_xor(maxz, o, 64));
        maxr = fmaxf(maxr, __shfl_xor(maxr, o, 64)); maxv = fmaxf(maxv, __shfl_xor(maxv, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        // positive floats order as unsigned ints; bias signed values through an order-preserving map
        auto enc = [](float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
        atomicMin(reinterpret_cast<unsigned*>(mm) + 0, enc(minx));
        atomicMin(reinterpret_cast<unsigned*>(mm) + 1, enc(minz));
        atomicMax(reinterpret_cast<unsigned*>(mm) + 2, enc(maxx));
        atomicMax(reinterpret_cast<unsigned*>(mm) + 3, enc(maxz));
        atomicMax(reinterpret_cast<unsigned*>(mm) + 4, enc(maxr));
        atomicMax(reinterpret_cast<unsigned*>(mm) + 5, enc(maxv));
    }
}

// grid parameters from the encoded bounds: one thread
__global__ void agentGridKernel(const unsigned* enc, AgentGrid* grid) {
    auto dec = [](unsigned u) { unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; return __uint_as_float(v); };
    AgentGrid G{};
    if (enc[0] != 0xffffffffu) { // at least one solid agent
        const float minx = dec(enc[0]), minz = dec(enc[1]), maxx = dec(enc[2]), maxz = dec(enc[3]);
        G.maxRadius = dec(enc[4]);
        G.maxSpeed = dec(enc[5]);
        float cell = 4.0f * (G.maxRadius > 0.25f ? G.maxRadius : 0.25f);
        int nx = (int)((maxx - minx) / cell) + 1, nz = (int)((maxz - minz) / cell) + 1;
        while ((long long)nx * nz > kAgentMaxCells) { cell *= 2; nx = (int)((maxx - minx) / cell) + 1; nz = (int)((maxz - minz) / cell) + 1; }
        G.originX = minx; G.originZ = minz; G.invCell = 1.0f / cell; G.nx = nx; G.nz = nz; G.cells = nx * nz;
    }
    *grid = G;
}
__device__ __forceinline__ int agentCell(const sge_agent_state& a, const AgentGrid& G) {
    int cx = (int)floorf((a.position[0] - G.originX) * G.invCell), cz = (int)floorf((a.position[2] - G.originZ) * G.invCell);
    cx = cx < 0 ? 0 : (cx >= G.nx ? G.nx - 1 : cx);
    cz = cz < 0 ? 0 : (cz >= G.nz ? G.nz - 1 : cz);
    return cz * G.nx + cx;
}
__global__ void agentClearKernel(const AgentGrid* grid, int* counts) {
    const int cells = grid->cells;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += gridDim.x * blockDim.x) counts[i] = 0;
}
__global__ void agentCountKernel(const sge_agent_state* a, int n, const AgentGrid* grid, int* counts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const AgentGrid G = *grid;
    if (G.nx == 0) return;
    if (i < n && a[i].radius >= 0) atomicAdd(&counts[agentCell(a[i], G)], 1);
}
__global__ void agentScanKernel(const int* counts, const AgentGrid* grid, int* start, int* cursor) {
    // single 1024-thread block exclusive scan (cells <= 1<<20)
    __shared__ int part[1024];
    const int cells = grid->cells;
    int tid = threadIdx.x;
    int per = (cells + 1023) / 1024;
    int b = tid * per, e = b + per < cells ? b + per : cells;
    if (b > cells) b = cells;
    int s = 0;
    for (int i = b; i < e; ++i) s += counts[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid == 0 ? 0 : part[tid - 1];
    for (int i = b; i < e; ++i) { start[i] = run; cursor[i] = run; run += counts[i]; }
    if (tid == 1023) start[cells] = part[1023];
}
__global__ void agentScatterKernel(const sge_agent_state* a, int n, const AgentGrid* grid, int* cursor, int* items) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const AgentGrid G = *grid;
    if (G.nx == 0) return;
    if (i < n && a[i].radius >= 0) items[atomicAdd(&cursor[agentCell(a[i], G)], 1)] = i;
}

// Bins `total` capsule records into a uniform XZ grid, entirely on the device and on stream `s` (sge_tick stays asynchronous):
// bounds -> grid parameters -> count / scan / scatter. The cell arrays (shared scratch of the context) are sized once for the
// largest grid; `items` receives the record indices sorted by cell, `grid` the parameters, `minmax` is 32 bytes of scratch.
int buildXZGrid(sge_context* c, hipStream_t s, const sge_agent_state* all, int total, DevBuf& minmax, DevBuf& gridBuf, DevBuf& items) {
    int rc;
    if ((rc = minmax.alloc(32)) != SGE_OK) return rc;
    if ((rc = gridBuf.alloc(sizeof(AgentGrid))) != SGE_OK) return rc;
    // counts live in dCellCursor's tail: [cursor cells][counts cells]
    if ((rc = c->dCellCursor.alloc((size_t)kAgentMaxCells * 8)) != SGE_OK) return rc;
    if ((rc = c->dCellStart.alloc((size_t)(kAgentMaxCells + 1) * 4)) != SGE_OK) return rc;
    if ((rc = items.alloc((size_t)total * 4)) != SGE_OK) return rc;
    static const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u};
    SGE_HIP(hipMemcpyAsync(minmax.p, init, sizeof(init), hipMemcpyHostToDevice, s));
    const int blocks = (total + 255) / 256;
    AgentGrid* grid = gridBuf.as<AgentGrid>();
    int* cursor = c->dCellCursor.as<int>();
    int* cnt = cursor + kAgentMaxCells;
    hipLaunchKernelGGL(agentBoundsKernel, dim3(blocks), dim3(256), 0, s, all, total, minmax.as<float>());
    hipLaunchKernelGGL(agentGridKernel, dim3(1), dim3(1), 0, s, minmax.as<unsigned>(), grid);
    hipLaunchKernelGGL(agentClearKernel, dim3(256), dim3(256), 0, s, grid, cnt);
    hipLaunchKernelGGL(agentCountKernel, dim3(blocks), dim3(256), 0, s, all, total, grid, cnt);
    hipLaunchKernelGGL(agentScanKernel, dim3(1), dim3(1024), 0, s, cnt, grid, c->dCellStart.as<int>(), cursor);
    hipLaunchKernelGGL(agentScatterKernel, dim3(blocks), dim3(256), 0, s, all, total, grid, cursor, items.as<int>());
    return SGE_OK;
}

// the gathered agents of the character-vs-character sweeps (SGE_STAGE_AGENTS)
int buildAgentGrid(sge_context* c) {
    DevAgents& ag = c->agents;
    if (!ag.all || ag.total <= 0) return SGE_OK;
    int rc = buildXZGrid(c, c->stream, ag.all, ag.total, c->dAgentMinMax, c->dAgentGrid, c->dCellItems);
    if (rc != SGE_OK) return rc;
    ag.grid = c->dAgentGrid.as<AgentGrid>();
    ag.cellStart = c->dCellStart.as<int>();
    ag.cellItems = c->dCellItems.as<int>();
    return SGE_OK;
}

// The instance level of `isect.intersect(ray, accel)` for rays that name no character (RTAccelerationBuilder.swift:168-185 rebuilds
// the TLAS over all items every frame): world boxes of every character, the characters sorted by the cell of the same XZ grid the
// agent sweeps use (over the boxes' centres), one box per 64 consecutive characters of THAT order, one per 64 such groups. A ray
// then tests 64 super-groups, 64 groups, 64 instances per step where the flat form scanned ceil(N / 64) index groups: 250,000
// characters are 62 super-groups. Fills the trace's pointers; everything on the context's stream.
int buildInstanceLevel(sge_context* c, BlasTrace& T) {
    const int N = c->crowd.count, groups = (N + 63) / 64, supers = (groups + 63) / 64;
    int rc;
    if ((rc = c->dBlasWorldBoxes.alloc(((size_t)N + groups) * 24)) != SGE_OK) return rc;
    if ((rc = c->dBlasInstPoints.alloc((size_t)N * sizeof(sge_agent_state))) != SGE_OK) return rc;
    if ((rc = c->dBlasGroupBoxes.alloc(((size_t)groups + supers) * 24)) != SGE_OK) return rc;
    T.worldBoxes = c->dBlasWorldBoxes.as<float>();
    launch_blas_world_boxes(T, c->dBlasWorldBoxes.as<float>(), c->stream);
    const bool flat = getenv("SGE_BLAS_FLAT_INSTANCES") != nullptr; // experiments / tests: the flat scan of 64 consecutive indices (read per call)
    if (flat) { T.instOrder = nullptr; return SGE_OK; }
    launch_blas_instance_points(c->dBlasWorldBoxes.as<float>(), N, c->dBlasInstPoints.as<sge_agent_state>(), c->stream);
    if ((rc = buildXZGrid(c, c->stream, c->dBlasInstPoints.as<sge_agent_state>(), N, c->dBlasInstMinMax, c->dBlasInstGrid, c->dBlasInstOrder)) != SGE_OK) return rc;
    float* gb = c->dBlasGroupBoxes.as<float>();
    launch_blas_group_boxes(c->dBlasWorldBoxes.as<float>(), c->dBlasInstOrder.as<int>(), N, gb, gb + (size_t)groups * 6, c->stream);
    T.instOrder = c->dBlasInstOrder.as<int>();
    T.groupBoxes = gb;
    T.superBoxes = gb + (size_t)groups * 6;
    return SGE_OK;
}

} // namespace

extern "C" {

int sge_abi_version(void) { return SGE_ABI_VERSION; }
const char* sge_last_error(void) { return g_error.c_str(); }

sge_context* sge_context_create(int device_index) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device visible (there is no CPU fallback)"); return nullptr; }
    if (device_index < 0 || device_index >= n) { set_error("device index out of range"); return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_index) != hipSuccess) { set_error("hipGetDeviceProperties failed"); return nullptr; }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        set_error(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
        return nullptr;
    }
    if (hipSetDevice(device_index) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    sge_context* c = new sge_context();
    c->device = device_index;
    c->evSkin.each = &c->skinLaunchMs;
    c->waveProfOn = getenv("SGE_WAVE_PROF") != nullptr;
    if (getenv("SGE_HEAVY_THRESHOLD")) c->heavyThreshold = atoi(getenv("SGE_HEAVY_THRESHOLD")); // experiments
    if (getenv("SGE_HEAVY_CAP")) c->heavyCap = std::max(1, atoi(getenv("SGE_HEAVY_CAP")));
    if (getenv("SGE_PLACEMENT_PROBES")) c->placementProbes = std::max(1, atoi(getenv("SGE_PLACEMENT_PROBES")));
    if (getenv("SGE_OVERLAP_SKIN_WORKGROUPS")) c->overlapSkinWorkgroups = atoi(getenv("SGE_OVERLAP_SKIN_WORKGROUPS"));
    if (getenv("SGE_SKIN_PERSISTENT")) c->residentSkinQuarters = atoi(getenv("SGE_SKIN_PERSISTENT"));
    if (getenv("SGE_SKIN_CPW")) c->residentSkinCharsPerUnit = atoi(getenv("SGE_SKIN_CPW"));
    if (getenv("SGE_OVERLAP_FUSED_WORKGROUPS")) c->overlapFusedWorkgroups = atoi(getenv("SGE_OVERLAP_FUSED_WORKGROUPS"));
    if (getenv("SGE_POSE_PIPELINE")) c->pipelinePose = atoi(getenv("SGE_POSE_PIPELINE")) != 0; // experiments: 0 = pose stays on the main stream
    // the latency-bound collision / pose launches go first when they compete with a streaming skin launch (overlap option)
    int prLeast = 0, prGreatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest);
    // experiments (SGE_STREAM_PRIO=main:heavy:pose:skin; 0 = greatest, 1 = one below it, ...)
    int prMain = prGreatest, prHeavy = prGreatest, prPose = prGreatest, prSkin = prLeast;
    if (getenv("SGE_STREAM_PRIO")) {
        int a = 0, b = 0, p = 0, k = 99;
        sscanf(getenv("SGE_STREAM_PRIO"), "%d:%d:%d:%d", &a, &b, &p, &k);
        auto lvl = [&](int v) { return std::min(prLeast, prGreatest + v); };
        prMain = lvl(a); prHeavy = lvl(b); prPose = lvl(p); prSkin = lvl(k);
    }
    if (hipStreamCreateWithPriority(&c->ownStream, hipStreamNonBlocking, prMain) != hipSuccess) { set_error("hipStreamCreate failed"); delete c; return nullptr; }
    c->stream = c->ownStream;
    // experiment (SGE_SKIN_CUS=n[:pattern]): the skin stream on a subset of the CUs, the rest left to the collision side alone
    hipError_t skinRc;
    if (getenv("SGE_SKIN_CUS")) {
        int n = 256, pattern = 0;
        sscanf(getenv("SGE_SKIN_CUS"), "%d:%d", &n, &pattern);
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // pattern 0: the first n bits; pattern 1: the first n / 8 bits of every 32-bit word
        for (int i = 0; i < 256; ++i) {
            const bool on = pattern == 0 ? i < n : (i % 32) < n / 8;
            if (on) mask[i / 32] |= 1u << (i % 32);
        }
        skinRc = hipExtStreamCreateWithCUMask(&c->skinStream, 8, mask);
    } else {
        skinRc = hipStreamCreateWithPriority(&c->skinStream, hipStreamNonBlocking, prSkin);
    }
    if (skinRc != hipSuccess ||
        hipEventCreateWithFlags(&c->evPoseDone, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithPriority(&c->poseStream, hipStreamNonBlocking, prPose) != hipSuccess ||
        hipEventCreateWithFlags(&c->evMoveDone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evPosePiped[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evPosePiped[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evSkinDone[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evSkinDone[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evMainMark, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evConsumed, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evListsReady, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithPriority(&c->heavyStream, hipStreamNonBlocking, prHeavy) != hipSuccess ||
        hipEventCreateWithFlags(&c->evClassified, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evHeavyDone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evTablesCopied, hipEventDisableTiming) != hipSuccess) { set_error("stream/event creation failed"); delete c; return nullptr; }
    for (int k = 0; k < 2; ++k)
        if (hipEventCreateWithFlags(&c->pull[k].evMain, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->pull[k].evPose, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->pull[k].evLanded, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->push[k].evCopied, hipEventDisableTiming) != hipSuccess) { set_error("event creation failed"); delete c; return nullptr; }
    if (hipStreamCreateWithPriority(&c->copyStream, hipStreamNonBlocking, prGreatest) != hipSuccess) { set_error("hipStreamCreate failed"); delete c; return nullptr; }
    if (c->dSkinQueue.alloc(256) != SGE_OK || hipMemsetAsync(c->dSkinQueue.p, 0, 256, c->stream) != hipSuccess) { delete c; return nullptr; } // the resident LBS forms leave their ticket words at zero
    if (hipHostMalloc(reinterpret_cast<void**>(&c->hHeavyDemand), 2 * sizeof(int), hipHostMallocDefault) == hipSuccess) { c->hHeavyDemand[0] = -1; c->hHeavyDemand[1] = -1; }
    else { (void)hipGetLastError(); c->hHeavyDemand = nullptr; }
    if (hipHostMalloc(reinterpret_cast<void**>(&c->hSepFlags), 2 * sizeof(int), hipHostMallocDefault) == hipSuccess) { c->hSepFlags[0] = 0; c->hSepFlags[1] = 0; }
    else { (void)hipGetLastError(); c->hSepFlags = nullptr; }
    if (c->dStats.alloc((size_t)kStatShards * 64) != SGE_OK || hipMemsetAsync(c->dStats.p, 0, (size_t)kStatShards * 64, c->stream) != hipSuccess) { delete c; return nullptr; }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { set_error("context creation: device synchronisation failed"); delete c; return nullptr; } // the zeroed words are read from other streams
    return c;
}

void sge_context_destroy(sge_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)syncAll(c);
    drainEvents(c->evMove); drainEvents(c->evPose); drainEvents(c->evSkin); drainEvents(c->evAgents); drainEvents(c->evBlas);
    DevBuf* bufs[] = {&c->dSlotBone, &c->dParent, &c->dDepth, &c->dLeanChain, &c->dPath, &c->dBindLocal, &c->dInvBind, &c->dRestT, &c->dRawRestT, &c->dPreRot,
                      &c->dCoeffs, &c->dCoeffCount, &c->dBonePresent, &c->dMeshPos, &c->dMeshNrm, &c->dMeshTan, &c->dMeshIdx, &c->dMeshWgt,
                      &c->dWide, &c->dTris, &c->dMaterials, &c->dBinNodes[0], &c->dBinNodes[1], &c->dSlotOfRank, &c->dCost, &c->dHint, &c->dHeavyFlags, &c->dJobTable, &c->dBlockJob, &c->dLists, &c->dListCounts, &c->dOrderHist, &c->dWaveProf, &c->dSepAgents, &c->dSepCounts, &c->dSepFlow, &c->dPlatforms, &c->dRayQueries, &c->dRayOut, &c->dBodies, &c->dParams, &c->dCtrl, &c->dIntents, &c->dLoco, &c->dActions,
                      &c->dPalettes[0], &c->dPalettes[1], &c->dPoseIn[0], &c->dPoseIn[1], &c->dPoseModel, &c->dPoseLocal, &c->dMoveScratch, &c->dOutPos, &c->dOutNrm, &c->dOutTan, &c->dCellStart, &c->dCellItems,
                      &c->dCellCursor, &c->dAgentMinMax, &c->dAgentGrid, &c->dAgentsAll, &c->dQueries, &c->dCastOut, &c->dOverlapOut, &c->dCounts, &c->dStats,
                      &c->dBlasEntryLink, &c->dBlasWideFirst, &c->dBlasWideParent, &c->dBlasWideLevel, &c->dBlasSlotIdx, &c->dBlasSlotTri,
                      &c->dBlasIndices, &c->dBlasBounds, &c->dBlasInstances, &c->dBlasRays, &c->dBlasHits, &c->dBlasTileStart,
                      &c->dBlasRoundLen, &c->dBlasRoundCluster, &c->dBlasRoundIds, &c->dBlasWorldBoxes, &c->dBlasUVs, &c->dBlasQueue, &c->dSkinQueue,
                      &c->dBlasInstPoints, &c->dBlasInstMinMax, &c->dBlasInstGrid, &c->dBlasInstOrder, &c->dBlasGroupBoxes};
    for (DevBuf* b : bufs) b->release();
    for (int k = 0; k < 2; ++k) {
        c->pull[k].stage.release();
        if (c->pull[k].host) (void)hipHostFree(c->pull[k].host);
        if (c->push[k].host) (void)hipHostFree(c->push[k].host);
        for (hipEvent_t e : {c->pull[k].evMain, c->pull[k].evPose, c->pull[k].evLanded, c->push[k].evCopied}) if (e) (void)hipEventDestroy(e);
    }
    if (c->copyStream) (void)hipStreamDestroy(c->copyStream);
    if (c->evPoseDone) (void)hipEventDestroy(c->evPoseDone);
    if (c->evMoveDone) (void)hipEventDestroy(c->evMoveDone);
    for (hipEvent_t e : c->evPosePiped) if (e) (void)hipEventDestroy(e);
    if (c->poseStream) (void)hipStreamDestroy(c->poseStream);
    if (c->hHeavyDemand) (void)hipHostFree(c->hHeavyDemand);
    if (c->hSepFlags) (void)hipHostFree(c->hSepFlags);
    if (c->evMainMark) (void)hipEventDestroy(c->evMainMark);
    if (c->evConsumed) (void)hipEventDestroy(c->evConsumed);
    if (c->evListsReady) (void)hipEventDestroy(c->evListsReady);
    for (hipEvent_t e : c->evSkinDone) if (e) (void)hipEventDestroy(e);
    if (c->skinStream) (void)hipStreamDestroy(c->skinStream);
    if (c->evClassified) (void)hipEventDestroy(c->evClassified);
    if (c->evHeavyDone) (void)hipEventDestroy(c->evHeavyDone);
    if (c->evTablesCopied) (void)hipEventDestroy(c->evTablesCopied);
    if (c->heavyStream) (void)hipStreamDestroy(c->heavyStream);
    if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    delete c;
}

int sge_context_set_stream(sge_context* c, void* hip_stream) {
    if (!c) return SGE_ERR_INVALID;
    int rc = syncAll(c);
    if (rc != SGE_OK) return rc;
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->ownStream;
    // (SGE_OPT_OVERLAP_SKIN keeps its meaning on a caller's stream: the skin launch then runs on the context's second stream, ordered
    // behind the pose stage on the caller's stream by events, and a consumer orders itself behind it with sge_skin_wait)
    c->customStream = hip_stream != nullptr;
    applyOverlapOption(c);
    return SGE_OK;
}

int sge_context_get_stream(sge_context* c, void** hip_stream) {
    if (!c || !hip_stream) return SGE_ERR_INVALID;
    *hip_stream = reinterpret_cast<void*>(c->stream);
    return SGE_OK;
}

int sge_synchronize(sge_context* c) {
    if (!c) return SGE_ERR_INVALID;
    return syncAll(c);
}

int sge_context_set_option(sge_context* c, int option, int value) {
    if (c && option == SGE_OPT_FUSE_BLAS_REFIT) { c->fuseBlas = value < 0 ? 0 : (value > 2 ? 2 : value); return SGE_OK; }
    if (!c) return SGE_ERR_INVALID;
    switch (option) {
    case SGE_OPT_STORE_POSE_DEBUG: c->storePoseDebug = value != 0; break;
    case SGE_OPT_SKIN_LAYOUT:
        if (value != SGE_LAYOUT_PACKED && value != SGE_LAYOUT_PADDED16) { set_error("bad layout"); return SGE_ERR_INVALID; }
        c->skinLayout = value;
        break;
    case SGE_OPT_PROFILE: c->profile = value != 0; break;
    case SGE_OPT_OVERLAP_SKIN: {
        int rcs = syncAll(c);
        if (rcs != SGE_OK) return rcs;
        c->overlapRequested = value <= 0 ? 0 : (value >= 2 ? 2 : 1);
        applyOverlapOption(c);
        break;
    }
    case SGE_OPT_HEAVY_THRESHOLD: c->heavyThreshold = value; break; // (lists built for the old threshold are joined and rebuilt: launch_move)
    case SGE_OPT_PLACEMENT_PROBES: c->placementProbes = value; break;
    default: set_error("unknown option"); return SGE_ERR_INVALID;
    }
    return SGE_OK;
}

// ---- skeleton ----------------------------------------------------------------
int sge_skeleton_upload(sge_context* c, const sge_skeleton_desc* d) {
    if (c) { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    if (!c || !d || d->boneCount <= 0 || d->boneCount > SGE_MAX_BONES || !d->parent || !d->bindLocal || !d->invBindModel ||
        !d->restTranslation || !d->rawRestTranslation || !d->preRotationDegrees) {
        set_error("sge_skeleton_upload: bad argument");
        return SGE_ERR_INVALID;
    }
    const int B = d->boneCount;
    std::vector<int32_t> depth(B), chain;
    int maxDepth = 0;
    for (int i = 0; i < B; ++i) {
        int p = d->parent[i];
        if (p >= i) { set_error("sge_skeleton_upload: parent index must precede child (Skeleton.swift:189-203)"); return SGE_ERR_INVALID; }
        depth[i] = p < 0 ? 0 : depth[p] + 1;
        maxDepth = depth[i] > maxDepth ? depth[i] : maxDepth;
    }
    if (d->pelvisIndex >= B || d->leanIndex >= B) { set_error("semantic bone index out of range"); return SGE_ERR_INVALID; }
    if (d->leanIndex >= 0) {
        for (int b = d->leanIndex; b >= 0; b = d->parent[b]) chain.insert(chain.begin(), b);
    }
    if (chain.empty()) chain.push_back(0);
    std::vector<int32_t> path((size_t)B * (maxDepth + 1), 0);
    for (int i = 0; i < B; ++i) {
        int k = depth[i];
        for (int b = i; b >= 0; b = d->parent[b]) path[(size_t)i * (maxDepth + 1) + k--] = b;
    }
    std::vector<float> bind12((size_t)B * 12), pre12((size_t)B * 12);
    for (int i = 0; i < B; ++i) {
        const float* m = d->bindLocal + i * 16;
        for (int col = 0; col < 4; ++col)
            for (int r = 0; r < 3; ++r) bind12[(size_t)i * 12 + col * 3 + r] = m[col * 4 + r];
        Aff pr = rotationXYZDegrees(F3{d->preRotationDegrees[i * 3], d->preRotationDegrees[i * 3 + 1], d->preRotationDegrees[i * 3 + 2]});
        const F3 cols[4] = {pr.c0, pr.c1, pr.c2, pr.c3};
        for (int col = 0; col < 4; ++col) {
            pre12[(size_t)i * 12 + col * 3 + 0] = cols[col].x;
            pre12[(size_t)i * 12 + col * 3 + 1] = cols[col].y;
            pre12[(size_t)i * 12 + col * 3 + 2] = cols[col].z;
        }
    }
    (void)hipSetDevice(c->device);
    int rc;
    hipStream_t s = c->stream;
    if ((rc = upload(c->dParent, d->parent, (size_t)B * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dDepth, depth.data(), (size_t)B * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dLeanChain, chain.data(), chain.size() * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dPath, path.data(), path.size() * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dBindLocal, bind12.data(), bind12.size() * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dRestT, d->restTranslation, (size_t)B * 12, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dRawRestT, d->rawRestTranslation, (size_t)B * 12, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dPreRot, pre12.data(), pre12.size() * 4, s)) != SGE_OK) return rc;
    SGE_HIP(hipStreamSynchronize(s)); // host vectors go out of scope
    c->boneCount = B;
    c->hostSkeletonInvBind.assign(d->invBindModel, d->invBindModel + (size_t)B * 16);
    c->hostParent.assign(d->parent, d->parent + B);
    c->hostAnimated.assign(B, 0); // profiles are uploaded against a skeleton: a new skeleton starts without any
    DevSkeleton& sk = c->sk;
    sk.boneCount = B;
    sk.pelvisIndex = d->pelvisIndex;
    sk.leanIndex = d->leanIndex;
    sk.leanParent = d->leanIndex >= 0 ? d->parent[d->leanIndex] : -1;
    sk.maxDepth = maxDepth;
    sk.leanChainLen = (int)chain.size();
    sk.unitScale = d->unitScale;
    sk.parent = c->dParent.as<int32_t>();
    sk.depth = c->dDepth.as<int32_t>();
    sk.leanChain = c->dLeanChain.as<int32_t>();
    sk.path = c->dPath.as<int32_t>();
    sk.bindLocal = c->dBindLocal.as<float>();
    sk.restT = c->dRestT.as<float>();
    sk.rawRestT = c->dRawRestT.as<float>();
    sk.preRot = c->dPreRot.as<float>();
    for (int col = 0; col < 4; ++col)
        for (int r = 0; r < 3; ++r) sk.rootFix[col * 3 + r] = d->rootRotationFix[col * 4 + r];
    rc = refreshInvBind(c, nullptr, 0);
    if (rc != SGE_OK) return rc;
    if ((rc = rebuildPoseSlots(c)) != SGE_OK) return rc;
    SGE_HIP(hipStreamSynchronize(s));
    // palettes depend on boneCount
    if (c->crowd.count > 0) return sge_characters_resize(c, c->crowd.count);
    return SGE_OK;
}

// ---- motion profiles -----------------------------------------------------------
int sge_motion_profiles_upload(sge_context* c, const sge_motion_profile_desc* p, int32_t count) {
    if (c) { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    if (!c || !p || count <= 0 || count > SGE_MAX_PROFILES) { set_error("sge_motion_profiles_upload: bad argument"); return SGE_ERR_INVALID; }
    if (c->boneCount == 0) { set_error("upload the skeleton first"); return SGE_ERR_STATE; }
    const int B = c->boneCount;
    int stride = 1;
    for (int k = 0; k < count; ++k) {
        if (p[k].order < 1 || p[k].order > SGE_MAX_FOURIER_ORDER || !p[k].bonePresent || !p[k].coeffCount || !p[k].coeffs) {
            set_error("sge_motion_profiles_upload: bad profile (order must be 1..8)");
            return SGE_ERR_INVALID;
        }
        for (int i = 0; i < B * 6; ++i) {
            int cc = p[k].coeffCount[i];
            if (cc != SGE_AXIS_ABSENT) {
                if (cc > SGE_MAX_COEFFS) { set_error("coefficient count exceeds SGE_MAX_COEFFS"); return SGE_ERR_INVALID; }
                stride = cc > stride ? cc : stride;
            }
        }
    }
    // (one more full row of zeros at the end: pose_kernel reads 2 KMAX + 1 floats of every row it touches, whatever `stride` is)
    std::vector<float> coeffs((size_t)count * B * 6 * stride + SGE_MAX_COEFFS, 0.f);
    std::vector<uint8_t> cc((size_t)count * B * 6), present((size_t)count * B);
    for (int k = 0; k < count; ++k) {
        std::memcpy(&cc[(size_t)k * B * 6], p[k].coeffCount, (size_t)B * 6);
        std::memcpy(&present[(size_t)k * B], p[k].bonePresent, (size_t)B);
        for (int i = 0; i < B * 6; ++i)
            std::memcpy(&coeffs[((size_t)k * B * 6 + i) * stride], p[k].coeffs + (size_t)i * SGE_MAX_COEFFS, (size_t)stride * 4);
    }
    (void)hipSetDevice(c->device);
    int rc;
    if ((rc = upload(c->dCoeffs, coeffs.data(), coeffs.size() * 4, c->stream)) != SGE_OK) return rc;
    if ((rc = upload(c->dCoeffCount, cc.data(), cc.size(), c->stream)) != SGE_OK) return rc;
    if ((rc = upload(c->dBonePresent, present.data(), present.size(), c->stream)) != SGE_OK) return rc;
    SGE_HIP(hipStreamSynchronize(c->stream));
    DevProfiles& pf = c->prof;
    pf.count = count;
    pf.stride = stride;
    pf.maxOrder = 0;
    pf.nonRootTranslation = 0;
    c->hostAnimated.assign(B, 0);
    for (int k = 0; k < count; ++k) {
        pf.order[k] = p[k].order; pf.cycleRaw[k] = p[k].cycleDuration;
        pf.maxOrder = p[k].order > pf.maxOrder ? p[k].order : pf.maxOrder;
        for (int i = 0; i < B; ++i) {
            if (p[k].bonePresent[i]) c->hostAnimated[i] = 1;
            for (int a = 0; a < 3; ++a)
                if (i > 0 && p[k].bonePresent[i] && p[k].coeffCount[i * 6 + a] != SGE_AXIS_ABSENT) pf.nonRootTranslation |= 1u << k;
        }
    }
    pf.coeffs = c->dCoeffs.as<float>();
    pf.coeffCount = c->dCoeffCount.as<uint8_t>();
    pf.bonePresent = c->dBonePresent.as<uint8_t>();
    return rebuildPoseSlots(c);
}

// ---- skinned mesh ----------------------------------------------------------------
int sge_skinned_mesh_upload(sge_context* c, const sge_skinned_mesh_desc* d) {
    if (c) { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    if (!c || !d || d->vertexCount <= 0 || !d->positions || !d->normals || !d->tangents || !d->boneIndices || !d->boneWeights) {
        set_error("sge_skinned_mesh_upload: bad argument");
        return SGE_ERR_INVALID;
    }
    if (c->boneCount == 0) { set_error("upload the skeleton first"); return SGE_ERR_STATE; }
    const size_t V = (size_t)d->vertexCount;
    for (size_t i = 0; i < V * 4; ++i)
        if (d->boneIndices[i] >= c->boneCount) { set_error("bone index out of range"); return SGE_ERR_INVALID; }
    (void)hipSetDevice(c->device);
    int rc;
    hipStream_t s = c->stream;
    if ((rc = upload(c->dMeshPos, d->positions, V * 12, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dMeshNrm, d->normals, V * 12, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dMeshTan, d->tangents, V * 16, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dMeshIdx, d->boneIndices, V * 8, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dMeshWgt, d->boneWeights, V * 16, s)) != SGE_OK) return rc;
    if ((rc = refreshInvBind(c, d->invBindModel, d->invBindCount)) != SGE_OK) return rc;
    SGE_HIP(hipStreamSynchronize(s));
    c->mesh = DevMesh{d->vertexCount, c->dMeshPos.as<float>(), c->dMeshNrm.as<float>(), c->dMeshTan.as<float>(),
                      c->dMeshIdx.as<uint16_t>(), c->dMeshWgt.as<float>()};
    c->hostMeshPos.assign(d->positions, d->positions + V * 3);
    c->hostBlas = HostBlas{}; // a new mesh invalidates the acceleration structure built for the old one
    c->blas = DevBlas{};
    c->blasHasUVs = false;
    if (c->crowd.count > 0) return allocCrowdOutputs(c);
    return SGE_OK;
}

int sge_skinned_mesh_buffers(sge_context* c, void** p, void** n, void** t, void** i, void** w) {
    if (!c || c->mesh.vertexCount == 0) { set_error("no skinned mesh"); return SGE_ERR_STATE; }
    if (p) *p = c->dMeshPos.p; if (n) *n = c->dMeshNrm.p; if (t) *t = c->dMeshTan.p; if (i) *i = c->dMeshIdx.p; if (w) *w = c->dMeshWgt.p;
    return SGE_OK;
}

int sge_crowd_buffers(sge_context* c, void** pal, void** op, void** on, void** ot) {
    if (!c || c->crowd.count == 0) { set_error("no characters"); return SGE_ERR_STATE; }
    if (pal) *pal = c->dPalettes[c->palRead].p; // the buffer the newest pose stage wrote (and the newest skin launch reads)
    if (op) *op = c->dOutPos.p; if (on) *on = c->dOutNrm.p; if (ot) *ot = c->dOutTan.p;
    return SGE_OK;
}

int sge_crowd_palette_buffers(sge_context* c, void** d_palettes, int32_t* latest) {
    if (!c || !d_palettes) return SGE_ERR_INVALID;
    if (c->crowd.count == 0) { set_error("no characters"); return SGE_ERR_STATE; }
    d_palettes[0] = c->dPalettes[0].p; d_palettes[1] = c->dPalettes[1].p;
    if (latest) *latest = c->palRead;
    return SGE_OK;
}

// The other direction of sge_skin_wait: a consumer that reads the skinned streams on a stream of its own (asynchronously, no host
// synchronisation) must have finished before the NEXT skin launch overwrites them — in the reference the single command buffer per
// frame gives that for free (Renderer.swift:159, 224). Orders every later skin / refit launch of the context behind what
// `consumer_stream` holds so far.
int sge_skin_consumed(sge_context* c, void* consumer_stream) {
    if (!c) return SGE_ERR_INVALID;
    (void)hipSetDevice(c->device);
    hipStream_t s = consumer_stream ? reinterpret_cast<hipStream_t>(consumer_stream) : c->stream;
    if (s == c->stream && !c->overlapSkin) return SGE_OK; // same stream, serial order: already ordered
    SGE_HIP(hipEventRecord(c->evConsumed, s));
    if (c->skinStream) SGE_HIP(hipStreamWaitEvent(c->skinStream, c->evConsumed, 0));
    // the palettes are a consumer's input as well (jobs built from sge_crowd_buffers): pose(n+2) rewrites the buffer skin(n) read, and
    // with the pose launch on a stream of its own neither of the two waits above stands in front of it
    if (c->poseStream) SGE_HIP(hipStreamWaitEvent(c->poseStream, c->evConsumed, 0));
    if (s != c->stream) SGE_HIP(hipStreamWaitEvent(c->stream, c->evConsumed, 0));
    return SGE_OK;
}

// RTSkinningEncoder.encode enqueues on the CALLER's command buffer and what is enqueued behind it sees the skinned streams
// (RTSkinningEncoder.swift:27-56, RayTracingScene.swift:35-43). Under SGE_OPT_OVERLAP_SKIN the skin stage of sge_tick runs on the
// context's second stream: this is the ordering primitive that gives a consumer the same guarantee.
int sge_skin_wait(sge_context* c, void* consumer_stream) {
    if (!c) return SGE_ERR_INVALID;
    (void)hipSetDevice(c->device);
    hipStream_t s = consumer_stream ? reinterpret_cast<hipStream_t>(consumer_stream) : c->stream;
    if (s == c->stream) return joinSkin(c);
    // launches on the skin stream complete in order: the newest one's event covers them all
    if (c->skinPending[0] || c->skinPending[1]) SGE_HIP(hipStreamWaitEvent(s, c->evSkinDone[c->lastSkin], 0));
    // a pose launch on the pose stream (they complete in order too)
    if (c->posePending[0] || c->posePending[1]) SGE_HIP(hipStreamWaitEvent(s, c->evPosePiped[c->poseSlot], 0));
    // and whatever the context has enqueued on its main stream so far (a serial skin stage, pose, move)
    SGE_HIP(hipEventRecord(c->evMainMark, c->stream));
    SGE_HIP(hipStreamWaitEvent(s, c->evMainMark, 0));
    return SGE_OK;
}

int sge_skinning_encode(sge_context* c, void* d_outPositions, void* d_outNormals, void* d_outTangents, int32_t out_layout,
                        const sge_skinning_job* jobs, int32_t job_count) {
    if (!c || !d_outPositions || !d_outNormals || !d_outTangents || (job_count > 0 && !jobs)) { set_error("sge_skinning_encode: bad argument"); return SGE_ERR_INVALID; }
    (void)hipSetDevice(c->device);
    { int rcj = joinSkin(c); if (rcj != SGE_OK) return rcj; }
    for (int j = 0; j < job_count; ++j)
        if (jobs[j].vertexCount > 0 && (jobs[j].paletteCount <= 0 || jobs[j].paletteCount > SGE_MAX_BONES)) { set_error("paletteCount out of range"); return SGE_ERR_INVALID; }
    if (job_count > 1) {
        // the reference dispatches once per job (RTSkinningEncoder.swift:37-54); here the whole list is one launch over a
        // device-side job table, workgroups of up to vertsPerBlock vertices each
        long long totalVerts = 0;
        for (int j = 0; j < job_count; ++j) totalVerts += jobs[j].vertexCount > 0 ? jobs[j].vertexCount : 0;
        if (totalVerts == 0) return SGE_OK;
        long long vpb = (totalVerts / 8192 + 255) / 256 * 256;
        const int vertsPerBlock = (int)std::min<long long>(std::max<long long>(vpb, 256), 256 * 64);
        std::vector<SkinJobDev> table;
        std::vector<int2> blockJob;
        for (int j = 0; j < job_count; ++j) {
            const sge_skinning_job& J = jobs[j];
            if (J.vertexCount <= 0) continue;
            const int idx = (int)table.size();
            table.push_back(SkinJobDev{J.d_sourcePositions, J.d_sourceNormals, J.d_sourceTangents, J.d_sourceBoneIndices, J.d_sourceBoneWeights,
                                       reinterpret_cast<const float*>(J.d_palette), J.paletteCount, J.vertexCount, (long long)J.dstBaseVertex,
                                       J.sourceLayout == SGE_LAYOUT_PADDED16 ? 4 : 3, 0});
            for (int v = 0; v < J.vertexCount; v += vertsPerBlock) blockJob.push_back(make_int2(idx, v));
        }
        int rc;
        if ((rc = upload(c->dJobTable, table.data(), table.size() * sizeof(SkinJobDev), c->stream)) != SGE_OK) return rc;
        if ((rc = upload(c->dBlockJob, blockJob.data(), blockJob.size() * sizeof(int2), c->stream)) != SGE_OK) return rc;
        SGE_HIP(hipEventRecord(c->evTablesCopied, c->stream));
        {
            Bracket br(c, &c->evSkin);
            launch_skin_jobs(c->dJobTable.as<SkinJobDev>(), c->dBlockJob.as<int2>(), (int)blockJob.size(), vertsPerBlock, out_layout,
                             d_outPositions, d_outNormals, d_outTangents, c->stream);
        }
        SGE_HIP(hipGetLastError());
        SGE_HIP(hipEventSynchronize(c->evTablesCopied)); // the host tables go out of scope; the kernel itself stays asynchronous
        return SGE_OK;
    }
    for (int j = 0; j < job_count; ++j) {
        const sge_skinning_job& J = jobs[j];
        if (J.vertexCount <= 0) continue;
        SkinLaunch L{J.d_sourcePositions, J.d_sourceNormals, J.d_sourceTangents, J.d_sourceBoneIndices, J.d_sourceBoneWeights,
                     reinterpret_cast<const float*>(J.d_palette), J.paletteCount, J.vertexCount, 1, (long long)J.dstBaseVertex,
                     J.sourceLayout, out_layout, d_outPositions, d_outNormals, d_outTangents};
        Bracket br(c, &c->evSkin);
        launch_skin(L, c->stream);
    }
    SGE_HIP(hipGetLastError());
    return SGE_OK;
}

// ---- collision world ---------------------------------------------------------------
namespace {

// Device form of one set at its place in the merged arrays: triangles in slot order with the reference's
// (offset) triangle index and visit rank, wide nodes with child / range references rebased.
void deviceArrays(const HostCollision& hc, int triOffset, int wideOffset, std::vector<DevTri>& tris, std::vector<DevNode>& wide,
                  std::vector<DevMaterial>* mats) {
    const int T = (int)hc.layers.size();
    tris.resize(T);
    for (int slot = 0; slot < T; ++slot) {
        int t = hc.triOrder[slot];
        const uint32_t* ix = &hc.indices[(size_t)t * 3];
        const float *p0 = &hc.positions[(size_t)ix[0] * 3], *p1 = &hc.positions[(size_t)ix[1] * 3], *p2 = &hc.positions[(size_t)ix[2] * 3];
        tris[slot] = DevTri{p0[0], p0[1], p0[2], p1[0], p1[1], p1[2], p2[0], p2[1], p2[2], hc.layers[t], triOffset + t, triOffset + hc.rank[t]};
    }
    wide = hc.wide;
    for (size_t i = 0; i < wide.size(); ++i) {
        if (hc.wideBinary[i] < 0) continue;
        if (wide[i].b > 0) wide[i].a = ~((~wide[i].a) + triOffset); // triangle range: first slot
        else wide[i].a += wideOffset;                                // child wide node
    }
    if (mats) {
        mats->resize(T);
        for (int t = 0; t < T; ++t) (*mats)[t] = DevMaterial{hc.materials[t].muS, hc.materials[t].muK, hc.materials[t].flattenGround};
    }
}

void binaryNodes(const HostCollision& hc, std::vector<DevNode>& nodes) {
    nodes.resize(hc.nodes.size());
    for (size_t i = 0; i < hc.nodes.size(); ++i) {
        const HostBVHNode& n = hc.nodes[i];
        DevNode d{n.mn[0], n.mn[1], n.mn[2], n.mx[0], n.mx[1], n.mx[2], 0, 0};
        if (n.left < 0) { d.a = ~n.start; d.b = n.count; } else { d.a = n.left; d.b = n.right; }
        nodes[i] = d;
    }
}

struct MergedLayout { int Ts, Td, Ws, Wd; };
MergedLayout mergedLayout(const sge_context* c) {
    return MergedLayout{(int)c->hostCol.layers.size(), (int)c->hostDyn.layers.size(),
                        (int)(c->hostCol.wide.size() / kWideWidth), (int)(c->hostDyn.wide.size() / kWideWidth)};
}

// (Re)allocates the merged device world and uploads both sets.
int uploadCollisionAll(sge_context* c) {
    const MergedLayout L = mergedLayout(c);
    if (c->hostCol.maxDepth > 120 || c->hostDyn.maxDepth > 120) { set_error("BVH deeper than the traversal stack policy allows"); return SGE_ERR_CAPACITY; }
    // a traversal pops the newest node first, so at most 63 siblings stay pending per wide level (+ the other set's root)
    if (std::max(c->hostCol.wideLevels, c->hostDyn.wideLevels) * 63 + 2 > kTraversalStackCap) { set_error("triangle set too large for the traversal stack"); return SGE_ERR_CAPACITY; }
    // a sweep work item packs (ray slot << 27 | triangle slot) into 32 bits (sge_ccd.hip)
    if (c->hostCol.triOrder.size() + c->hostDyn.triOrder.size() >= ((size_t)1 << 27)) { set_error("more than 2^27 collision triangles"); return SGE_ERR_CAPACITY; }
    std::vector<DevTri> ts, td;
    std::vector<DevNode> ws, wd;
    std::vector<DevMaterial> ms, md;
    deviceArrays(c->hostCol, 0, 0, ts, ws, &ms);
    deviceArrays(c->hostDyn, L.Ts, L.Ws, td, wd, &md);
    ts.insert(ts.end(), td.begin(), td.end());
    ws.insert(ws.end(), wd.begin(), wd.end());
    ms.insert(ms.end(), md.begin(), md.end());
    (void)hipSetDevice(c->device);
    int rc;
    if ((rc = upload(c->dWide, ws.data(), ws.size() * sizeof(DevNode), c->stream)) != SGE_OK) return rc;
    if ((rc = upload(c->dTris, ts.data(), ts.size() * sizeof(DevTri), c->stream)) != SGE_OK) return rc;
    if ((rc = upload(c->dMaterials, ms.data(), ms.size() * sizeof(DevMaterial), c->stream)) != SGE_OK) return rc;
    std::vector<int> slotOfRank(ts.size());
    for (size_t slot = 0; slot < ts.size(); ++slot) slotOfRank[ts[slot].rank] = (int)slot;
    if ((rc = upload(c->dSlotOfRank, slotOfRank.data(), slotOfRank.size() * sizeof(int), c->stream)) != SGE_OK) return rc;
    std::vector<DevNode> bs, bd;
    binaryNodes(c->hostCol, bs);
    binaryNodes(c->hostDyn, bd);
    if ((rc = upload(c->dBinNodes[0], bs.data(), bs.size() * sizeof(DevNode), c->stream)) != SGE_OK) return rc;
    if ((rc = upload(c->dBinNodes[1], bd.data(), bd.size() * sizeof(DevNode), c->stream)) != SGE_OK) return rc;
    SGE_HIP(hipStreamSynchronize(c->stream));
    const int T = L.Ts + L.Td;
    c->col = DevCollision{(int)(c->hostCol.nodes.size() + c->hostDyn.nodes.size()), T, T > 0 ? 0 : -1, c->dWide.as<DevNode>(), L.Ws + L.Wd,
                          (L.Ts > 0 && L.Td > 0) ? L.Ws : -1, c->dTris.as<DevTri>(), c->dMaterials.as<DevMaterial>(),
                          {c->dBinNodes[0].as<DevNode>(), c->dBinNodes[1].as<DevNode>()}, {c->hostCol.root, c->hostDyn.root}, {0, L.Ts},
                          c->dSlotOfRank.as<int>()};
    return SGE_OK;
}

// After updateTransforms: topology and sizes are unchanged, only this set's triangles and wide bounds are rewritten.
int uploadCollisionSet(sge_context* c, int set) {
    const MergedLayout L = mergedLayout(c);
    const HostCollision& hc = set == SGE_SET_STATIC ? c->hostCol : c->hostDyn;
    const int triOffset = set == SGE_SET_STATIC ? 0 : L.Ts, wideOffset = set == SGE_SET_STATIC ? 0 : L.Ws;
    std::vector<DevTri> ts;
    std::vector<DevNode> ws;
    deviceArrays(hc, triOffset, wideOffset, ts, ws, nullptr);
    (void)hipSetDevice(c->device);
    if (!ts.empty()) SGE_HIP(hipMemcpyAsync(c->dTris.as<DevTri>() + triOffset, ts.data(), ts.size() * sizeof(DevTri), hipMemcpyHostToDevice, c->stream));
    if (!ws.empty()) SGE_HIP(hipMemcpyAsync(c->dWide.as<DevNode>() + (size_t)wideOffset * kWideWidth, ws.data(), ws.size() * sizeof(DevNode), hipMemcpyHostToDevice, c->stream));
    std::vector<DevNode> bn;
    binaryNodes(hc, bn);
    if (!bn.empty()) SGE_HIP(hipMemcpyAsync(c->dBinNodes[set].p, bn.data(), bn.size() * sizeof(DevNode), hipMemcpyHostToDevice, c->stream));
    SGE_HIP(hipStreamSynchronize(c->stream));
    return SGE_OK;
}

int checkEntities(const sge_static_mesh_entity* ents, int32_t count) {
    for (int e = 0; e < count; ++e) {
        if (!ents[e].positions || !ents[e].indices || ents[e].vertexCount < 0 || ents[e].indexCount < 0) { set_error("bad entity"); return SGE_ERR_INVALID; }
        for (int i = 0; i < ents[e].indexCount; ++i)
            if (ents[e].indices[i] >= (uint32_t)ents[e].vertexCount) { set_error("index out of range"); return SGE_ERR_INVALID; }
    }
    return SGE_OK;
}

// Builds the new set beside the old one and swaps it in only when the merged world passes uploadCollisionAll's capacity checks:
// after a rejected rebuild the host arrays and the device world still describe the same (old) triangles.
int rebuildSet(sge_context* c, HostCollision& slot, const sge_static_mesh_entity* ents, int32_t count) {
    HostCollision fresh;
    fresh.rebuild(ents, count);
    std::swap(slot, fresh);
    const int rc = uploadCollisionAll(c);
    if (rc != SGE_OK) {
        std::swap(slot, fresh);
        // the capacity checks come before any device work; anything else may have left the device buffers half written
        if (rc != SGE_ERR_CAPACITY) (void)uploadCollisionAll(c);
    }
    return rc;
}
} // namespace

int sge_collision_rebuild_static(sge_context* c, const sge_static_mesh_entity* ents, int32_t count) {
    if (c) { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    if (!c || count < 0 || (count > 0 && !ents)) { set_error("sge_collision_rebuild_static: bad argument"); return SGE_ERR_INVALID; }
    int rc = checkEntities(ents, count);
    if (rc != SGE_OK) return rc;
    return rebuildSet(c, c->hostCol, ents, count);
}

int sge_collision_rebuild_dynamic(sge_context* c, const sge_static_mesh_entity* ents, int32_t count) {
    if (c) { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    if (!c || count < 0 || (count > 0 && !ents)) { set_error("sge_collision_rebuild_dynamic: bad argument"); return SGE_ERR_INVALID; }
    int rc = checkEntities(ents, count);
    if (rc != SGE_OK) return rc;
    return rebuildSet(c, c->hostDyn, ents, count);
}

int sge_collision_update_transforms(sge_context* c, int32_t set, const int32_t* entities, const float* matrices, int32_t n) {
    if (!c || (set != SGE_SET_STATIC && set != SGE_SET_DYNAMIC) || n < 0 || (n > 0 && (!entities || !matrices))) {
        set_error("sge_collision_update_transforms: bad argument");
        return SGE_ERR_INVALID;
    }
    { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; } // queries in flight read the arrays about to change
    HostCollision& hc = set == SGE_SET_STATIC ? c->hostCol : c->hostDyn;
    if (hc.updateTransforms(entities, matrices, n) == 0) return SGE_OK;
    return uploadCollisionSet(c, set);
}

int sge_mesh_world_aabb(const float* positions, int32_t count, const float* M, float* outMin, float* outMax) {
    if (!positions || count <= 0 || !M || !outMin || !outMax) { set_error("sge_mesh_world_aabb: empty mesh"); return SGE_ERR_INVALID; }
    float mn[3] = {kFloatMax, kFloatMax, kFloatMax}, mx[3] = {-kFloatMax, -kFloatMax, -kFloatMax};
    for (int i = 0; i < count; ++i) {
        const float x = positions[i * 3], y = positions[i * 3 + 1], z = positions[i * 3 + 2];
        for (int r = 0; r < 3; ++r) { // simd_mul(modelMatrix, (p,1)).xyz, then simd_min / simd_max
            float w = ((M[r] * x + M[4 + r] * y) + M[8 + r] * z) + M[12 + r] * 1.0f;
            mn[r] = fminf(mn[r], w); mx[r] = fmaxf(mx[r], w);
        }
    }
    for (int r = 0; r < 3; ++r) { outMin[r] = mn[r]; outMax[r] = mx[r]; }
    return SGE_OK;
}

int sge_platforms_upload(sge_context* c, const sge_platform_state* p, int32_t count) {
    if (!c || count < 0 || count > SGE_MAX_PLATFORMS || (count > 0 && !p)) { set_error("sge_platforms_upload: bad argument (at most SGE_MAX_PLATFORMS)"); return SGE_ERR_INVALID; }
    { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    (void)hipSetDevice(c->device);
    int rc;
    if ((rc = upload(c->dPlatforms, p, (size_t)count * sizeof(*p), c->stream)) != SGE_OK) return rc;
    SGE_HIP(hipStreamSynchronize(c->stream));
    c->platformCount = count;
    return SGE_OK;
}

int sge_raycast_batch(sge_context* c, const sge_ray_query* q, int32_t count, sge_raycast_hit* out) {
    if (!c || count < 0 || (count > 0 && (!q || !out))) { set_error("sge_raycast_batch: bad argument"); return SGE_ERR_INVALID; }
    if (count == 0) return SGE_OK;
    (void)hipSetDevice(c->device);
    int rc;
    if ((rc = upload(c->dRayQueries, q, (size_t)count * sizeof(*q), c->stream)) != SGE_OK) return rc;
    if ((rc = c->dRayOut.alloc((size_t)count * sizeof(*out))) != SGE_OK) return rc;
    launch_raycast_queries(c->col, c->dRayQueries.as<sge_ray_query>(), count, c->dRayOut.as<sge_raycast_hit>(), c->stream);
    SGE_HIP(hipGetLastError());
    SGE_HIP(hipMemcpyAsync(out, c->dRayOut.p, (size_t)count * sizeof(*out), hipMemcpyDeviceToHost, c->stream));
    SGE_HIP(hipStreamSynchronize(c->stream));
    return SGE_OK;
}

int sge_collision_counts_set(sge_context* c, int32_t set, int32_t* v, int32_t* t, int32_t* n) {
    if (!c || (set != SGE_SET_STATIC && set != SGE_SET_DYNAMIC)) return SGE_ERR_INVALID;
    const HostCollision& hc = set == SGE_SET_STATIC ? c->hostCol : c->hostDyn;
    if (v) *v = (int32_t)(hc.positions.size() / 3);
    if (t) *t = (int32_t)hc.layers.size();
    if (n) *n = (int32_t)hc.nodes.size();
    return SGE_OK;
}
int sge_collision_counts(sge_context* c, int32_t* v, int32_t* t, int32_t* n) { return sge_collision_counts_set(c, SGE_SET_STATIC, v, t, n); }

int sge_collision_copy(sge_context* c, float* positions, uint32_t* indices, float* aabbs, sge_bvh_node* nodes,
                       int32_t* triOrder, int32_t* triLeaf) {
    return sge_collision_copy_set(c, SGE_SET_STATIC, positions, indices, aabbs, nodes, triOrder, triLeaf);
}

int sge_collision_copy_set(sge_context* c, int32_t set, float* positions, uint32_t* indices, float* aabbs, sge_bvh_node* nodes,
                           int32_t* triOrder, int32_t* triLeaf) {
    if (!c || (set != SGE_SET_STATIC && set != SGE_SET_DYNAMIC)) return SGE_ERR_INVALID;
    const HostCollision& hc = set == SGE_SET_STATIC ? c->hostCol : c->hostDyn;
    if (positions) std::memcpy(positions, hc.positions.data(), hc.positions.size() * 4);
    if (indices) std::memcpy(indices, hc.indices.data(), hc.indices.size() * 4);
    if (aabbs) std::memcpy(aabbs, hc.aabbs.data(), hc.aabbs.size() * 4);
    if (nodes)
        for (size_t i = 0; i < hc.nodes.size(); ++i) {
            const HostBVHNode& b = hc.nodes[i];
            nodes[i] = sge_bvh_node{{b.mn[0], b.mn[1], b.mn[2]}, {b.mx[0], b.mx[1], b.mx[2]}, b.left, b.right, b.start, b.count, b.parent};
        }
    if (triOrder) std::memcpy(triOrder, hc.triOrder.data(), hc.triOrder.size() * 4);
    if (triLeaf) std::memcpy(triLeaf, hc.triLeaf.data(), hc.triLeaf.size() * 4);
    return SGE_OK;
}

int sge_capsule_cast_batch(sge_context* c, const sge_capsule_query* q, int32_t count, sge_capsule_cast_hit* out) {
    if (!c || count < 0 || (count > 0 && (!q || !out))) { set_error("sge_capsule_cast_batch: bad argument"); return SGE_ERR_INVALID; }
    if (count == 0) return SGE_OK;
    (void)hipSetDevice(c->device);
    int rc;
    if ((rc = upload(c->dQueries, q, (size_t)count * sizeof(*q), c->stream)) != SGE_OK) return rc;
    if ((rc = c->dCastOut.alloc((size_t)count * sizeof(*out))) != SGE_OK) return rc;
    launch_cast_queries(c->col, c->dQueries.as<sge_capsule_query>(), count, c->dCastOut.as<sge_capsule_cast_hit>(),
                        c->dStats.as<unsigned long long>(), c->stream);
    SGE_HIP(hipGetLastError());
    SGE_HIP(hipMemcpyAsync(out, c->dCastOut.p, (size_t)count * sizeof(*out), hipMemcpyDeviceToHost, c->stream));
    SGE_HIP(hipStreamSynchronize(c->stream));
    return SGE_OK;
}

int sge_capsule_overlap_all_batch(sge_context* c, const sge_capsule_query* q, int32_t count, int32_t max_hits,
                                  sge_capsule_overlap_hit* out, int32_t* out_counts) {
    if (!c || count < 0 || max_hits < 1 || max_hits > SGE_MAX_OVERLAP_HITS || (count > 0 && (!q || !out || !out_counts))) {
        set_error("sge_capsule_overlap_all_batch: bad argument");
        return SGE_ERR_INVALID;
    }
    if (count == 0) return SGE_OK;
    (void)hipSetDevice(c->device);
    int rc;
    if ((rc = upload(c->dQueries, q, (size_t)count * sizeof(*q), c->stream)) != SGE_OK) return rc;
    if ((rc = c->dOverlapOut.alloc((size_t)count * max_hits * sizeof(*out))) != SGE_OK) return rc;
    if ((rc = c->dCounts.alloc((size_t)count * 4)) != SGE_OK) return rc;
    launch_overlap_queries(c->col, c->dQueries.as<sge_capsule_query>(), count, max_hits, c->dOverlapOut.as<sge_capsule_overlap_hit>(),
                           c->dCounts.as<int32_t>(), c->dStats.as<unsigned long long>(), c->stream);
    SGE_HIP(hipGetLastError());
    SGE_HIP(hipMemcpyAsync(out, c->dOverlapOut.p, (size_t)count * max_hits * sizeof(*out), hipMemcpyDeviceToHost, c->stream));
    SGE_HIP(hipMemcpyAsync(out_counts, c->dCounts.p, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream));
    SGE_HIP(hipStreamSynchronize(c->stream));
    return SGE_OK;
}

int sge_capsule_overlap_batch(sge_context* c, const sge_capsule_query* q, int32_t count, sge_capsule_overlap_hit* out,
                              int32_t* out_found) {
    if (!c || count < 0 || (count > 0 && (!q || !out || !out_found))) { set_error("sge_capsule_overlap_batch: bad argument"); return SGE_ERR_INVALID; }
    if (count == 0) return SGE_OK;
    (void)hipSetDevice(c->device);
    int rc;
    if ((rc = upload(c->dQueries, q, (size_t)count * sizeof(*q), c->stream)) != SGE_OK) return rc;
    if ((rc = c->dOverlapOut.alloc((size_t)count * sizeof(*out))) != SGE_OK) return rc;
    if ((rc = c->dCounts.alloc((size_t)count * 4)) != SGE_OK) return rc;
    launch_overlap_deepest_queries(c->col, c->dQueries.as<sge_capsule_query>(), count, c->dOverlapOut.as<sge_capsule_overlap_hit>(),
                                   c->dCounts.as<int32_t>(), c->dStats.as<unsigned long long>(), c->stream);
    SGE_HIP(hipGetLastError());
    SGE_HIP(hipMemcpyAsync(out, c->dOverlapOut.p, (size_t)count * sizeof(*out), hipMemcpyDeviceToHost, c->stream));
    SGE_HIP(hipMemcpyAsync(out_found, c->dCounts.p, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream));
    SGE_HIP(hipStreamSynchronize(c->stream));
    return SGE_OK;
}

// ---- characters ----------------------------------------------------------------------
int sge_characters_resize(sge_context* c, int32_t count) {
    if (!c || count < 0) { set_error("sge_characters_resize: bad argument"); return SGE_ERR_INVALID; }
    (void)hipSetDevice(c->device);
    { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    const size_t N = (size_t)count, B = (size_t)c->boneCount;
    int rc;
#define SGE_ZALLOC(buf, bytes) if ((rc = (buf).alloc(bytes)) != SGE_OK) return rc; if ((bytes) > 0) SGE_HIP(hipMemsetAsync((buf).p, 0, (bytes), c->stream));
    SGE_ZALLOC(c->dBodies, N * sizeof(sge_body_state));
    SGE_ZALLOC(c->dParams, N * sizeof(sge_controller_params));
    SGE_ZALLOC(c->dCtrl, N * sizeof(sge_controller_state));
    SGE_ZALLOC(c->dIntents, N * sizeof(sge_move_intent));
    SGE_ZALLOC(c->dLoco, N * sizeof(sge_locomotion_state));
    SGE_ZALLOC(c->dActions, N * sizeof(sge_action_state));
    SGE_ZALLOC(c->dPalettes[0], N * B * 64);
    SGE_ZALLOC(c->dPalettes[1], N * B * 64);
    SGE_ZALLOC(c->dPoseIn[0], N * sizeof(PoseInput));
    SGE_ZALLOC(c->dPoseIn[1], N * sizeof(PoseInput));
    c->palRead = 0;
    c->listsValid = false;
    SGE_ZALLOC(c->dMoveScratch, N * (size_t)kMoveScratchBytes);
    SGE_ZALLOC(c->dCost, N * sizeof(int));
    SGE_ZALLOC(c->dHint, N);
    SGE_ZALLOC(c->dHeavyFlags, N);
    SGE_ZALLOC(c->dLists, 2 * N * sizeof(int));
    SGE_ZALLOC(c->dListCounts, 4 * sizeof(int));
    SGE_ZALLOC(c->dOrderHist, 64 * sizeof(int));
    if (c->storePoseDebug) { SGE_ZALLOC(c->dPoseModel, N * B * 64); SGE_ZALLOC(c->dPoseLocal, N * B * 64); }
#undef SGE_ZALLOC
    c->crowd = DevCrowd{count, c->dBodies.as<sge_body_state>(), c->dParams.as<sge_controller_params>(),
                        c->dCtrl.as<sge_controller_state>(), c->dIntents.as<sge_move_intent>(),
                        c->dLoco.as<sge_locomotion_state>(), c->dActions.as<sge_action_state>(), c->dPalettes[0].as<float>(),
                        c->storePoseDebug ? c->dPoseModel.as<float>() : nullptr, c->storePoseDebug ? c->dPoseLocal.as<float>() : nullptr,
                        nullptr};
    if (c->mesh.vertexCount > 0 && (rc = allocCrowdOutputs(c)) != SGE_OK) return rc;
    if ((rc = ensureBlasBuffers(c)) != SGE_OK) return rc;
    SGE_HIP(hipStreamSynchronize(c->stream));
    return SGE_OK;
}

#define SGE_RANGE_CHECK() \
    if (!c || first < 0 || count < 0 || first + count > c->crowd.count) { set_error("character range out of bounds"); return SGE_ERR_INVALID; }

int sge_characters_upload(sge_context* c, int32_t first, int32_t count, const sge_body_state* b, const sge_controller_params* p,
                          const sge_controller_state* cs, const sge_move_intent* in, const sge_locomotion_state* l,
                          const sge_action_state* a) {
    SGE_RANGE_CHECK();
    (void)hipSetDevice(c->device);
    if (l) {
        for (int i = 0; i < count; ++i) {
            for (int k = 0; k < 4; ++k)
                if ((l[i].flags & SGE_LOCO_PRESENT) && (l[i].profile[k] < 0 || l[i].profile[k] >= c->prof.count)) { set_error("locomotion profile index out of range"); return SGE_ERR_INVALID; }
            if ((l[i].flags & SGE_MOTION_PRESENT) && (l[i].motionProfile < 0 || l[i].motionProfile >= c->prof.count)) { set_error("motion profile index out of range"); return SGE_ERR_INVALID; }
            if ((l[i].flags & SGE_LOCO_PRESENT) && ((unsigned)l[i].state > 3u || (unsigned)l[i].fromState > 3u)) { set_error("locomotion state out of range"); return SGE_ERR_INVALID; }
        }
    }
    if (a)
        for (int i = 0; i < count; ++i)
            if ((a[i].flags & SGE_ACTION_PRESENT) && (a[i].profile < 0 || a[i].profile >= c->prof.count)) { set_error("action profile index out of range"); return SGE_ERR_INVALID; }
    { int rcp = joinPose(c); if (rcp != SGE_OK) return rcp; } // a pose launch on the pose stream owns the animation states meanwhile
    hipStream_t s = c->stream;
#define SGE_UP(ptr, buf, T) if (ptr) SGE_HIP(hipMemcpyAsync((buf).template as<T>() + first, ptr, (size_t)count * sizeof(T), hipMemcpyHostToDevice, s));
    SGE_UP(b, c->dBodies, sge_body_state) SGE_UP(p, c->dParams, sge_controller_params) SGE_UP(cs, c->dCtrl, sge_controller_state)
    SGE_UP(in, c->dIntents, sge_move_intent) SGE_UP(l, c->dLoco, sge_locomotion_state) SGE_UP(a, c->dActions, sge_action_state)
#undef SGE_UP
    SGE_HIP(hipStreamSynchronize(s));
    return SGE_OK;
}

int sge_characters_download(sge_context* c, int32_t first, int32_t count, sge_body_state* b, sge_controller_params* p,
                            sge_controller_state* cs, sge_move_intent* in, sge_locomotion_state* l, sge_action_state* a) {
    SGE_RANGE_CHECK();
    (void)hipSetDevice(c->device);
    { int rcp = joinPose(c); if (rcp != SGE_OK) return rcp; }
    hipStream_t s = c->stream;
#define SGE_DN(ptr, buf, T) if (ptr) SGE_HIP(hipMemcpyAsync(ptr, (buf).template as<T>() + first, (size_t)count * sizeof(T), hipMemcpyDeviceToHost, s));
    SGE_DN(b, c->dBodies, sge_body_state) SGE_DN(p, c->dParams, sge_controller_params) SGE_DN(cs, c->dCtrl, sge_controller_state)
    SGE_DN(in, c->dIntents, sge_move_intent) SGE_DN(l, c->dLoco, sge_locomotion_state) SGE_DN(a, c->dActions, sge_action_state)
#undef SGE_DN
    SGE_HIP(hipStreamSynchronize(s));
    return SGE_OK;
}

int sge_palettes_download(sge_context* c, int32_t first, int32_t count, float* palette, float* model, float* local) {
    SGE_RANGE_CHECK();
    if ((model || local) && !c->crowd.poseModel) { set_error("model/local need SGE_OPT_STORE_POSE_DEBUG set before sge_characters_resize"); return SGE_ERR_STATE; }
    (void)hipSetDevice(c->device);
    const size_t B = (size_t)c->boneCount, off = (size_t)first * B * 16, n = (size_t)count * B * 64;
    { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    if (palette) SGE_HIP(hipMemcpyAsync(palette, c->dPalettes[c->palRead].as<float>() + off, n, hipMemcpyDeviceToHost, c->stream));
    if (model) SGE_HIP(hipMemcpyAsync(model, c->dPoseModel.as<float>() + off, n, hipMemcpyDeviceToHost, c->stream));
    if (local) SGE_HIP(hipMemcpyAsync(local, c->dPoseLocal.as<float>() + off, n, hipMemcpyDeviceToHost, c->stream));
    SGE_HIP(hipStreamSynchronize(c->stream));
    return SGE_OK;
}

int sge_skinned_download(sge_context* c, int64_t first_vertex, int64_t vertex_count, float* positions, float* normals, float* tangents) {
    if (!c || first_vertex < 0 || vertex_count < 0 ||
        (size_t)(first_vertex + vertex_count) > (size_t)c->crowd.count * (size_t)c->mesh.vertexCount) { set_error("vertex range out of bounds"); return SGE_ERR_INVALID; }
    (void)hipSetDevice(c->device);
    const bool padded = c->outLayoutAllocated == SGE_LAYOUT_PADDED16;
    const size_t n = (size_t)vertex_count, f = (size_t)first_vertex;
    { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    hipStream_t s = c->stream;
    if (!padded) {
        if (positions) SGE_HIP(hipMemcpyAsync(positions, c->dOutPos.as<float>() + f * 3, n * 12, hipMemcpyDeviceToHost, s));
        if (normals) SGE_HIP(hipMemcpyAsync(normals, c->dOutNrm.as<float>() + f * 3, n * 12, hipMemcpyDeviceToHost, s));
    } else { // strided 16 B -> packed 12 B
        if (positions) SGE_HIP(hipMemcpy2DAsync(positions, 12, c->dOutPos.as<float>() + f * 4, 16, 12, n, hipMemcpyDeviceToHost, s));
        if (normals) SGE_HIP(hipMemcpy2DAsync(normals, 12, c->dOutNrm.as<float>() + f * 4, 16, 12, n, hipMemcpyDeviceToHost, s));
    }
    if (tangents) SGE_HIP(hipMemcpyAsync(tangents, c->dOutTan.as<float>() + f * 4, n * 16, hipMemcpyDeviceToHost, s));
    SGE_HIP(hipStreamSynchronize(s));
    return SGE_OK;
}

// ---- asynchronous World synchronisation ------------------------------------------------------
namespace {
// the arrays of sge_state_*: bit k of `which` <-> row k
constexpr int kStateArrays = 5;
constexpr size_t kStateBytes[kStateArrays] = {sizeof(sge_body_state), sizeof(sge_controller_state), sizeof(sge_locomotion_state),
                                              sizeof(sge_action_state), sizeof(sge_move_intent)};
// 16-byte chunks of every struct that the move / separation stage writes, and that the animation stages write. A body is both sides':
// words 0..15 (position, velocity, rotation) and the body type are the move stage's, transformRotation (chunk 4) is the pose
// stage's write-back — with the pose launch on a stream of its own the two halves of one step are complete at different places.
constexpr uint32_t kMoveSideChunks[kStateArrays] = {0x2Fu, 0xFFu, 0u, 0u, 0x3u};
constexpr uint32_t kPoseSideChunks[kStateArrays] = {0x10u, 0u, 0x3Fu, 0x3u, 0u};
static_assert(sizeof(sge_body_state) == 6 * 16 && offsetof(sge_body_state, transformRotation) == 4 * 16, "chunk masks");
static_assert(sizeof(sge_controller_state) == 8 * 16 && sizeof(sge_locomotion_state) == 6 * 16 && sizeof(sge_action_state) == 2 * 16 &&
              sizeof(sge_move_intent) == 2 * 16, "chunk masks");

struct SnapArray { const uint4* src; uint4* dst; int chunks; uint32_t mask; };
struct SnapLaunch { SnapArray a[kStateArrays]; int first, count; };
// One thread per 16-byte chunk, blockIdx.y = array: dst is [count] structs, src the context's array of the whole crowd.
__global__ void state_snapshot_kernel(SnapLaunch S) {
    const SnapArray A = S.a[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.count * A.chunks) return;
    const int ch = i / A.chunks, k = i - ch * A.chunks;
    if (!((A.mask >> k) & 1u)) return;
    A.dst[i] = A.src[(size_t)(S.first + ch) * A.chunks + k];
}

void* stateArray(sge_context* c, int k) {
    switch (k) {
    case 0: return c->dBodies.p; case 1: return c->dCtrl.p; case 2: return c->dLoco.p; case 3: return c->dActions.p; default: return c->dIntents.p;
    }
}
// offsets of the requested arrays in one packed block, [count] structs each; returns the block's size
size_t stateLayout(uint32_t which, int count, size_t off[kStateArrays]) {
    size_t total = 0;
    for (int k = 0; k < kStateArrays; ++k) {
        off[k] = total;
        if (which & (1u << k)) total += (size_t)count * kStateBytes[k];
    }
    return total;
}
int pinnedReserve(void*& host, size_t& have, size_t need) {
    if (need <= have && host) return SGE_OK;
    if (host) (void)hipHostFree(host);
    host = nullptr; have = 0;
    if (need == 0) return SGE_OK;
    need = (need + 4095) / 4096 * 4096;
    SGE_HIP(hipHostMalloc(&host, need, hipHostMallocDefault));
    have = need;
    return SGE_OK;
}
void fillView(sge_state_view* v, char* base, uint32_t which, int first, int count, int ticket, const size_t off[kStateArrays]) {
    *v = sge_state_view{};
    v->first = first; v->count = count; v->which = which; v->ticket = ticket;
    if (!base) return;
    if (which & SGE_STATE_BODIES) v->bodies = reinterpret_cast<sge_body_state*>(base + off[0]);
    if (which & SGE_STATE_CONTROLLERS) v->controllers = reinterpret_cast<sge_controller_state*>(base + off[1]);
    if (which & SGE_STATE_LOCOMOTION) v->locomotion = reinterpret_cast<sge_locomotion_state*>(base + off[2]);
    if (which & SGE_STATE_ACTIONS) v->actions = reinterpret_cast<sge_action_state*>(base + off[3]);
    if (which & SGE_STATE_INTENTS) v->intents = reinterpret_cast<sge_move_intent*>(base + off[4]);
}
} // namespace

int sge_state_pull_async(sge_context* c, uint32_t which, int32_t first, int32_t count, int32_t* ticket) {
    if (!c || !ticket || which == 0 || (which & ~(uint32_t)SGE_STATE_WORLD)) { set_error("sge_state_pull_async: bad argument (which = SGE_STATE_BODIES | CONTROLLERS | LOCOMOTION | ACTIONS)"); return SGE_ERR_INVALID; }
    if (count == 0) { first = 0; count = c->crowd.count; }
    SGE_RANGE_CHECK();
    (void)hipSetDevice(c->device);
    sge_context::PullSlot& P = c->pull[c->pullTickets & 1];
    if (P.inFlight) { SGE_HIP(hipEventSynchronize(P.evLanded)); P.inFlight = false; } // the pull before last: its memory is taken over
    const size_t total = stateLayout(which, count, P.off);
    int rc;
    if ((rc = P.stage.alloc(total)) != SGE_OK || (rc = pinnedReserve(P.host, P.hostBytes, total)) != SGE_OK) return rc;
    P.ticket = c->pullTickets++;
    P.which = which; P.first = first; P.count = count;
    *ticket = P.ticket;
    if (total == 0) return SGE_OK;
    // The animation stages' half of the step: where the newest pose launch ran. A launch on the pose stream is followed there (the
    // next tick's pose launch queues behind the snapshot, its move stage does not wait for either); otherwise everything is the
    // main stream's and one launch takes both halves.
    const bool piped = c->posePending[0] || c->posePending[1];
    SnapLaunch M{}, A{};
    M.first = A.first = first; M.count = A.count = count;
    int nm = 0, na = 0;
    for (int k = 0; k < 4; ++k) {
        if (!(which & (1u << k))) continue;
        const uint32_t ms = piped ? kMoveSideChunks[k] : (kMoveSideChunks[k] | kPoseSideChunks[k]), as = piped ? kPoseSideChunks[k] : 0u;
        const int chunks = (int)(kStateBytes[k] / 16);
        uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<char*>(P.stage.p) + P.off[k]);
        if (ms) M.a[nm++] = SnapArray{reinterpret_cast<const uint4*>(stateArray(c, k)), dst, chunks, ms};
        if (as) A.a[na++] = SnapArray{reinterpret_cast<const uint4*>(stateArray(c, k)), dst, chunks, as};
    }
    const int blocks = (count * 8 + 255) / 256;
    if (nm) hipLaunchKernelGGL(state_snapshot_kernel, dim3(blocks, nm), dim3(256), 0, c->stream, M);
    SGE_HIP(hipEventRecord(P.evMain, c->stream));
    SGE_HIP(hipStreamWaitEvent(c->copyStream, P.evMain, 0));
    if (na) {
        hipLaunchKernelGGL(state_snapshot_kernel, dim3(blocks, na), dim3(256), 0, c->poseStream, A);
        SGE_HIP(hipEventRecord(P.evPose, c->poseStream));
        SGE_HIP(hipStreamWaitEvent(c->copyStream, P.evPose, 0));
    }
    SGE_HIP(hipGetLastError());
    SGE_HIP(hipMemcpyAsync(P.host, P.stage.p, total, hipMemcpyDeviceToHost, c->copyStream));
    SGE_HIP(hipEventRecord(P.evLanded, c->copyStream));
    P.inFlight = true;
    return SGE_OK;
}

namespace {
sge_context::PullSlot* findPull(sge_context* c, int32_t ticket) {
    for (auto& P : c->pull) if (P.ticket == ticket && ticket >= 0) return &P;
    return nullptr;
}
} // namespace

int sge_state_wait(sge_context* c, int32_t ticket, sge_state_view* view) {
    if (!c || !view) { set_error("sge_state_wait: bad argument"); return SGE_ERR_INVALID; }
    sge_context::PullSlot* P = findPull(c, ticket);
    if (!P) { set_error("sge_state_wait: unknown ticket (two pulls have been enqueued since, or none with this number)"); return SGE_ERR_STATE; }
    (void)hipSetDevice(c->device);
    if (P->inFlight) { SGE_HIP(hipEventSynchronize(P->evLanded)); P->inFlight = false; }
    fillView(view, reinterpret_cast<char*>(P->host), P->which, P->first, P->count, P->ticket, P->off);
    return SGE_OK;
}

int sge_state_poll(sge_context* c, int32_t ticket) {
    if (!c) return SGE_ERR_INVALID;
    sge_context::PullSlot* P = findPull(c, ticket);
    if (!P) { set_error("sge_state_poll: unknown ticket"); return SGE_ERR_STATE; }
    if (!P->inFlight) return SGE_OK;
    (void)hipSetDevice(c->device);
    const hipError_t e = hipEventQuery(P->evLanded);
    if (e == hipSuccess) return SGE_OK;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return SGE_ERR_NOT_READY; }
    return hip_fail(e, "hipEventQuery");
}

int sge_state_push_begin(sge_context* c, uint32_t which, int32_t first, int32_t count, sge_state_view* staging) {
    if (!c || !staging || which == 0 || (which & ~(uint32_t)(SGE_STATE_WORLD | SGE_STATE_INTENTS))) { set_error("sge_state_push_begin: bad argument"); return SGE_ERR_INVALID; }
    if (c->pushOpen) { set_error("sge_state_push_begin: the previous staging has not been committed"); return SGE_ERR_STATE; }
    if (count == 0) { first = 0; count = c->crowd.count; }
    SGE_RANGE_CHECK();
    (void)hipSetDevice(c->device);
    sge_context::PushSlot& S = c->push[c->pushSlot];
    if (S.inFlight) { SGE_HIP(hipEventSynchronize(S.evCopied)); S.inFlight = false; } // the push before last still reads this staging
    const size_t total = stateLayout(which, count, c->pushOff);
    int rc = pinnedReserve(S.host, S.hostBytes, total);
    if (rc != SGE_OK) return rc;
    c->pushWhich = which; c->pushFirst = first; c->pushCount = count; c->pushOpen = true;
    fillView(staging, reinterpret_cast<char*>(S.host), which, first, count, -1, c->pushOff);
    return SGE_OK;
}

int sge_state_push_commit(sge_context* c) {
    if (!c) return SGE_ERR_INVALID;
    if (!c->pushOpen) { set_error("sge_state_push_commit without sge_state_push_begin"); return SGE_ERR_STATE; }
    c->pushOpen = false;
    sge_context::PushSlot& S = c->push[c->pushSlot];
    const int count = c->pushCount, first = c->pushFirst;
    if (count == 0) return SGE_OK;
    if (first + count > c->crowd.count) { set_error("sge_state_push_commit: the crowd was resized under an open staging"); return SGE_ERR_STATE; }
    sge_state_view v;
    fillView(&v, reinterpret_cast<char*>(S.host), c->pushWhich, first, count, -1, c->pushOff);
    // the same index checks as sge_characters_upload: the kernels index the profile table with these
    if (v.locomotion)
        for (int i = 0; i < count; ++i) {
            const sge_locomotion_state& l = v.locomotion[i];
            for (int k = 0; k < 4; ++k)
                if ((l.flags & SGE_LOCO_PRESENT) && (l.profile[k] < 0 || l.profile[k] >= c->prof.count)) { set_error("locomotion profile index out of range"); return SGE_ERR_INVALID; }
            if ((l.flags & SGE_MOTION_PRESENT) && (l.motionProfile < 0 || l.motionProfile >= c->prof.count)) { set_error("motion profile index out of range"); return SGE_ERR_INVALID; }
            if ((l.flags & SGE_LOCO_PRESENT) && ((unsigned)l.state > 3u || (unsigned)l.fromState > 3u)) { set_error("locomotion state out of range"); return SGE_ERR_INVALID; }
        }
    if (v.actions)
        for (int i = 0; i < count; ++i)
            if ((v.actions[i].flags & SGE_ACTION_PRESENT) && (v.actions[i].profile < 0 || v.actions[i].profile >= c->prof.count)) { set_error("action profile index out of range"); return SGE_ERR_INVALID; }
    (void)hipSetDevice(c->device);
    // a pose launch on the pose stream owns the animation states and transformRotation meanwhile; intents are nobody's but the host's
    if (c->pushWhich & ~(uint32_t)SGE_STATE_INTENTS) { int rcp = joinPose(c); if (rcp != SGE_OK) return rcp; }
    const void* src[kStateArrays] = {v.bodies, v.controllers, v.locomotion, v.actions, v.intents};
    for (int k = 0; k < kStateArrays; ++k)
        if (src[k]) SGE_HIP(hipMemcpyAsync(reinterpret_cast<char*>(stateArray(c, k)) + (size_t)first * kStateBytes[k], src[k], (size_t)count * kStateBytes[k], hipMemcpyHostToDevice, c->stream));
    SGE_HIP(hipEventRecord(S.evCopied, c->stream));
    S.inFlight = true;
    c->pushSlot ^= 1;
    return SGE_OK;
}

// ---- the batched fixed step --------------------------------------------------------------
int sge_tick(sge_context* c, const sge_tick_desc* d) {
    if (!c || !d) { set_error("sge_tick: bad argument"); return SGE_ERR_INVALID; }
    int first = d->first, count = d->count;
    if (count == 0) { first = 0; count = c->crowd.count; }
    SGE_RANGE_CHECK();
    if (count == 0) return SGE_OK;
    (void)hipSetDevice(c->device);
    const uint32_t st = d->stages;
    // Overlap mode, a whole-crowd tick with move + pose + skin: the pose launch goes to the pose stream and reads the move stage's
    // results from the copy the move kernels write (PoseInput), so that the NEXT tick's move stage does not queue behind it.
    // (not with the separation stage, which moves bodies after the move kernels have written the copy)
    const bool pipedPose = c->pipelinePose && c->overlapSkin && c->poseStream && first == 0 && count == c->crowd.count &&
                           (st & (SGE_STAGE_INTENT | SGE_STAGE_GRAVITY | SGE_STAGE_MOVE)) && !(st & SGE_STAGE_SEPARATION) &&
                           (st & SGE_STAGE_POSE) && (st & SGE_STAGE_SKIN) && c->boneCount > 0 && c->prof.count > 0 &&
                           c->mesh.vertexCount > 0;
    const int poseSlot = c->poseSlot ^ 1;
    bool movedOnTwoStreams = false;
    if (st & (SGE_STAGE_INTENT | SGE_STAGE_GRAVITY | SGE_STAGE_MOVE)) {
        if ((st & SGE_STAGE_MOVE) && c->col.root < 0 && c->col.triCount != 0) { set_error("collision world not built"); return SGE_ERR_STATE; }
        if ((st & SGE_STAGE_AGENTS) && (st & SGE_STAGE_MOVE)) {
            if (!c->agents.all) { set_error("SGE_STAGE_AGENTS needs sge_agents_import"); return SGE_ERR_STATE; }
            Bracket br(c, &c->evAgents);
            int rc = buildAgentGrid(c);
            if (rc != SGE_OK) return rc;
        }
        MoveLaunch L{c->crowd, c->col, c->agents, d->dt, d->gravity[0], d->gravity[1], d->gravity[2], st, first, count,
                     c->dStats.as<unsigned long long>(), c->dMoveScratch.p, c->dPlatforms.as<sge_platform_state>(), c->platformCount,
                     c->dCost.as<int>(), getenv("SGE_NO_SPEC") ? nullptr : c->dHint.as<uint8_t>(), c->dLists.as<int>(), c->dListCounts.as<int>(), c->dHeavyFlags.as<uint8_t>(),
                     c->heavyThreshold, c->heavyCap, c->heavyStream, c->evClassified, c->evHeavyDone, c->hHeavyDemand, nullptr, nullptr, nullptr, nullptr, c->dOrderHist.as<int>(), 0, 0, 0, c->evListsReady, nullptr};
        // Grid of the multi-wave launch: the characters that asked for it in the newest step whose count has reached the host
        // (+ 50 % + 8), not the cap — every workgroup of that grid, the ones beyond the list included, has to find a CU with two
        // free places per SIMD. The list is cut to the grid (classify_kernel); whoever does not fit stays with the grouped launch,
        // which gives the same result. -1: nothing has come back yet.
        if (st & SGE_STAGE_MOVE) c->lastMoveCount = count;
        int capNow = c->heavyCap;
        if (c->hHeavyDemand) {
            const int demand = *(volatile int*)c->hHeavyDemand;
            if (demand >= 0) capNow = std::min(c->heavyCap, demand + demand / 2 + 8);
        }
        // lists built behind the previous step's move stage serve this one if nothing they depend on has changed; their cap is the grid
        const bool listsReady = c->listsValid && c->listsFirst == first && c->listsCount == count && c->listsThreshold == c->heavyThreshold;
        L.listsReady = listsReady ? 1 : 0;
        L.listsPending = c->listsValid ? 1 : 0; // a build enqueued behind an earlier move stage; launch_move joins it when it cannot use it
        L.heavyCap = listsReady ? c->listsCap : capNow;
        L.nextHeavyCap = capNow;
        if (c->waveProfOn) {
            if (c->dWaveProf.alloc((size_t)c->crowd.count * 3 * 64) != SGE_OK) return SGE_ERR_DEVICE;
            SGE_HIP(hipMemsetAsync(c->dWaveProf.p, 0, (size_t)c->crowd.count * 2 * 64, c->stream)); // (region 2 is cleared on the stream the pose launch takes)
            L.waveProf = c->dWaveProf.as<unsigned long long>();
        }
        if (!(st & SGE_STAGE_AGENTS) || !c->agents.grid) L.agents.all = nullptr;
        if (pipedPose) { // the copy pose(n-2) read must be free again
            if (c->posePending[poseSlot]) { SGE_HIP(hipStreamWaitEvent(c->stream, c->evPosePiped[poseSlot], 0)); c->posePending[poseSlot] = false; }
            L.crowd.poseIn = c->dPoseIn[poseSlot].as<PoseInput>();
        }
        {
            Bracket br(c, &c->evMove);
            const bool built = launch_move(L, c->stream);
            if (st & SGE_STAGE_MOVE) {
                c->listsValid = built;
                c->listsFirst = first; c->listsCount = count; c->listsThreshold = c->heavyThreshold; c->listsCap = capNow;
            }
        }
        // what the pose stream waits for: with the two-stream move stage its two "done" events (no packet of its own on the main
        // stream, whose next kernel is the next tick's part 0), otherwise a marker behind the stage
        if (pipedPose) {
            movedOnTwoStreams = (st & SGE_STAGE_MOVE) && c->heavyThreshold >= 0;
            if (!movedOnTwoStreams) SGE_HIP(hipEventRecord(c->evMoveDone, c->stream));
        }
    }
    if (st & SGE_STAGE_SEPARATION) { // AgentSeparationSystem: after the move stage, before the animation stages (DemoScene.swift:66-71)
        if (first != 0 || count != c->crowd.count) { set_error("SGE_STAGE_SEPARATION works on the whole crowd (first = 0, count = all)"); return SGE_ERR_INVALID; }
        // The pair loop is sequential over the WHOLE crowd in index order (Systems.swift:1906-2210): a context that holds one shard of
        // it (an imported snapshot that names agents of other contexts) cannot produce the reference's result, with or without a
        // halo — refused, not approximated.
        if (c->agents.all && (c->agents.selfOffset > 0 || c->agents.total > c->crowd.count)) {
            set_error("SGE_STAGE_SEPARATION needs the whole crowd in one context: this one holds a shard (sge_agents_import / _allgather named agents of other contexts)");
            return SGE_ERR_STATE;
        }
        if (c->col.root < 0 && c->col.triCount != 0) { set_error("collision world not built"); return SGE_ERR_STATE; }
        int rc;
        const size_t sepAgents = (size_t)std::max(c->crowd.count, (int)SGE_MAX_SEPARATION_AGENTS);
        if ((rc = c->dSepAgents.alloc(sepAgents * kSeparationAgentBytes)) != SGE_OK) return rc;
        if ((rc = c->dSepCounts.alloc(2 * sizeof(int))) != SGE_OK) return rc;
        if ((rc = c->dSepFlow.alloc(separationFlowBytes(c->crowd.count))) != SGE_OK) return rc;
        // A crowd whose agents get pushed further than a cell per pass (spawned on top of itself: 31,250 characters on the benchmark
        // scene) takes its candidates from 7 x 7 cells for the next 64 steps; the step that shows it first is redone by one wavefront
        // (results are the reference's either way)
        if (c->hSepFlags && (*(volatile int*)c->hSepFlags & 2)) { c->sepWideSteps = 64; c->hSepFlags[0] = 0; }
        const int reach = c->sepWideSteps > 0 ? 3 : 2;
        if (c->sepWideSteps > 0) c->sepWideSteps -= 1;
        launch_separation(c->crowd, c->col, c->separationIterations, c->separationMargin, c->separationHeightMargin, c->dSepAgents.p,
                          c->dSepCounts.as<int>(), c->dSepFlow.p, c->stream, reach, c->hSepFlags);
    }
    if (st & (SGE_STAGE_LOCOMOTION | SGE_STAGE_ACTION | SGE_STAGE_POSE | SGE_STAGE_WRITEBACK)) {
        if ((st & SGE_STAGE_POSE) && (c->boneCount == 0 || c->prof.count == 0)) { set_error("pose stage needs a skeleton and motion profiles"); return SGE_ERR_STATE; }
        hipStream_t ps = pipedPose ? c->poseStream : c->stream;
        if (!pipedPose) { int rcp = joinPose(c); if (rcp != SGE_OK) return rcp; } // the animation states are the pose stream's meanwhile
        if (st & SGE_STAGE_POSE) {
            const bool flip = c->overlapSkin && first == 0 && count == c->crowd.count;
            if (flip) { // write the buffer no skin launch newer than skin(n-1) reads
                const int f = c->palRead ^ 1;
                if (c->skinPending[f]) {
                    SGE_HIP(hipStreamWaitEvent(ps, c->evSkinDone[f], 0));
                    if (!pipedPose) c->skinPending[f] = false; // (piped: the main stream itself has not joined that launch)
                }
                c->palRead = f;
            } else { // palettes are rewritten in place
                int rcj = joinSkin(c);
                if (rcj != SGE_OK) return rcj;
            }
            c->crowd.palettes = c->dPalettes[c->palRead].as<float>();
        }
        PoseLaunch L{c->crowd, c->sk, c->prof, d->dt, st, first, count,
                     c->waveProfOn && c->dWaveProf.p ? c->dWaveProf.as<unsigned long long>() + (size_t)c->crowd.count * 2 * 8 : nullptr};
        if (pipedPose) {
            if (movedOnTwoStreams) {
                SGE_HIP(hipStreamWaitEvent(ps, c->evClassified, 0)); // part 0 + the multi-wave launch (main stream)
                SGE_HIP(hipStreamWaitEvent(ps, c->evHeavyDone, 0));  // the grouped launch (second stream)
            } else {
                SGE_HIP(hipStreamWaitEvent(ps, c->evMoveDone, 0));
            }
            L.crowd.poseIn = c->dPoseIn[poseSlot].as<PoseInput>();
        }
        if (L.waveProf) SGE_HIP(hipMemsetAsync(L.waveProf, 0, (size_t)c->crowd.count * 64, ps));
        {
            Bracket br(c, &c->evPose, ps);
            launch_pose(L, ps);
        }
        if (pipedPose) {
            SGE_HIP(hipEventRecord(c->evPosePiped[poseSlot], ps));
            c->posePending[poseSlot] = true;
            c->poseSlot = poseSlot;
        }
    }
    if (st & SGE_STAGE_SKIN) {
        if (c->mesh.vertexCount == 0) { set_error("skin stage needs a skinned mesh"); return SGE_ERR_STATE; }
        if (c->outLayoutAllocated != c->skinLayout) {
            int rc = syncAll(c);
            if (rc != SGE_OK || (rc = allocCrowdOutputs(c)) != SGE_OK) return rc;
        }
        // one RTSkinningJob per character, dstBaseVertex = running vertex offset (RTGeometryCache.swift:266-315)
        SkinLaunch L{c->mesh.positions, c->mesh.normals, c->mesh.tangents, c->mesh.boneIndices, c->mesh.boneWeights,
                     c->crowd.palettes + (size_t)first * c->boneCount * 16, c->boneCount, c->mesh.vertexCount, count,
                     (long long)first * c->mesh.vertexCount, SGE_LAYOUT_PACKED, c->skinLayout, c->dOutPos.p, c->dOutNrm.p, c->dOutTan.p};
        hipStream_t ss = c->stream;
        const bool overlap = c->overlapSkin;
        if (overlap) {
            // skin(n) on its own stream after pose(n) (and, stream order, after skin(n-1)); move(n+1) and pose(n+1) may run on
            // the main stream meanwhile
            if (pipedPose) SGE_HIP(hipStreamWaitEvent(c->skinStream, c->evPosePiped[poseSlot], 0));
            else {
                SGE_HIP(hipEventRecord(c->evPoseDone, c->stream));
                SGE_HIP(hipStreamWaitEvent(c->skinStream, c->evPoseDone, 0));
            }
            ss = c->skinStream;
        } else { // the output streams may still be written by an overlapped launch of an earlier tick
            int rcj = joinSkin(c);
            if (rcj != SGE_OK) return rcj;
        }
        const bool refit = (st & SGE_STAGE_BLAS_REFIT) != 0;
        if (refit && c->blas.entryCount == 0) { set_error("SGE_STAGE_BLAS_REFIT needs sge_blas_build"); return SGE_ERR_STATE; }
        float* boxes = refit ? c->dBlasBounds.as<float>() + (size_t)first * (c->blas.entryCount + 1) * 6 : nullptr;
        if (refit && (c->fuseBlas == 2 || (c->fuseBlas == 1 && !overlap))) {
            // one launch: the LBS kernel keeps every position it computes in an LDS tile and reduces the boxes from there
            Bracket br(c, &c->evSkin, ss);
            int rc = launch_skin_refit(L, c->blas, boxes, c->dBlasQueue.as<int>(), ss, overlap ? c->overlapFusedWorkgroups : 0);
            if (rc != SGE_OK) return rc;
        } else {
            {
                Bracket br(c, &c->evSkin, ss);
                // resident workgroups or workgroups that hand their places over (see kResidentSkinCharacters)
                int quarters = 0;
                if (overlap && c->residentSkinQuarters >= 0) quarters = c->residentSkinQuarters;
                else if (overlap && count >= kResidentSkinCharacters) quarters = (st & SGE_STAGE_AGENTS) ? kResidentSkinQuartersWithAgents : kResidentSkinQuarters;
                const int form = launch_skin(L, ss, overlap ? c->overlapSkinWorkgroups : 0, c->dSkinQueue.as<int>(), quarters, c->residentSkinCharsPerUnit);
                SGE_HIP(hipGetLastError()); // a launch that could not start would leave the previous frame's streams in place
                c->lastSkinQuarters = form > 0 ? quarters : 0; // what ran, not what was asked for (sge_debug_skin_form)
                c->lastSkinCharsPerUnit = form > 0 ? form : 1;
            }
            if (refit) { // RTAccelerationBuilder.build is enqueued right behind the skinning encoder (RayTracingScene.swift:35-43)
                Bracket br(c, &c->evBlas, ss);
                int rc = launch_blas_refit(c->blas, c->dOutPos.p, c->skinLayout, (long long)first * c->mesh.vertexCount, count, boxes, c->dBlasQueue.as<int>(), ss);
                if (rc != SGE_OK) return rc;
            }
        }
        if (overlap) { SGE_HIP(hipEventRecord(c->evSkinDone[c->palRead], c->skinStream)); c->skinPending[c->palRead] = true; c->lastSkin = c->palRead; }
    } else if (st & SGE_STAGE_BLAS_REFIT) {
        int rc = sge_blas_refit(c, first, count);
        if (rc != SGE_OK) return rc;
    }
    SGE_HIP(hipGetLastError());
    return SGE_OK;
}

// ---- skinned-geometry acceleration structure ------------------------------------------------
int sge_blas_build(sge_context* c, const uint32_t* indices, int32_t index_count) {
    if (!c || !indices || index_count <= 0) { set_error("sge_blas_build: bad argument"); return SGE_ERR_INVALID; }
    if (c->mesh.vertexCount == 0) { set_error("sge_blas_build needs sge_skinned_mesh_upload"); return SGE_ERR_STATE; }
    (void)hipSetDevice(c->device);
    { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    std::string err;
    HostBlas hb;
    if (!hb.build(c->hostMeshPos.data(), c->mesh.vertexCount, indices, index_count, err)) { set_error(err); return SGE_ERR_INVALID; }
    int rc;
    // a closest-hit query pops the newest wide node first: at most 63 siblings stay pending per level
    if (hb.levels * 63 + 1 > kBlasTraversalStackCap) { set_error("sge_blas_build: hierarchy too deep for the closest-hit traversal stack"); return SGE_ERR_CAPACITY; }
    // ticket counters of the refit kernels: [0] launches of sge_tick (possibly on the skin stream), [16] the stand-alone entry points
    if ((rc = c->dBlasQueue.alloc(256)) != SGE_OK) return rc;
    hipStream_t s = c->stream;
    if ((rc = upload(c->dBlasEntryLink, hb.entryLink.data(), hb.entryLink.size() * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dBlasWideFirst, hb.wideFirst.data(), hb.wideFirst.size() * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dBlasWideParent, hb.wideParentEntry.data(), hb.wideParentEntry.size() * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dBlasWideLevel, hb.wideLevelStart.data(), hb.wideLevelStart.size() * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dBlasSlotIdx, hb.slotIndices.data(), hb.slotIndices.size() * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dBlasSlotTri, hb.slotTriangle.data(), hb.slotTriangle.size() * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dBlasIndices, indices, (size_t)index_count * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dBlasTileStart, hb.tileRoundStart.data(), hb.tileRoundStart.size() * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dBlasRoundLen, hb.roundLen.data(), hb.roundLen.size() * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dBlasRoundCluster, hb.roundCluster.data(), hb.roundCluster.size() * 4, s)) != SGE_OK) return rc;
    if ((rc = upload(c->dBlasRoundIds, hb.roundIds.data(), hb.roundIds.size() * 4, s)) != SGE_OK) return rc;
    SGE_HIP(hipStreamSynchronize(s));
    c->hostBlas = std::move(hb);
    const HostBlas& h = c->hostBlas;
    c->blas = DevBlas{h.entryCount(), h.wideCount(), h.triCount, h.vertexCount, h.clusterCount, h.levels,
                      c->dBlasEntryLink.as<int2>(), c->dBlasWideFirst.as<int>(), c->dBlasWideParent.as<int>(), c->dBlasWideLevel.as<int>(),
                      c->dBlasSlotIdx.as<uint32_t>(), c->dBlasSlotTri.as<uint32_t>(),
                      h.tileVerts, h.tileCount, h.tileCap, c->dBlasTileStart.as<int>(), c->dBlasRoundLen.as<int>(), c->dBlasRoundCluster.as<int>(), c->dBlasRoundIds.as<uint32_t>()};
    c->blasBoundsChars = 0;
    c->blasHasUVs = false;
    return ensureBlasBuffers(c);
}

int sge_blas_set_uvs(sge_context* c, const float* uvs, int32_t vertex_count) {
    if (!c || !uvs) { set_error("sge_blas_set_uvs: bad argument"); return SGE_ERR_INVALID; }
    if (c->blas.entryCount == 0) { set_error("sge_blas_set_uvs needs sge_blas_build"); return SGE_ERR_STATE; }
    if (vertex_count != c->mesh.vertexCount) { set_error("sge_blas_set_uvs: vertex count differs from the skinned mesh"); return SGE_ERR_INVALID; }
    (void)hipSetDevice(c->device);
    int rc = upload(c->dBlasUVs, uvs, (size_t)vertex_count * 8, c->stream);
    if (rc != SGE_OK) return rc;
    SGE_HIP(hipStreamSynchronize(c->stream));
    c->blasHasUVs = true;
    return SGE_OK;
}

int sge_blas_info_get(sge_context* c, sge_blas_info* info) {
    if (!c || !info) { set_error("sge_blas_info_get: bad argument"); return SGE_ERR_INVALID; }
    if (c->blas.entryCount == 0) { set_error("no acceleration structure built"); return SGE_ERR_STATE; }
    const HostBlas& h = c->hostBlas;
    *info = sge_blas_info{h.triCount, h.clusterCount, h.entryCount(), h.wideCount(), h.levels, (int32_t)h.vertexEntries.size()};
    return SGE_OK;
}

int sge_blas_refit(sge_context* c, int32_t first, int32_t count) {
    SGE_RANGE_CHECK();
    if (c->blas.entryCount == 0) { set_error("sge_blas_refit needs sge_blas_build"); return SGE_ERR_STATE; }
    if (count == 0) return SGE_OK;
    (void)hipSetDevice(c->device);
    { int rcj = joinSkin(c); if (rcj != SGE_OK) return rcj; }
    Bracket br(c, &c->evBlas);
    int rc = launch_blas_refit(c->blas, c->dOutPos.p, c->outLayoutAllocated, (long long)first * c->mesh.vertexCount, count,
                               c->dBlasBounds.as<float>() + (size_t)first * (c->blas.entryCount + 1) * 6, c->dBlasQueue.as<int>() + 16, c->stream);
    if (rc != SGE_OK) return rc;
    SGE_HIP(hipGetLastError());
    return SGE_OK;
}

int sge_blas_refit_buffers(sge_context* c, const void* d_positions, int32_t layout, int64_t first_vertex, int32_t count, void* d_bounds) {
    if (!c || !d_positions || !d_bounds || count < 0 || first_vertex < 0 || (layout != SGE_LAYOUT_PACKED && layout != SGE_LAYOUT_PADDED16)) {
        set_error("sge_blas_refit_buffers: bad argument");
        return SGE_ERR_INVALID;
    }
    if (c->blas.entryCount == 0) { set_error("sge_blas_refit_buffers needs sge_blas_build"); return SGE_ERR_STATE; }
    (void)hipSetDevice(c->device);
    { int rcj = joinSkin(c); if (rcj != SGE_OK) return rcj; }
    Bracket br(c, &c->evBlas);
    int rc = launch_blas_refit(c->blas, d_positions, layout, (long long)first_vertex, count, reinterpret_cast<float*>(d_bounds), c->dBlasQueue.as<int>() + 16, c->stream);
    if (rc != SGE_OK) return rc;
    SGE_HIP(hipGetLastError());
    return SGE_OK;
}

int sge_blas_bounds_download(sge_context* c, int32_t first, int32_t count, float* bounds) {
    SGE_RANGE_CHECK();
    if (!bounds) { set_error("sge_blas_bounds_download: bad argument"); return SGE_ERR_INVALID; }
    if (c->blas.entryCount == 0) { set_error("no acceleration structure built"); return SGE_ERR_STATE; }
    (void)hipSetDevice(c->device);
    { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    const size_t row = (size_t)(c->blas.entryCount + 1) * 24;
    SGE_HIP(hipMemcpy(bounds, reinterpret_cast<const char*>(c->dBlasBounds.p) + (size_t)first * row, (size_t)count * row, hipMemcpyDeviceToHost));
    return SGE_OK;
}

int sge_blas_buffers(sge_context* c, void** d_bounds, void** d_indices) {
    if (!c || c->blas.entryCount == 0) { set_error("no acceleration structure built"); return SGE_ERR_STATE; }
    if (d_bounds) *d_bounds = c->dBlasBounds.p;
    if (d_indices) *d_indices = c->dBlasIndices.p;
    return SGE_OK;
}

int sge_blas_instances_upload(sge_context* c, int32_t first, int32_t count, const float* model_matrices) {
    SGE_RANGE_CHECK();
    if (!model_matrices) { set_error("sge_blas_instances_upload: bad argument"); return SGE_ERR_INVALID; }
    if (c->blas.entryCount == 0) { set_error("sge_blas_instances_upload needs sge_blas_build"); return SGE_ERR_STATE; }
    (void)hipSetDevice(c->device);
    SGE_HIP(hipMemcpyAsync(c->dBlasInstances.as<float>() + (size_t)first * 16, model_matrices, (size_t)count * 64, hipMemcpyHostToDevice, c->stream));
    SGE_HIP(hipStreamSynchronize(c->stream));
    return SGE_OK;
}

int sge_blas_intersect_batch(sge_context* c, const sge_blas_ray* rays, int32_t count, sge_blas_hit* hits) {
    if (!c || count < 0 || (count > 0 && (!rays || !hits))) { set_error("sge_blas_intersect_batch: bad argument"); return SGE_ERR_INVALID; }
    if (c->blas.entryCount == 0 || c->crowd.count == 0) { set_error("sge_blas_intersect_batch needs sge_blas_build and characters"); return SGE_ERR_STATE; }
    if (count == 0) return SGE_OK;
    (void)hipSetDevice(c->device);
    { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    int rc;
    if ((rc = upload(c->dBlasRays, rays, (size_t)count * sizeof(sge_blas_ray), c->stream)) != SGE_OK) return rc;
    if ((rc = c->dBlasHits.alloc((size_t)count * sizeof(sge_blas_hit))) != SGE_OK) return rc;
    bool anyInstance = false;
    for (int i = 0; i < count && !anyInstance; ++i) anyInstance = rays[i].instance < 0;
    BlasTrace T{c->blas, c->dOutPos.p, c->dOutNrm.p, c->dOutTan.as<float>(), c->dBlasIndices.as<uint32_t>(), c->outLayoutAllocated,
                c->dBlasBounds.as<float>(), c->dBlasInstances.as<float>(), c->crowd.count, nullptr,
                c->blasHasUVs ? c->dBlasUVs.as<float>() : nullptr};
    if (anyInstance && (rc = buildInstanceLevel(c, T)) != SGE_OK) return rc;
    launch_blas_intersect(T, c->dBlasRays.as<sge_blas_ray>(), count, c->dBlasHits.as<sge_blas_hit>(), anyInstance, c->stream);
    SGE_HIP(hipGetLastError());
    SGE_HIP(hipMemcpyAsync(hits, c->dBlasHits.p, (size_t)count * sizeof(sge_blas_hit), hipMemcpyDeviceToHost, c->stream));
    SGE_HIP(hipStreamSynchronize(c->stream));
    return SGE_OK;
}

int sge_blas_intersect_device(sge_context* c, const void* d_rays, int32_t count, void* d_hits, int32_t any_instance) {
    if (!c || count < 0 || (count > 0 && (!d_rays || !d_hits))) { set_error("sge_blas_intersect_device: bad argument"); return SGE_ERR_INVALID; }
    if (c->blas.entryCount == 0 || c->crowd.count == 0) { set_error("sge_blas_intersect_device needs sge_blas_build and characters"); return SGE_ERR_STATE; }
    if (count == 0) return SGE_OK;
    (void)hipSetDevice(c->device);
    { int rcj = joinSkin(c); if (rcj != SGE_OK) return rcj; }
    int rc;
    BlasTrace T{c->blas, c->dOutPos.p, c->dOutNrm.p, c->dOutTan.as<float>(), c->dBlasIndices.as<uint32_t>(), c->outLayoutAllocated,
                c->dBlasBounds.as<float>(), c->dBlasInstances.as<float>(), c->crowd.count, nullptr,
                c->blasHasUVs ? c->dBlasUVs.as<float>() : nullptr};
    if (any_instance && (rc = buildInstanceLevel(c, T)) != SGE_OK) return rc;
    launch_blas_intersect(T, reinterpret_cast<const sge_blas_ray*>(d_rays), count, reinterpret_cast<sge_blas_hit*>(d_hits), any_instance != 0, c->stream);
    SGE_HIP(hipGetLastError());
    return SGE_OK;
}

int sge_blas_profile_read(sge_context* c, double* ms, int64_t* launches, int reset) {
    if (!c) return SGE_ERR_INVALID;
    (void)hipSetDevice(c->device);
    int rc = drainEvents(c->evBlas);
    if (rc != SGE_OK) return rc;
    if (ms) *ms = c->evBlas.ms;
    if (launches) *launches = c->evBlas.launches;
    if (reset) { c->evBlas.ms = 0; c->evBlas.launches = 0; }
    return SGE_OK;
}

// ---- agents ---------------------------------------------------------------------------------
int sge_agents_export(sge_context* c, void* d_out) {
    if (!c || !d_out) { set_error("sge_agents_export: bad argument"); return SGE_ERR_INVALID; }
    (void)hipSetDevice(c->device);
    launch_agents_export(c->crowd, reinterpret_cast<sge_agent_state*>(d_out), c->stream);
    SGE_HIP(hipGetLastError());
    return SGE_OK;
}

int sge_agents_import(sge_context* c, const void* d_all, int32_t total, int32_t self_offset) {
    if (!c || total < 0 || self_offset < 0 || (total > 0 && !d_all)) { set_error("sge_agents_import: bad argument"); return SGE_ERR_INVALID; }
    c->agents = DevAgents{};
    c->agents.all = total > 0 ? reinterpret_cast<const sge_agent_state*>(d_all) : nullptr;
    c->agents.total = total;
    c->agents.selfOffset = self_offset;
    return SGE_OK;
}

// RCCL is an optional run-time dependency of exactly one entry point: resolved with dlopen so that single-GPU users of the library
// need no librccl.so.
namespace {
typedef int (*NcclAllGatherFn)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/, hipStream_t);
NcclAllGatherFn resolveAllGather() {
    static NcclAllGatherFn fn = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        // The communicator the host hands in belongs to the RCCL copy that created it: use the copy already in the process (a torch
        // wheel bundles its own librccl.so.1, not on the loader path) before loading one by name — two copies in one process and a
        // foreign ncclComm_t passed across them is undefined behaviour.
        fn = reinterpret_cast<NcclAllGatherFn>(dlsym(RTLD_DEFAULT, "ncclAllGather"));
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            if (fn) break;
            if (void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD)) fn = reinterpret_cast<NcclAllGatherFn>(dlsym(h, "ncclAllGather"));
        }
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) { // none resident: load by name
            if (fn) break;
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) fn = reinterpret_cast<NcclAllGatherFn>(dlsym(h, "ncclAllGather"));
        }
    }
    return fn;
}
} // namespace

int sge_agents_allgather(sge_context* c, void* nccl_comm, int32_t rank, int32_t world_size, int32_t slot) {
    if (!c || world_size < 1 || rank < 0 || rank >= world_size || slot < c->crowd.count || (world_size > 1 && !nccl_comm)) {
        set_error("sge_agents_allgather: bad argument (slot must be >= this context's character count)");
        return SGE_ERR_INVALID;
    }
    (void)hipSetDevice(c->device);
    const size_t slotBytes = (size_t)slot * sizeof(sge_agent_state);
    int rc;
    if ((rc = c->dAgentsAll.alloc(slotBytes * world_size)) != SGE_OK) return rc;
    sge_agent_state* all = c->dAgentsAll.as<sge_agent_state>();
    sge_agent_state* mine = all + (size_t)rank * slot; // in place: this rank's records sit where the gather puts them
    // padding entries: radius < 0 (every float of the record = -1 does it)
    if (slot > c->crowd.count) launch_agents_pad(mine + c->crowd.count, slot - c->crowd.count, c->stream);
    launch_agents_export(c->crowd, mine, c->stream);
    if (world_size > 1) {
        NcclAllGatherFn allGather = resolveAllGather();
        if (!allGather) { set_error("sge_agents_allgather: librccl.so not found (ncclAllGather)"); return SGE_ERR_STATE; }
        const int ncclChar = 0; // ncclInt8 / ncclChar
        const int nr = allGather(mine, all, slotBytes, ncclChar, nccl_comm, c->stream);
        if (nr != 0) { set_error("ncclAllGather failed with code " + std::to_string(nr)); return SGE_ERR_DEVICE; }
    }
    SGE_HIP(hipGetLastError());
    return sge_agents_import(c, all, slot * world_size, rank * slot);
}

// ---- diagnostics ------------------------------------------------------------------------------
int sge_profile_read(sge_context* c, sge_stage_times* out, int reset) {
    if (!c || !out) return SGE_ERR_INVALID;
    (void)hipSetDevice(c->device);
    int rc;
    if ((rc = drainEvents(c->evMove)) != SGE_OK || (rc = drainEvents(c->evPose)) != SGE_OK ||
        (rc = drainEvents(c->evSkin)) != SGE_OK || (rc = drainEvents(c->evAgents)) != SGE_OK) return rc;
    *out = sge_stage_times{c->evMove.ms, c->evPose.ms, c->evSkin.ms, c->evAgents.ms,
                           c->evMove.launches, c->evPose.launches, c->evSkin.launches, c->evAgents.launches};
    if (reset) {
        for (Events* e : {&c->evMove, &c->evPose, &c->evSkin, &c->evAgents}) { e->ms = 0; e->launches = 0; }
        c->skinLaunchMs.clear();
    }
    return SGE_OK;
}

// diagnostics: the HIP-event duration of every skin launch since the last reset of sge_profile_read, oldest first (what the
// `roofline.ms_per_launch` of bench.py averages); *count = how many there are, min(*count, cap) are written
int sge_debug_skin_launch_times(sge_context* c, float* out_ms, int32_t cap, int32_t* count) {
    if (!c || !count || cap < 0 || (cap > 0 && !out_ms)) return SGE_ERR_INVALID;
    (void)hipSetDevice(c->device);
    int rc = drainEvents(c->evSkin);
    if (rc != SGE_OK) return rc;
    *count = (int32_t)c->skinLaunchMs.size();
    for (int i = 0; i < cap && i < *count; ++i) out_ms[i] = c->skinLaunchMs[(size_t)i];
    return SGE_OK;
}

// diagnostics: what allocCrowdOutputs found when it placed the skinned output streams: the kept placement's time for one
// three-stream store pass (0: not probed) and how many placements were timed
int sge_debug_placement(sge_context* c, float* kept_ms, int32_t* tried) {
    if (!c) return SGE_ERR_INVALID;
    if (kept_ms) *kept_ms = c->placementMs;
    if (tried) *tried = c->placementTried;
    return SGE_OK;
}

int sge_separation_params(sge_context* c, int32_t iterations, float separation_margin, float height_margin) {
    if (!c) return SGE_ERR_INVALID;
    c->separationIterations = iterations < 1 ? 1 : iterations; // max(1, iterations) :2146
    c->separationMargin = separation_margin;
    c->separationHeightMargin = height_margin;
    return SGE_OK;
}

// which form the newest skin stage of sge_tick took: quarters of a resident workgroup per CU (0: workgroups that come and go) and
// characters per work unit
int sge_debug_skin_form(sge_context* c, int32_t* quarters, int32_t* chars_per_unit) {
    if (!c) return SGE_ERR_INVALID;
    if (quarters) *quarters = c->lastSkinQuarters;
    if (chars_per_unit) *chars_per_unit = c->lastSkinCharsPerUnit;
    return SGE_OK;
}

// diagnostics of the crowd path of the separation stage, last pass: out[0] listed agents, out[1] loops drawn, out[2] redo flags
// (bit 0: an agent had more than 64 candidates, bit 1: an agent was pushed further than a cell; either: the pass ran serially)
int sge_debug_separation(sge_context* c, int32_t* out) {
    if (!c || !out) return SGE_ERR_INVALID;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!c->dSepFlow.p || c->crowd.count <= 0) return SGE_OK;
    (void)hipSetDevice(c->device);
    { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    SGE_HIP(hipMemcpy(out, reinterpret_cast<const char*>(c->dSepFlow.p) + separationFlowControlOffset(c->crowd.count), 16, hipMemcpyDeviceToHost));
    return SGE_OK;
}

int sge_move_cost_read(sge_context* c, int32_t first, int32_t count, int32_t* evaluations) {
    SGE_RANGE_CHECK();
    if (!evaluations) return SGE_ERR_INVALID;
    (void)hipSetDevice(c->device);
    { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    if (count > 0) SGE_HIP(hipMemcpy(evaluations, c->dCost.as<int>() + first, (size_t)count * sizeof(int), hipMemcpyDeviceToHost));
    return SGE_OK;
}

// diagnostics: per-wavefront cycle counters of the last grouped move launch (SGE_WAVE_PROF=1), 8 x u64 per wavefront
int sge_debug_wave_profile(sge_context* c, uint64_t* out, int32_t waves) {
    if (!c || !out || !c->dWaveProf.p || (size_t)waves * 64 > c->dWaveProf.bytes) return SGE_ERR_INVALID;
    (void)hipSetDevice(c->device);
    { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    SGE_HIP(hipMemcpy(out, c->dWaveProf.p, (size_t)waves * 64, hipMemcpyDeviceToHost));
    return SGE_OK;
}

// diagnostics: the order / heavy lists of the last move launch: lists[2 * count], counts[2] (listed, heavy)
int sge_debug_move_lists(sge_context* c, int32_t* lists, int32_t* counts) {
    if (!c || !lists || !counts) return SGE_ERR_INVALID;
    (void)hipSetDevice(c->device);
    { int rcs = syncAll(c); if (rcs != SGE_OK) return rcs; }
    SGE_HIP(hipMemcpy(lists, c->dLists.p, (size_t)c->crowd.count * 8, hipMemcpyDeviceToHost));
    SGE_HIP(hipMemcpy(counts, c->dListCounts.p, 8, hipMemcpyDeviceToHost));
    return SGE_OK;
}

int sge_move_stats_read(sge_context* c, sge_move_stats* out, int reset) {
    if (!c || !out) return SGE_ERR_INVALID;
    (void)hipSetDevice(c->device);
    std::vector<unsigned long long> shards((size_t)kStatShards * 8);
    SGE_HIP(hipMemcpyAsync(shards.data(), c->dStats.p, shards.size() * 8, hipMemcpyDeviceToHost, c->stream));
    SGE_HIP(hipStreamSynchronize(c->stream));
    unsigned long long h[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int sdx = 0; sdx < kStatShards; ++sdx)
        for (int k = 0; k < 7; ++k) h[k] += shards[(size_t)sdx * 8 + k];
    out->queries = h[0]; out->candidates = h[1]; out->sweepIterations = h[2]; out->overflow = h[3];
    out->traversalSteps = h[4]; out->sweepTrips = h[5]; out->prunedPairs = h[6];
    if (reset) { SGE_HIP(hipMemsetAsync(c->dStats.p, 0, (size_t)kStatShards * 64, c->stream)); SGE_HIP(hipStreamSynchronize(c->stream)); }
    return SGE_OK;
}

} // extern "C"
