// Capsule-CCD move-and-slide against the static and dynamic triangle sets, for gfx950.
// Replaces, per fixed step and per character:
//   PhysicsIntentSystem (controller branch)   Game/Systems.swift:205-250
//   GravitySystem                             Game/Systems.swift:596-620
//   KinematicMoveStopSystem.fixedUpdate       Game/Systems.swift:1823-1902
//     PlatformCarry :644-732, DepenetrationResolver :734-808, resolveKinematicSweep :1658-1765,
//     SlideResolver.resolveHit :1229-1375, GroundProbe/GroundSnap/SlopeFriction :826-1021,
//     DefaultContactCachePolicy/ContactManifoldCache :1102-1205, AgentSweepSolver :1053-1091
//   CollisionQuery.raycast / capsuleCast* / capsuleOverlap / capsuleOverlapAll   Game/CollisionQuery.swift:85-159, 768-1283
//   sweepCapsuleTriangle / refineTOI / segmentTriangleDistance  Game/CollisionQuery.swift:1285-1573 (sge_ccd_prims.hpp)
//
// Mapping: ONE WAVEFRONT PER CHARACTER (64-thread workgroups; the few characters whose previous step was very
// expensive get a 512-thread workgroup, see "heavy characters"). The per-character state machine (platform carry ->
// depenetration -> slide iterations -> ground probe -> snap -> friction -> write-back) is wave-uniform and lives in
// LDS; each BVH query inside it is wave-cooperative:
//   * traversal: a wide BVH (64 entries per node, cut out of the reference's binary tree) — one step pops a node and
//     tests its 64 entries, one per lane, or scans a range of <= 64 triangles; survivors are compacted into LDS lists
//     with ballot + prefix-popcount. Static and dynamic sets share the arrays: a traversal starts from both roots;
//   * candidates: the reference's result depends on its right-child-first visit order only through strict `<`
//     tie-breaks and capsuleOverlapAll's "first maxHits"; every triangle carries its visit rank (dynamic set offset
//     by the static count), so lanes may run in any order and the wave reduces on (toi, rank) / selects by rank;
//   * sweep: conservative advancement + 10-step bisection + final contact evaluation run as a per-lane state machine
//     in which every loop trip performs exactly one segmentTriangleDistance evaluation, so lanes in different phases
//     stay convergent on the expensive code; lanes stream — each pulls the next (ray, triangle) work item from an LDS
//     queue as soon as its own is finished. A lane stops early once its last safe t exceeds the best accepted TOI of its
//     ray (its hit could no longer win the strict `<`), which is what makes the 200-unit fall probe cheap.
// Compiled with -ffp-contract=off; float32 arithmetic is IEEE and in the oracle's order, so TOIs, normals and the
// discrete contact state match bit for bit.
#include "sge_internal.hpp"
#include "sge_ccd_prims.hpp"

namespace sge {

constexpr int kWave = 64;
constexpr int kStackCap = kTraversalStackCap; // wide nodes pending (each pop adds <= 64; sge_api checks the tree depth against it)
constexpr int kCandCap = 128;  // every traversal loop stops expanding at 64 pending candidates: 63 + 64 is the most there can be
constexpr int kRangeCap = 128;
constexpr int kItemCap = 512;   // the queue is topped up / swept before it exceeds 64 + 64 * kMaxRays
constexpr int kMaxRays = 6;
#ifndef SGE_GROUP
#define SGE_GROUP 4
#endif
// Experiment: one collision wavefront per SIMD that cannot take the place of an LBS workgroup (see launch_move)
#ifndef SGE_CCD_EXCLUSIVE
#define SGE_CCD_EXCLUSIVE 0
#endif
#ifdef SGE_CCD_SETPRIO // experiment: issue priority of the collision wavefronts over the LBS wavefronts they share SIMDs with
#define SGE_PAD_VGPRS() __builtin_amdgcn_s_setprio(SGE_CCD_SETPRIO)
#elif SGE_CCD_EXCLUSIVE == 2
#define SGE_PAD_VGPRS() do { asm volatile("" ::: "v175"); __builtin_amdgcn_s_setprio(3); } while (0)
#elif SGE_CCD_EXCLUSIVE
#define SGE_PAD_VGPRS() asm volatile("" ::: "v175")
#else
#define SGE_PAD_VGPRS()
#endif
#ifdef SGE_CCD_SETPRIO
#define SGE_HEAVY_PRIO() __builtin_amdgcn_s_setprio(SGE_CCD_SETPRIO)
#else
#define SGE_HEAVY_PRIO()
#endif
#ifndef SGE_GROUP_WAVES
#define SGE_GROUP_WAVES 3
#endif
constexpr int kGroup = SGE_GROUP;             // characters per wavefront in move_group_kernel
constexpr int kRaySlots = kMaxRays * kGroup;  // ray slots [g * kMaxRays, (g + 1) * kMaxRays) belong to character g of the wavefront
constexpr int kItemRayShift = 27;             // work item = (ray slot << 27) | triangle slot (sge_api checks triCount < 2^27)
constexpr int kItemSlotMask = (1 << kItemRayShift) - 1;
constexpr int kRefillIdle = 16;  // idle lanes that trigger a queue top-up while others still march

struct OverlapRec { float depth; F3 position, normal, triNormal; int triIndex, rank; };
struct CastRec { float toi; F3 position, normal, triNormal; int triIndex; };

struct WaveShared {
    int stack[kStackCap];
    int cand[kCandCap];
    int ranges[kRangeCap];   // pending (firstSlot << 7 | count) triangle ranges of the wide BVH
    int items[kItemCap];     // (ray << 28) | slot work items of a multi-ray cast
    // up to kMaxRays casts that share radius/halfHeight/filters run as ONE traversal + shared sweep batches
    int rayCount;
    F3 rayFrom[kRaySlots], rayDelta[kRaySlots], rayDir[kRaySlots], rayMin[kRaySlots], rayMax[kRaySlots];
    float rayLen[kRaySlots];
    int rayMaxIter[kRaySlots], rayValid[kRaySlots];
    int rayVertical[kRaySlots];  // delta = (0, dy, 0): the ground probe's casts (verticalSweepMisses)
    // capsule and acceptance filters of the cast a ray belongs to (move_group_kernel sweeps rays of several characters together)
    float rayRadius[kRaySlots], rayHalfHeight[kRaySlots], rayMinNormalY[kRaySlots];
    int rayFilter[kRaySlots];    // bit 0: blockingOnly, bit 1: minNormalY set
    unsigned long long rayKey[kRaySlots]; // (toi bits << 32) | visit rank of the best accepted hit so far
    CastRec rayRec[kRaySlots];
};

struct WaveStats { unsigned int queries, candidates, evals, overflow, steps, trips, pruned; };
// The counters are sharded over kStatShards cache lines (8 x u64 each): thousands of waves adding to ONE line
// serialise at ~11 ns per atomic on this chip, which would cost more than the collision work itself.
__device__ __forceinline__ unsigned long long* statShard(unsigned long long* stats) {
    return stats + (size_t)(blockIdx.x % kStatShards) * 8;
}
#ifdef SGE_CCD_TIMING
// diagnostic build only: shader-clock cycles per wave spent in traversal / sweep / everything
__device__ unsigned long long g_cycTraverse, g_cycSweep, g_cycTotal;
#define SGE_T0() long long _t0 = (long long)__builtin_amdgcn_s_memtime()
#define SGE_T1(acc) acc += (long long)__builtin_amdgcn_s_memtime() - _t0
#else
#define SGE_T0()
#define SGE_T1(acc)
#endif

// One instance per 64-thread workgroup (= per character / per query).
__shared__ WaveShared sh;
// results of capsuleOverlapAll (their own variables, so that kernels without an overlap query do not pay for them in LDS)
__shared__ OverlapRec sOvl[SGE_MAX_OVERLAP_HITS], sOvlTmp[SGE_MAX_OVERLAP_HITS];
// The wave-uniform per-character state lives in LDS, not in registers: it is touched only
// between queries, and every lane reads/writes the same value in lockstep.
// (indexed by the character's place g in its wavefront: always 0 in the one-character kernels)
__shared__ sge_body_state sBodyA[kGroup];
__shared__ sge_controller_params sParamsA[kGroup];
__shared__ sge_controller_state sCtrlA[kGroup];

__device__ __forceinline__ int laneId() { return threadIdx.x & (kWave - 1); }
// Write-back of LDS slot g to character e: the body words the move stage owns, the controller, and (crowd.poseIn) the animation
// stages' copy of what they read of the two (PoseInput: lanes 32..47 gather its 16 dwords from the same LDS slots).
__device__ __forceinline__ void storeCharacter(const DevCrowd& crowd, const int e, const int g, const int lane) {
    uint32_t* gb = reinterpret_cast<uint32_t*>(crowd.bodies + e);
    uint32_t* gc = reinterpret_cast<uint32_t*>(crowd.controllers + e);
    const uint32_t* sb = reinterpret_cast<const uint32_t*>(&sBodyA[g]);
    const uint32_t* sc = reinterpret_cast<const uint32_t*>(&sCtrlA[g]);
    if (lane < kBodyMoveWords) gb[lane] = sb[lane];
    if (lane < 32) gc[lane] = sc[lane];
    if (crowd.poseIn && lane >= 32 && lane < 48) {
        const int j = lane - 32;
        const uint32_t* src = j < 10 ? sb + 6 + j : (j < 13 ? sc + (j - 10) : sc + kCtrlFlagsWord + (j - 13)); // j = 15: the controller's first pad word
        reinterpret_cast<uint32_t*>(crowd.poseIn + e)[j] = *src;
    }
}
__device__ __forceinline__ int prefixCount(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}
__device__ __forceinline__ unsigned long long waveMinU64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned lo = __shfl_xor((unsigned)v, o, kWave), hi = __shfl_xor((unsigned)(v >> 32), o, kWave);
        unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w < v ? w : v;
    }
    return v;
}

// ---------------------------------------------------------------------------
// BVH traversal step: pop <= 64 nodes, push children, append leaf triangles
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool boxDisjoint(F3 bmin, F3 bmax, F3 minP, F3 maxP) {
    return bmax.x < minP.x || bmin.x > maxP.x || bmax.y < minP.y || bmin.y > maxP.y || bmax.z < minP.z || bmin.z > maxP.z;
}

// One traversal step over the wide BVH (DevCollision::wide): either scan one pending triangle range
// (<= 64 triangles, one per lane, coalesced 48-B records) into the candidate list, or pop one wide node and
// test its 64 entries (one per lane, one coalesced 2-KB load). Ranges are consumed before the next pop, so
// at most 64 of them are ever pending.
__device__ __forceinline__ void expandNodes(const DevCollision& col, F3 minP, F3 maxP, uint32_t mask,
                                            int& stackSize, int& rangeCount, int& candCount, WaveStats& st) {
    const int lane = laneId();
    st.steps += 1;
    if (rangeCount > 0) {
        rangeCount -= 1;
        const int packed = sh.ranges[rangeCount]; // (firstSlot << 7) | count, count in 1..64
        const int first = packed >> 7, cnt = packed & 127;
        bool c = lane < cnt;
        const int slot = first + lane;
        if (c) {
            const float4* tp = reinterpret_cast<const float4*>(col.tris + slot);
            float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
            F3 v0{t0.x, t0.y, t0.z}, v1{t0.w, t1.x, t1.y}, v2{t1.z, t1.w, t2.x};
            uint32_t layer = __float_as_uint(t2.y);
            c = (layer & mask) != 0 && !boxDisjoint(vmin(v0, vmin(v1, v2)), vmax(v0, vmax(v1, v2)), minP, maxP);
        }
        unsigned long long mc = __ballot(c);
        int totc = __popcll(mc);
        if (candCount + totc > kCandCap) { st.overflow += 1; }
        else {
            if (c) sh.cand[candCount + prefixCount(mc)] = slot;
            candCount += totc;
        }
        __syncthreads();
        return;
    }
    stackSize -= 1;
    const int w = sh.stack[stackSize];
    __syncthreads();
    const float4* np = reinterpret_cast<const float4*>(col.wide + (size_t)w * kWideWidth + lane);
    float4 n0 = np[0], n1 = np[1];
    const bool overlap = !boxDisjoint(F3{n0.x, n0.y, n0.z}, F3{n0.w, n1.x, n1.y}, minP, maxP);
    const int a = __float_as_int(n1.z), b = __float_as_int(n1.w);
    const bool isRange = b > 0;
    const unsigned long long mi = __ballot(overlap && !isRange), mr = __ballot(overlap && isRange);
    const int ti = __popcll(mi), tr = __popcll(mr);
    if (stackSize + ti > kStackCap || rangeCount + tr > kRangeCap) { st.overflow += 1; }
    else {
        if (overlap && !isRange) sh.stack[stackSize + prefixCount(mi)] = a;
        if (overlap && isRange) sh.ranges[rangeCount + prefixCount(mr)] = ((~a) << 7) | b;
        stackSize += ti;
        rangeCount += tr;
    }
    __syncthreads();
}

// Start a traversal over both triangle sets: the static root, and the dynamic set's root when both exist.
__device__ __forceinline__ int initTraversal(const DevCollision& col) {
    if (laneId() == 0) { sh.stack[0] = 0; sh.stack[1] = col.dynWide; }
    return col.dynWide >= 0 ? 2 : 1;
}

// Conservative reject for a VERTICAL sweep (delta = (0, dy, 0), the ground probe's snap / fall / offset casts, Systems.swift:844-921).
// center = from + dir * t keeps x and z exactly, so at every t of the march the capsule axis lies on the vertical line through
// (px, pz) and its distance to the triangle is at least the XZ-plane distance from (px, pz) to the triangle's projection. When that
// 2-D distance exceeds radius + contactEps by a margin that covers the rounding of both computations, no evaluation of
// sweepCapsuleTriangle (CollisionQuery.swift:1303-1322) can report contact: the reference marches such a triangle to `t > maxDistance`
// or to its iteration cap and returns nil. Skipping it changes no result — and it is what a capsule falling beside a wall is made
// of: hundreds of near-vertical triangles a skin width away that each crawl through up to 256 advancement steps.
// No division: an edge's distance test is cross^2 > R^2 |e|^2. A point inside the projection (or a projection that degenerates to a
// line through the point) is never rejected.
__device__ __forceinline__ bool edgeFarXZ(float px, float pz, float ax, float az, float bx, float bz, float R2, float& crossOut) {
    const float ex = bx - ax, ez = bz - az, wx = px - ax, wz = pz - az;
    const float c1 = wx * ex + wz * ez, c2 = ex * ex + ez * ez;
    const float cr = ex * wz - ez * wx;
    crossOut = cr;
    const float ux = px - bx, uz = pz - bz;
    const float dv = c1 <= 0 ? wx * wx + wz * wz : ux * ux + uz * uz;   // nearest end point
    return (c1 <= 0 || c1 >= c2) ? dv > R2 : cr * cr > R2 * c2;
}
// A ground cast (capsuleCastGround: minNormalY set) drops every hit whose triangleNormal.y is below minNormalY WITHOUT lowering
// bestT (CollisionQuery.swift:1095-1097), and triangleNormal is +-normalize(cross(v1 - v0, v2 - v0)) (:1335-1339, the sign follows
// the contact normal). A triangle with |normal.y| < minNormalY is therefore rejected whichever way it is hit: its whole march —
// for a wall below a capsule that walked off a ledge, a real contact found after ~17 evaluations, times the thousands of wall
// triangles along a 200-unit fall probe — has no effect on the result. Same expression, same bits as the march's own triNormal.
__device__ __forceinline__ bool tooSteepForGroundCast(F3 v0, F3 v1, F3 v2, float minNormalY) {
    return fabsf(normalize(cross(v1 - v0, v2 - v0)).y) < minNormalY;
}
__device__ __forceinline__ bool verticalSweepMisses(float px, float pz, float radius, F3 v0, F3 v1, F3 v2) {
    const float R = radius + 1e-5f + (0.01f + 4e-5f * (fabsf(px) + fabsf(pz)));
    const float R2 = R * R;
    float k0, k1, k2;
    const bool f0 = edgeFarXZ(px, pz, v0.x, v0.z, v1.x, v1.z, R2, k0);
    const bool f1 = edgeFarXZ(px, pz, v1.x, v1.z, v2.x, v2.z, R2, k1);
    const bool f2 = edgeFarXZ(px, pz, v2.x, v2.z, v0.x, v0.z, R2, k2);
    const bool inside = (k0 >= 0 && k1 >= 0 && k2 >= 0) || (k0 <= 0 && k1 <= 0 && k2 <= 0);
    return f0 && f1 && f2 && !inside;
}

struct Tri { F3 v0, v1, v2; int triIndex, rank; };
__device__ __forceinline__ Tri loadTri(const DevCollision& col, int slot) {
    const float4* tp = reinterpret_cast<const float4*>(col.tris + slot);
    float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
    Tri t;
    t.v0 = F3{t0.x, t0.y, t0.z}; t.v1 = F3{t0.w, t1.x, t1.y}; t.v2 = F3{t1.z, t1.w, t2.x};
    t.triIndex = __float_as_int(t2.z);
    t.rank = __float_as_int(t2.w);
    return t;
}

// ---------------------------------------------------------------------------
// capsuleCastCombined over the static set (CollisionQuery.swift:980-1117)
// ---------------------------------------------------------------------------
enum { PH_MARCH = 0, PH_REFINE = 1, PH_FINAL = 2, PH_DONE = 3 };

// one capsuleCastCombined (CollisionQuery.swift:980-1009)
__device__ __forceinline__ bool rayHit(int r) { return (unsigned)(sh.rayKey[r] & 0xffffffffull) != 0xffffffffu; }

// ---- heavy characters: eight wavefronts sweep one character's work items --------------------------------------------
// A capsule falling beside a wall meets hundreds of triangles that each crawl through up to 256 dependent advancement
// steps; one wavefront then needs a thousand trips for a single query while the rest of the GPU idles. Characters whose
// previous step was that expensive run in a 512-thread workgroup: wave 0 executes the step exactly as the one-wave kernel
// does, and whenever a cast has gathered its work items the other seven waves join the sweep (items are handed out through an
// LDS cursor; the result is still the minimum (toi, visit rank) key, so the answer is bit-identical). The helper waves
// otherwise sit in a loop matching wave 0's barriers.
enum { HCMD_NONE = 0, HCMD_MARCH = 1, HCMD_EXIT = 2 };
#ifndef SGE_HEAVY_WAVES
#define SGE_HEAVY_WAVES 8
#endif
constexpr int kHeavyWaves = SGE_HEAVY_WAVES;
constexpr int kHeavyItemCap = 2048;
struct HeavyShared {
    int items[kHeavyItemCap];
    int cursor, count, cmd;
    unsigned evalSum; // distance evaluations of the current step over all the workgroup's waves (the cost the classifier reads)
    float radius, halfHeight, minNormalY;
    int blockingOnly, hasMinNormalY;
};
__shared__ HeavyShared hv;

// The sweep of hv.items[0..hv.count) by every wave of the workgroup, between two workgroup barriers. No barrier inside:
// the waves run different numbers of trips. Accepted hits only lower sh.rayKey; the records are rebuilt afterwards.
__device__ __forceinline__ void heavyMarch(const DevCollision& col, WaveStats& st) {
    const int lane = laneId();
    const float radius = hv.radius, halfHeight = hv.halfHeight, minNormalY = hv.minNormalY;
    const bool blockingOnly = hv.blockingOnly != 0, hasMinNormalY = hv.hasMinNormalY != 0;
    const int total = hv.count;
    const float minAdvance = smax(radius * 0.02f, 1e-4f);
    const float contactEps = 1e-5f;
    int phase = PH_DONE, myRay = 0, iter = 0, refineK = 0, maxIter = 0;
    bool more = true;
    Tri tri;
    tri.v0 = tri.v1 = tri.v2 = F3{0, 0, 0}; tri.triIndex = -1; tri.rank = 0x7fffffff;
    F3 triNormal{0, 0, 0};
    float t = 0, lastSafeT = 0, lo = 0, hi = 0, tEval = 0, len = 0;
    int crawlRun = 0; // consecutive march steps that advanced by exactly minAdvance
    unsigned evals = 0;
    while (true) {
        const unsigned long long want = __ballot(phase == PH_DONE && more);
        if (want) {
            const int first = __ffsll((long long)want) - 1;
            int base = 0;
            if (lane == first) base = atomicAdd(&hv.cursor, __popcll(want));
            base = __shfl(base, first, kWave);
            if (phase == PH_DONE && more) {
                const int idx = base + prefixCount(want);
                if (idx < total) {
                    const int it = hv.items[idx];
                    myRay = (unsigned)it >> kItemRayShift;
                    tri = loadTri(col, it & kItemSlotMask);
                    triNormal = normalize(cross(tri.v1 - tri.v0, tri.v2 - tri.v0));
                    len = sh.rayLen[myRay];
                    maxIter = sh.rayMaxIter[myRay];
                    phase = PH_MARCH;
                    t = 0; lastSafeT = 0; lo = 0; hi = 0; tEval = 0; iter = 0; refineK = 0; crawlRun = 0;
                } else {
                    more = false;
                }
            }
        }
        if (!__any(phase != PH_DONE)) break;
        st.trips += 1;
        const float bestToi = __uint_as_float((unsigned)(*(volatile unsigned long long*)&sh.rayKey[myRay] >> 32));
        if (phase == PH_MARCH) {
            if (iter >= maxIter || t > len || lastSafeT > bestToi) phase = PH_DONE;
            else { iter += 1; tEval = t; }
        } else if (phase == PH_REFINE) {
            if (lo > bestToi) phase = PH_DONE;
            else tEval = 0.5f * (lo + hi);
        }
        // ---- speculative crawl (see groupSweep): once the workgroup's queue has run dry, the idle lanes of THIS wavefront evaluate
        // a creeping item at t_{k+1}, t_{k+2}, ... in the trip in which its owner evaluates t_k. Shuffles and ballots only: the
        // wavefronts of the workgroup run different numbers of trips, there is no barrier in here.
        const unsigned long long crawlMask = __ballot(phase == PH_MARCH && crawlRun >= 3);
        const unsigned long long helpMask = __ballot(phase == PH_DONE && !more);
        const int nHelp = __popcll(helpMask);
        const bool spec = crawlMask != 0 && nHelp >= kWave / 2;
        const int owner = spec ? __ffsll((long long)crawlMask) - 1 : 0;
        const bool helper = spec && ((helpMask >> lane) & 1);
        const int h = prefixCount(helpMask); // helper h evaluates t_{k+1+h}
        bool cEval = phase != PH_DONE;
        float cT = tEval, cPrev = 0;
        F3 cFrom = sh.rayFrom[myRay], cDir = sh.rayDir[myRay], c0 = tri.v0, c1 = tri.v1, c2 = tri.v2;
        if (spec) {
            const int bRay = __shfl(myRay, owner, kWave), bIter = __shfl(iter, owner, kWave), bMaxIter = __shfl(maxIter, owner, kWave);
            const float bT = __shfl(t, owner, kWave), bLen = __shfl(len, owner, kWave);
            const F3 b0{__shfl(tri.v0.x, owner, kWave), __shfl(tri.v0.y, owner, kWave), __shfl(tri.v0.z, owner, kWave)};
            const F3 b1{__shfl(tri.v1.x, owner, kWave), __shfl(tri.v1.y, owner, kWave), __shfl(tri.v1.z, owner, kWave)};
            const F3 b2{__shfl(tri.v2.x, owner, kWave), __shfl(tri.v2.y, owner, kWave), __shfl(tri.v2.z, owner, kWave)};
            if (helper) {
                c0 = b0; c1 = b1; c2 = b2;
                cFrom = sh.rayFrom[bRay]; cDir = sh.rayDir[bRay];
                cT = bT;
            }
            for (int k = 0; k < nHelp; ++k) if (helper && k <= h) { cPrev = cT; cT += minAdvance; } // the march's own sequential additions
            if (helper) cEval = (bIter + h < bMaxIter) && !(cT > bLen); // loop head of :1303-1307 for this evaluation
        }
        float dist = 0;
        F3 segP{0, 0, 0}, triP{0, 0, 0};
        if (cEval) dist = segmentTriangleDistance(cFrom + cDir * cT, halfHeight, c0, c1, c2, segP, triP);
        if (phase != PH_DONE) {
            evals += 1;
            const F3 dir = cDir;
            if (phase == PH_MARCH) {
                if (dist <= radius + contactEps) {
                    float k0 = smax(0.0f, smin(lastSafeT, len));
                    float k1 = smax(0.0f, smin(t, len));
                    lo = smin(k0, k1);
                    hi = smax(k0, k1);
                    if (hi - lo < 1e-5f) { phase = PH_FINAL; tEval = hi; }
                    else { phase = PH_REFINE; refineK = 0; }
                    crawlRun = 0;
                } else {
                    lastSafeT = t;
                    float advance = smax(dist - radius, minAdvance);
                    crawlRun = advance == minAdvance ? crawlRun + 1 : 0;
                    if (advance <= 0) t += minAdvance; else t += advance;
                }
            } else if (phase == PH_REFINE) {
                if (dist <= radius) hi = tEval; else lo = tEval;
                refineK += 1;
                if (refineK == 10) { phase = PH_FINAL; tEval = hi; }
            } else { // PH_FINAL
                float tHit = tEval;
                F3 nrm;
                if (dist < 1e-6f) nrm = dot(triNormal, dir) > 0 ? -triNormal : triNormal;
                else nrm = normalize(segP - triP);
                F3 triN = triNormal;
                if (dot(triN, nrm) < 0) triN = -triN;
                phase = PH_DONE;
                bool ok = tHit < len;
                if (ok && blockingOnly) {
                    F3 delta = sh.rayDelta[myRay];
                    ok = !(dot(delta, nrm) >= 0) && !(dot(delta, triN) >= 0);
                }
                if (ok && hasMinNormalY) ok = !(triN.y < minNormalY);
                if (ok) atomicMin(&sh.rayKey[myRay], ((unsigned long long)__float_as_uint(tHit) << 32) | (unsigned)tri.rank);
            }
        }
        if (spec) {
            // did the owner's own evaluation of this trip creep (still marching, advance == minAdvance)? then the helpers' points are its
            const int ownerCrept = __shfl((int)(phase == PH_MARCH && crawlRun > 0), owner, kWave);
            if (ownerCrept) {
                const bool contact = helper && cEval && dist <= radius + contactEps;
                const float adv = smax(dist - radius, minAdvance);
                const bool breaks = helper && cEval && !contact && adv != minAdvance;
                const unsigned long long ev = __ballot(helper && (!cEval || contact || breaks));
                const int lastHelper = 63 - __clzll((long long)helpMask);
                const int jl = ev ? __ffsll((long long)ev) - 1 : lastHelper; // the helper whose evaluation decides
                const int rJ = __shfl(h, jl, kWave);
                const float tJ = __shfl(cT, jl, kWave), tPrev = __shfl(cPrev, jl, kWave), advJ = __shfl(adv, jl, kWave);
                const int evalJ = __shfl((int)cEval, jl, kWave), contactJ = __shfl((int)contact, jl, kWave);
                if (lane == owner) {
                    if (!ev) {                    // every helper found another creeping step
                        iter += nHelp; evals += nHelp; crawlRun += nHelp; lastSafeT = tJ; t = tJ + minAdvance;
                    } else if (!evalJ) {          // the loop ends at helper rJ's evaluation (budget or t > maxDistance): the next head sees it
                        iter += rJ; evals += rJ; lastSafeT = tPrev; t = tJ; crawlRun = 0;
                    } else if (contactJ) {        // contact: refineTOI(t0: lastSafeT, t1: t) :1361-1377
                        iter += rJ + 1; evals += rJ + 1; lastSafeT = tPrev; t = tJ; crawlRun = 0;
                        float k0 = smax(0.0f, smin(lastSafeT, len));
                        float k1 = smax(0.0f, smin(t, len));
                        lo = smin(k0, k1);
                        hi = smax(k0, k1);
                        if (hi - lo < 1e-5f) { phase = PH_FINAL; tEval = hi; }
                        else { phase = PH_REFINE; refineK = 0; }
                    } else {                      // this evaluation advances further than minAdvance: back to the ordinary march
                        iter += rJ + 1; evals += rJ + 1; lastSafeT = tJ; t = tJ + advJ; crawlRun = 0;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) evals += __shfl_xor(evals, o, kWave);
    if (lane == 0) atomicAdd(&hv.evalSum, evals);
    st.evals += evals >> 6; // per-lane convention of WaveStats: the kernel epilogue sums over the 64 lanes
}

// Waves 1..7 of a heavy workgroup: match every barrier of wave 0, join the sweeps it announces, leave on HCMD_EXIT.
// The iteration bound is a safety net only (a finished wave no longer takes part in barriers).
__device__ __forceinline__ void heavyHelperLoop(const DevCollision& col, WaveStats& st) {
    for (int guard = 0; guard < (1 << 22); ++guard) {
        __syncthreads();
        const int cmd = *(volatile int*)&hv.cmd;
        if (cmd == HCMD_EXIT) return;
        if (cmd == HCMD_MARCH) {
            heavyMarch(col, st);
            __syncthreads();
        }
    }
}

// Casts sh.rayFrom/rayDelta[0..rayCount) (same capsule, same filters) in one pass. Each ray is an
// independent capsuleCastCombined call of the reference; a work item is a (ray, triangle) pair whose
// triangle AABB overlaps THAT ray's swept box, so every ray sees exactly its own candidate set.
// Results: sh.rayKey[r] != initial  <=>  hit, record in sh.rayRec[r].
// farRay (multi-wave kernel only): the ground probe's 200-unit fall probe, which shares origin and capsule with the snap cast — see
// the near / far passes of move_group_kernel.
template <bool HEAVY = false>
__device__ __forceinline__ void waveCastRays(const DevCollision& col, float radius, float halfHeight, bool blockingOnly,
                                             bool hasMinNormalY, float minNormalY, uint32_t mask, WaveStats& st, int farRay = -1) {
    const int lane = laneId();
    const int R = sh.rayCount;
    // per-ray setup (:1021-1035), lanes 0..R-1 in parallel
    if (lane < R) {
        F3 from = sh.rayFrom[lane], delta = sh.rayDelta[lane];
        float len = length(delta);
        bool valid = !(len < 1e-6f) && col.root >= 0; // :987-988, :1020
        F3 dir = delta / len;
        F3 up{0, 1, 0};
        F3 a0 = from + up * halfHeight, b0 = from - up * halfHeight;
        F3 a1 = a0 + delta, b1 = b0 + delta;
        F3 minP = vmin(vmin(a0, b0), vmin(a1, b1));
        F3 maxP = vmax(vmax(a0, b0), vmax(a1, b1));
        F3 ext{radius, radius, radius};
        sh.rayMin[lane] = minP - ext; sh.rayMax[lane] = maxP + ext;
        sh.rayDir[lane] = dir; sh.rayLen[lane] = len; sh.rayValid[lane] = valid ? 1 : 0;
        sh.rayVertical[lane] = (delta.x == 0.0f && delta.z == 0.0f) ? 1 : 0;
        const float minAdv = smax(radius * 0.02f, 1e-4f);
        int maxIter = (int)ceilf(len / minAdv) + 1; // :1296
        sh.rayMaxIter[lane] = maxIter < 256 ? maxIter : 256;
        // a hit must have toi < len (:1084 with bestT = len): start the key at (len, +inf rank)
        sh.rayKey[lane] = ((unsigned long long)__float_as_uint(len) << 32) | 0xffffffffull;
    }
    __syncthreads();
    // union box of the valid rays drives the traversal
    F3 minP{kFloatMax, kFloatMax, kFloatMax}, maxP{-kFloatMax, -kFloatMax, -kFloatMax};
    int nValid = 0;
    for (int r = 0; r < R; ++r) {
        if (!sh.rayValid[r]) continue;
        minP = vmin(minP, sh.rayMin[r]); maxP = vmax(maxP, sh.rayMax[r]);
        nValid += 1;
    }
    if (nValid == 0) return;
    st.queries += nValid;

    const float minAdvance = smax(radius * 0.02f, 1e-4f); // :1295
    const float contactEps = 1e-5f;
    long long cycTrav = 0, cycSweep = 0; (void)cycTrav; (void)cycSweep;
    int stackSize = initTraversal(col), rangeCount = 0, candCount = 0, itemCount = 0;
    __syncthreads();

    if constexpr (HEAVY) {
        // wave 0 of a heavy workgroup: gather work items in chunks, let all eight waves sweep each chunk
        if (lane == 0) {
            hv.radius = radius; hv.halfHeight = halfHeight; hv.minNormalY = minNormalY;
            hv.blockingOnly = blockingOnly ? 1 : 0; hv.hasMinNormalY = hasMinNormalY ? 1 : 0;
        }
        // one traversal over [bMin, bMax] making items for rays [r0, r1); triangles whose AABB overlaps the skip box belong to an
        // earlier pass of the same rays
        auto runPass = [&](F3 bMin, F3 bMax, int r0, int r1, bool skip, F3 sMin, F3 sMax) {
            int hCount = 0;
            while (true) {
                while ((stackSize > 0 || rangeCount > 0 || candCount > 0) && hCount <= kHeavyItemCap - kWave * kMaxRays) {
                    while ((stackSize > 0 || rangeCount > 0) && candCount < kWave) expandNodes(col, bMin, bMax, mask, stackSize, rangeCount, candCount, st);
                    int n = candCount < kWave ? candCount : kWave;
                    candCount -= n;
                    st.candidates += n;
                    int slot = -1;
                    F3 v0{0, 0, 0}, v1{0, 0, 0}, v2{0, 0, 0};
                    if (lane < n) {
                        slot = sh.cand[candCount + lane];
                        const float4* tp = reinterpret_cast<const float4*>(col.tris + slot);
                        float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                        v0 = F3{t0.x, t0.y, t0.z}; v1 = F3{t0.w, t1.x, t1.y}; v2 = F3{t1.z, t1.w, t2.x};
                        if (hasMinNormalY && tooSteepForGroundCast(v0, v1, v2, minNormalY)) { slot = -1; st.pruned += r1 - r0; }
                    }
                    const F3 bmin = vmin(v0, vmin(v1, v2)), bmax = vmax(v0, vmax(v1, v2));
                    if (skip && !boxDisjoint(bmin, bmax, sMin, sMax)) slot = -1;
                    for (int r = r0; r < r1; ++r) {
                        bool c = slot >= 0 && sh.rayValid[r] && !boxDisjoint(bmin, bmax, sh.rayMin[r], sh.rayMax[r]);
                        if (c && sh.rayVertical[r]) { c = !verticalSweepMisses(sh.rayFrom[r].x, sh.rayFrom[r].z, radius, v0, v1, v2); st.pruned += c ? 0 : 1; }
                        unsigned long long mc = __ballot(c);
                        if (c) hv.items[hCount + prefixCount(mc)] = (int)(((unsigned)r << kItemRayShift) | (unsigned)slot);
                        hCount += __popcll(mc);
                    }
                    __syncthreads();
                }
                if (hCount == 0) break;
                if (lane == 0) { hv.count = hCount; hv.cursor = 0; hv.cmd = HCMD_MARCH; }
                __syncthreads();
                heavyMarch(col, st);
                __syncthreads();
                if (lane == 0) hv.cmd = HCMD_NONE;
                hCount = 0;
                if (!(stackSize > 0 || rangeCount > 0 || candCount > 0)) break;
            }
        };
        // near pass: with a fall probe among the rays (and a snap cast beside it) the traversal box is everybody else's; the fall
        // probe takes its items from the same candidates
        const bool twoPass = farRay == 1 && R >= 2 && sh.rayValid[0] && sh.rayValid[1]; // ray 0: the snap cast from the same origin
        F3 nMin = minP, nMax = maxP;
        if (twoPass) {
            nMin = F3{kFloatMax, kFloatMax, kFloatMax}; nMax = F3{-kFloatMax, -kFloatMax, -kFloatMax};
            for (int r = 0; r < R; ++r)
                if (r != farRay && sh.rayValid[r]) { nMin = vmin(nMin, sh.rayMin[r]); nMax = vmax(nMax, sh.rayMax[r]); }
        }
        runPass(nMin, nMax, 0, R, false, nMin, nMax);
        if (twoPass) {
            // far pass: the rest of the fall probe's box, clipped to the best hit so far — a triangle entirely below the capsule's
            // lowest point at t = bestToi cannot be touched before bestToi
            const int r = farRay;
            const F3 from = sh.rayFrom[r];
            const float len = sh.rayLen[r];
            float reach = len;
            if (rayHit(r)) reach = smin(len, __uint_as_float((unsigned)(sh.rayKey[r] >> 32)) + (0.05f + 1e-4f * fabsf(from.y)));
            const F3 ext{radius, radius, radius};
            const F3 fMin = F3{from.x, from.y - halfHeight - reach, from.z} - ext;
            const F3 fMax = F3{from.x, from.y + halfHeight, from.z} + ext;
            if (!(fMin.y >= nMin.y)) { // otherwise inside the near box (same x / z extent: same origin and capsule)
                __syncthreads();
                if (lane == 0) { sh.rayMin[r] = vmax(sh.rayMin[r], fMin); sh.rayMax[r] = vmin(sh.rayMax[r], fMax); }
                stackSize = initTraversal(col); rangeCount = 0; candCount = 0;
                __syncthreads();
                runPass(sh.rayMin[r], sh.rayMax[r], r, r + 1, true, nMin, nMax);
            }
        }
        // rebuild each hit ray's record from its winning (toi, visit rank): the FINAL evaluation of that triangle again
        if (lane < R && rayHit(lane)) {
            const unsigned long long key = sh.rayKey[lane];
            const float tHit = __uint_as_float((unsigned)(key >> 32));
            const Tri tri = loadTri(col, col.slotOfRank[(unsigned)(key & 0xffffffffull)]);
            const F3 from = sh.rayFrom[lane], dir = sh.rayDir[lane];
            const F3 triNormal = normalize(cross(tri.v1 - tri.v0, tri.v2 - tri.v0));
            F3 center = from + dir * tHit;
            F3 segP, triP;
            float dist = segmentTriangleDistance(center, halfHeight, tri.v0, tri.v1, tri.v2, segP, triP);
            F3 nrm;
            if (dist < 1e-6f) nrm = dot(triNormal, dir) > 0 ? -triNormal : triNormal;
            else nrm = normalize(segP - triP);
            F3 triN = triNormal;
            if (dot(triN, nrm) < 0) triN = -triN;
            sh.rayRec[lane] = CastRec{tHit, triP, nrm, triN, tri.triIndex};
        }
        __syncthreads();
        return;
    }
    // Streaming sweep: every lane owns one (ray, triangle) work item at a time and pulls the next one from the
    // LDS queue the moment its own finishes, so a query with hundreds of candidates keeps all 64 lanes marching
    // instead of waiting, batch after batch, for each batch's slowest lane (conservative advancement runs up to
    // 256 dependent iterations per triangle). The result does not depend on the processing order: the winner is
    // the minimum (toi, visit rank) key.
    int phase = PH_DONE;
    int myRay = 0;
    Tri tri;
    tri.v0 = tri.v1 = tri.v2 = F3{0, 0, 0}; tri.triIndex = -1; tri.rank = 0x7fffffff;
    F3 triNormal{0, 0, 0};
    float t = 0, lastSafeT = 0, lo = 0, hi = 0, tEval = 0, len = 0, bestToi = 0;
    int iter = 0, refineK = 0, maxIter = 0;
    while (true) {
        const unsigned long long idleMask = __ballot(phase == PH_DONE);
        const int nIdle = __popcll(idleMask);
        const bool moreCandidates = stackSize > 0 || rangeCount > 0 || candCount > 0;
        // 1. top the queue up when idle lanes cannot be fed from it: traverse to a batch of candidates and turn
        //    them into (ray, slot) items
        if (itemCount < nIdle && moreCandidates && (nIdle >= kRefillIdle || nIdle == kWave)) {
            SGE_T0();
            while ((stackSize > 0 || rangeCount > 0) && candCount < kWave) expandNodes(col, minP, maxP, mask, stackSize, rangeCount, candCount, st);
            int n = candCount < kWave ? candCount : kWave;
            candCount -= n;
            st.candidates += n;
            int slot = -1;
            F3 v0{0, 0, 0}, v1{0, 0, 0}, v2{0, 0, 0};
            if (lane < n) {
                slot = sh.cand[candCount + lane];
                const float4* tp = reinterpret_cast<const float4*>(col.tris + slot);
                float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                v0 = F3{t0.x, t0.y, t0.z}; v1 = F3{t0.w, t1.x, t1.y}; v2 = F3{t1.z, t1.w, t2.x};
                if (hasMinNormalY && tooSteepForGroundCast(v0, v1, v2, minNormalY)) { slot = -1; st.pruned += R; }
            }
            const F3 bmin = vmin(v0, vmin(v1, v2)), bmax = vmax(v0, vmax(v1, v2));
            for (int r = 0; r < R; ++r) {
                bool c = slot >= 0 && sh.rayValid[r] && !boxDisjoint(bmin, bmax, sh.rayMin[r], sh.rayMax[r]);
                if (c && sh.rayVertical[r]) { c = !verticalSweepMisses(sh.rayFrom[r].x, sh.rayFrom[r].z, radius, v0, v1, v2); st.pruned += c ? 0 : 1; }
                unsigned long long mc = __ballot(c);
                if (c) sh.items[itemCount + prefixCount(mc)] = (int)(((unsigned)r << kItemRayShift) | (unsigned)slot);
                itemCount += __popcll(mc);
            }
            __syncthreads();
            SGE_T1(cycTrav);
            continue;
        }
        // 2. idle lanes take the newest items
        if (nIdle > 0 && itemCount > 0) {
            const int take = nIdle < itemCount ? nIdle : itemCount;
            const int p = prefixCount(idleMask);
            if (phase == PH_DONE && p < take) {
                const int it = sh.items[itemCount - 1 - p];
                myRay = (unsigned)it >> kItemRayShift;
                tri = loadTri(col, it & kItemSlotMask);
                triNormal = normalize(cross(tri.v1 - tri.v0, tri.v2 - tri.v0));
                len = sh.rayLen[myRay];
                maxIter = sh.rayMaxIter[myRay];
                bestToi = __uint_as_float((unsigned)(sh.rayKey[myRay] >> 32));
                phase = PH_MARCH;
                t = 0; lastSafeT = 0; lo = 0; hi = 0; tEval = 0; iter = 0; refineK = 0;
            }
            itemCount -= take;
            __syncthreads(); // the slots just read may be rewritten by the next top-up
        } else if (nIdle == kWave) {
            break; // nothing marching, nothing queued, nothing left to traverse
        }
        // 3. one trip: one distance evaluation per marching lane
        st.trips += 1;
        SGE_T0();
        if (phase == PH_MARCH) {
            // loop head of :1303-1307 — iteration budget, then `if t > maxDistance return nil`
            if (iter >= maxIter || t > len || lastSafeT > bestToi) phase = PH_DONE;
            else { iter += 1; tEval = t; }
        } else if (phase == PH_REFINE) {
            if (lo > bestToi) phase = PH_DONE;
            else tEval = 0.5f * (lo + hi);
        }
        bool finished = false;
        unsigned long long myKey = ~0ull;
        CastRec rec;
        rec.toi = 0; rec.position = rec.normal = rec.triNormal = F3{0, 0, 0}; rec.triIndex = -1;
        if (phase != PH_DONE) {
            st.evals += 1;
            const F3 from = sh.rayFrom[myRay], dir = sh.rayDir[myRay];
            F3 center = from + dir * tEval;
            F3 segP, triP;
            float dist = segmentTriangleDistance(center, halfHeight, tri.v0, tri.v1, tri.v2, segP, triP);
            if (phase == PH_MARCH) {
                if (dist <= radius + contactEps) {
                    // refineTOI(t0: lastSafeT, t1: t) :1361-1377
                    float c0 = smax(0.0f, smin(lastSafeT, len));
                    float c1 = smax(0.0f, smin(t, len));
                    lo = smin(c0, c1);
                    hi = smax(c0, c1);
                    if (hi - lo < 1e-5f) { phase = PH_FINAL; tEval = hi; }
                    else { phase = PH_REFINE; refineK = 0; }
                } else {
                    lastSafeT = t;
                    float advance = smax(dist - radius, minAdvance);
                    if (advance <= 0) t += minAdvance; else t += advance;
                }
            } else if (phase == PH_REFINE) {
                if (dist <= radius) hi = tEval; else lo = tEval;
                refineK += 1;
                if (refineK == 10) { phase = PH_FINAL; tEval = hi; }
            } else { // PH_FINAL :1325-1346
                float tHit = tEval;
                F3 nrm;
                if (dist < 1e-6f) nrm = dot(triNormal, dir) > 0 ? -triNormal : triNormal;
                else nrm = normalize(segP - triP);
                F3 triN = triNormal;
                if (dot(triN, nrm) < 0) triN = -triN;
                phase = PH_DONE;
                // acceptance filters of capsuleCastBVH :1084-1097 (toi < len; blocking; minNormalY)
                bool ok = tHit < len;
                if (ok && blockingOnly) {
                    F3 delta = sh.rayDelta[myRay];
                    ok = !(dot(delta, nrm) >= 0) && !(dot(delta, triN) >= 0);
                }
                if (ok && hasMinNormalY) ok = !(triN.y < minNormalY);
                if (ok) {
                    rec = CastRec{tHit, triP, nrm, triN, tri.triIndex};
                    myKey = ((unsigned long long)__float_as_uint(tHit) << 32) | (unsigned)tri.rank;
                    atomicMin(&sh.rayKey[myRay], myKey);
                    finished = true;
                }
            }
        }
        // publish: the lane holding its ray's best key records the hit; every lane tightens its prune bound
        if (__any(finished)) {
            __syncthreads();
            if (finished && sh.rayKey[myRay] == myKey) sh.rayRec[myRay] = rec;
            bestToi = __uint_as_float((unsigned)(sh.rayKey[myRay] >> 32));
        }
        SGE_T1(cycSweep);
    }
    __syncthreads();
#ifdef SGE_CCD_TIMING
    if (lane == 0) { atomicAdd(&g_cycTraverse, (unsigned long long)cycTrav); atomicAdd(&g_cycSweep, (unsigned long long)cycSweep); }
#endif
}


// ---------------------------------------------------------------------------
// capsuleOverlapAll over the static set (CollisionQuery.swift:852-882, 1201-1283):
// the first `maxHits` overlapping triangles in visit order = the maxHits lowest ranks
// ---------------------------------------------------------------------------
__device__ __forceinline__ int waveCapsuleOverlapAll(const DevCollision& col, F3 from, float radius,
                                                  float halfHeight, int maxHits, uint32_t mask, WaveStats& st) {
    const int lane = laneId();
    if (col.root < 0) return 0;
    st.queries += 1;
    F3 up{0, 1, 0};
    F3 a0 = from + up * halfHeight, b0 = from - up * halfHeight;
    F3 minP = vmin(a0, b0), maxP = vmax(a0, b0);
    F3 ext{radius, radius, radius};
    minP = minP - ext; maxP = maxP + ext;
    int count = 0; // entries in sOvl, sorted by rank
    int stackSize = initTraversal(col), rangeCount = 0, candCount = 0;
    __syncthreads();
    while (true) {
        while ((stackSize > 0 || rangeCount > 0) && candCount < kWave) expandNodes(col, minP, maxP, mask, stackSize, rangeCount, candCount, st);
        if (candCount == 0) break;
        int n = candCount < kWave ? candCount : kWave;
        candCount -= n;
        st.candidates += n;
        bool pending = false;
        OverlapRec rec;
        rec.depth = 0; rec.position = rec.normal = rec.triNormal = F3{0, 0, 0}; rec.triIndex = -1; rec.rank = 0x7fffffff;
        if (lane < n) {
            Tri tri = loadTri(col, sh.cand[candCount + lane]);
            st.evals += 1;
            F3 segP, triP;
            float dist = segmentTriangleDistance(from, halfHeight, tri.v0, tri.v1, tri.v2, segP, triP);
            if (!(dist >= radius)) { // :1253
                F3 triNormal = normalize(cross(tri.v1 - tri.v0, tri.v2 - tri.v0));
                F3 nn = dist < 1e-6f ? triNormal : normalize(segP - triP);
                F3 triN = triNormal;
                if (dot(triN, nn) < 0) triN = -triN;
                rec.depth = radius - dist; rec.position = triP; rec.normal = nn; rec.triNormal = triN;
                rec.triIndex = tri.triIndex; rec.rank = tri.rank;
                pending = true;
            }
        }
        if (__any(pending)) {
            // merge: existing sorted list (sOvl[0..count)) with this batch's hits extracted in rank order
            int ie = 0, k = 0;
            for (; k < maxHits; ++k) {
                unsigned pr = pending ? (unsigned)rec.rank : 0xffffffffu;
                unsigned long long pmin = waveMinU64((unsigned long long)pr);
                unsigned er = ie < count ? (unsigned)sOvl[ie].rank : 0xffffffffu;
                if (pmin == 0xffffffffull && er == 0xffffffffu) break;
                if ((unsigned)pmin < er) {
                    if (pending && (unsigned)rec.rank == (unsigned)pmin) { sOvlTmp[k] = rec; pending = false; }
                } else {
                    if (lane == 0) sOvlTmp[k] = sOvl[ie];
                    ie += 1;
                }
            }
            __syncthreads();
            if (lane < k) sOvl[lane] = sOvlTmp[lane];
            count = k;
            __syncthreads();
        }
    }
    return count;
}

// ---------------------------------------------------------------------------
// contact cache (Systems.swift:1102-1205)
// ---------------------------------------------------------------------------
__device__ __forceinline__ F3 ld3(const float* p) { return F3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, F3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

__device__ __forceinline__ void cacheDecay(sge_controller_state& c) { // :1105
    if (c.sideContactFrames > 0) c.sideContactFrames -= 1;
    if (c.manifoldFrames > 0) {
        c.manifoldFrames -= 1;
        if (c.manifoldFrames == 0) {
            c.manifoldCount = 0;
            c.manifoldFrames = 0;
            st3(c.sideContactNormal, F3{0, 0, 0});
        }
    }
}
__device__ __forceinline__ bool cachedNormal(const sge_controller_state& c, int triangleIndex, F3& out) { // :1169
#pragma unroll
    for (int i = 0; i < SGE_MANIFOLD_MAX; ++i)
        if (i < c.manifoldCount && c.manifoldTriangles[i] == triangleIndex) { out = ld3(c.manifoldNormals[i]); return true; }
    return false;
}
__device__ __forceinline__ void manifoldUpdate(sge_controller_state& c, int triangleIndex, F3 normal) { // :1177
    F3 n = normal;
    if (lengthSq(n) < 1e-8f) return;
    c.manifoldFrames = 8;
#pragma unroll
    for (int i = 0; i < SGE_MANIFOLD_MAX; ++i) {
        if (i < c.manifoldCount && c.manifoldTriangles[i] == triangleIndex) {
            F3 cached = ld3(c.manifoldNormals[i]);
            if (dot(cached, n) < 0) n = -n;
            const float blend = 0.25f;
            F3 combined = normalize(cached * (1 - blend) + n * blend);
            st3(c.manifoldNormals[i], combined);
            st3(c.sideContactNormal, combined);
            return;
        }
    }
    if (c.manifoldCount >= SGE_MANIFOLD_MAX) c.manifoldCount -= 1;
#pragma unroll
    for (int i = SGE_MANIFOLD_MAX - 1; i > 0; --i) {
        if (i <= c.manifoldCount) {
            c.manifoldTriangles[i] = c.manifoldTriangles[i - 1];
            st3(c.manifoldNormals[i], ld3(c.manifoldNormals[i - 1]));
        }
    }
    c.manifoldTriangles[0] = triangleIndex;
    st3(c.manifoldNormals[0], normalize(n));
    c.manifoldCount += 1;
    st3(c.sideContactNormal, ld3(c.manifoldNormals[0]));
}
__device__ __forceinline__ void cacheRecord(sge_controller_state& c, int triangleIndex, F3 normal, bool isSideContact) { // :1122
    manifoldUpdate(c, triangleIndex, normal);
    if (isSideContact) {
        st3(c.sideContactNormal, normalize(normal));
        c.sideContactFrames = 3;
    }
}

// ---------------------------------------------------------------------------
// capsule-capsule sweep (Systems.swift:1417-1590)
// ---------------------------------------------------------------------------
struct Interval { float s, e; bool ok; };
__device__ __forceinline__ Interval clampInterval(float start, float end) {
    float s = smax(start, 0.0f), e = smin(end, 1.0f);
    if (e < s) return Interval{0, 0, false};
    return Interval{s, e, true};
}
__device__ __forceinline__ Interval intervalGreaterEqual(float y0, float vy, float threshold) {
    if (fabsf(vy) < 1e-6f) return y0 >= threshold ? Interval{0, 1, true} : Interval{0, 0, false};
    float t = (threshold - y0) / vy;
    if (vy > 0) return clampInterval(t, 1);
    return clampInterval(0, t);
}
__device__ __forceinline__ Interval intervalLessEqual(float y0, float vy, float threshold) {
    if (fabsf(vy) < 1e-6f) return y0 <= threshold ? Interval{0, 1, true} : Interval{0, 0, false};
    float t = (threshold - y0) / vy;
    if (vy > 0) return clampInterval(0, t);
    return clampInterval(t, 1);
}
__device__ __forceinline__ bool earliestRoot(float A, float B, float C, float tMin, float tMax, float& out) {
    const float eps = 1e-6f;
    if (fabsf(A) < eps) {
        if (fabsf(B) < eps) { if (C <= 0) { out = tMin; return true; } return false; }
        float t = -C / B;
        if (t >= tMin && t <= tMax) { out = t; return true; }
        return false;
    }
    float disc = B * B - 4 * A * C;
    if (disc < 0) return false;
    float sqrtD = sqrtf(disc);
    float inv2A = 1 / (2 * A);
    float t0 = (-B - sqrtD) * inv2A;
    float t1 = (-B + sqrtD) * inv2A;
    float enter = smin(t0, t1), exit = smax(t0, t1);
    float s = smax(enter, tMin), e = smin(exit, tMax);
    if (e >= s) { out = s; return true; }
    return false;
}
__device__ __forceinline__ float capsuleSeparationY(float yRel, float hSum) {
    if (yRel > hSum) return yRel - hSum;
    if (yRel < -hSum) return yRel + hSum;
    return 0;
}
__device__ __forceinline__ F3 capsuleHitNormal(F3 rel, float hSum) {
    float sepY = capsuleSeparationY(rel.y, hSum);
    F3 sep{rel.x, sepY, rel.z};
    float lenSq = lengthSq(sep);
    if (lenSq > 1e-8f) return sep / sqrtf(lenSq);
    F3 lateral{rel.x, 0, rel.z};
    float l2 = lengthSq(lateral);
    if (l2 > 1e-8f) return lateral / sqrtf(l2);
    return F3{1, 0, 0};
}
__device__ __forceinline__ bool capsuleCapsuleSweep(F3 from, F3 delta, float radius, float halfHeight, F3 otherPos,
                                                    F3 otherDelta, float otherRadius, float otherHalfHeight,
                                                    float& toiOut, F3& normalOut) {
    F3 relStart = from - otherPos;
    F3 relDelta = delta - otherDelta;
    float rSum = radius + otherRadius, hSum = halfHeight + otherHalfHeight;
    float relLen = length(relDelta), moveLen = length(delta);
    if (relLen < 1e-6f) {
        float sepY = capsuleSeparationY(relStart.y, hSum);
        float distSq = relStart.x * relStart.x + relStart.z * relStart.z + sepY * sepY;
        if (distSq <= rSum * rSum) { toiOut = 0; normalOut = capsuleHitNormal(relStart, hSum); return true; }
        return false;
    }
    float y0 = relStart.y, vy = relDelta.y, vx = relDelta.x, vz = relDelta.z, r0x = relStart.x, r0z = relStart.z;
    bool have = false;
    float bestT = 0, t;
    Interval upper = intervalGreaterEqual(y0, vy, hSum);
    if (upper.ok) {
        float A = vx * vx + vz * vz + vy * vy;
        float B = 2 * (r0x * vx + r0z * vz + (y0 - hSum) * vy);
        float C = r0x * r0x + r0z * r0z + (y0 - hSum) * (y0 - hSum) - rSum * rSum;
        if (earliestRoot(A, B, C, upper.s, upper.e, t)) { bestT = t; have = true; }
    }
    Interval lower = intervalLessEqual(y0, vy, -hSum);
    if (lower.ok) {
        float A = vx * vx + vz * vz + vy * vy;
        float B = 2 * (r0x * vx + r0z * vz + (y0 + hSum) * vy);
        float C = r0x * r0x + r0z * r0z + (y0 + hSum) * (y0 + hSum) - rSum * rSum;
        if (earliestRoot(A, B, C, lower.s, lower.e, t)) { if (!have || t < bestT) { bestT = t; have = true; } }
    }
    if (fabsf(vy) < 1e-6f) {
        if (fabsf(y0) <= hSum) {
            float A = vx * vx + vz * vz;
            float B = 2 * (r0x * vx + r0z * vz);
            float C = r0x * r0x + r0z * r0z - rSum * rSum;
            if (earliestRoot(A, B, C, 0, 1, t)) { if (!have || t < bestT) { bestT = t; have = true; } }
        }
    } else {
        float t1 = (hSum - y0) / vy, t2 = (-hSum - y0) / vy;
        Interval ov = clampInterval(smin(t1, t2), smax(t1, t2));
        if (ov.ok) {
            float A = vx * vx + vz * vz;
            float B = 2 * (r0x * vx + r0z * vz);
            float C = r0x * r0x + r0z * r0z - rSum * rSum;
            if (earliestRoot(A, B, C, ov.s, ov.e, t)) { if (!have || t < bestT) { bestT = t; have = true; } }
        }
    }
    if (!have) return false;
    F3 relAtHit = relStart + relDelta * bestT;
    normalOut = capsuleHitNormal(relAtHit, hSum);
    toiOut = bestT * moveLen;
    return true;
}

// AgentSweepSolver.bestHit (Systems.swift:1053-1091). The reference loops over every
// solid agent; only agents whose XZ footprint can be reached within this segment can
// hit, so the wave scans the grid cells covering that reach, 64 agents at a time, and
// reduces on (toi, agent index) — the index reproduces the reference's first-wins
// `<` over its (index-ordered) snapshot.
__device__ __forceinline__ bool waveAgentBestHit(const DevAgents& ag, F3 position, F3 remaining, float remainingLen,
                                              float baseMoveLen, float dt, int selfIndex, float selfRadius,
                                              float halfHeight, float& toiOut, F3& normalOut) {
    const int lane = laneId();
    const AgentGrid G = *ag.grid; // wave-uniform
    if (G.nx == 0) return false;  // no solid agent in the snapshot
    const float maxAgentRadius = G.maxRadius, maxAgentSpeed = G.maxSpeed;
    float timeScale = baseMoveLen > 1e-6f ? smin(remainingLen / baseMoveLen, 1.0f) : 1.0f;
    float segmentDt = dt * timeScale;
    // conservative XZ reach: own move + the fastest other agent's move + both radii
    float reach = remainingLen + maxAgentSpeed * segmentDt + selfRadius + maxAgentRadius + 1e-3f;
    int cx0 = (int)floorf((position.x - reach - G.originX) * G.invCell), cx1 = (int)floorf((position.x + reach - G.originX) * G.invCell);
    int cz0 = (int)floorf((position.z - reach - G.originZ) * G.invCell), cz1 = (int)floorf((position.z + reach - G.originZ) * G.invCell);
    cx0 = cx0 < 0 ? 0 : cx0; cz0 = cz0 < 0 ? 0 : cz0;
    cx1 = cx1 >= G.nx ? G.nx - 1 : cx1; cz1 = cz1 >= G.nz ? G.nz - 1 : cz1;
    // a character carried or pushed beyond the snapshot's bounds by more than its reach meets nobody (and must not index
    // cellStart past the grid)
    if (cx0 > cx1 || cz0 > cz1) return false;
    unsigned long long bestKey = ~0ull;
    F3 bestN{0, 0, 0};
    for (int cz = cz0; cz <= cz1; ++cz) {
        // cells of one row are contiguous in cellStart: scan [start(cx0), start(cx1+1))
        int s = ag.cellStart[cz * G.nx + cx0], e = ag.cellStart[cz * G.nx + cx1 + 1];
        for (int base = s; base < e; base += kWave) {
            int j = base + lane;
            unsigned long long key = ~0ull;
            F3 nrm{0, 0, 0};
            if (j < e) {
                int other = ag.cellItems[j];
                if (other != selfIndex) {
                    sge_agent_state o = ag.all[other];
                    F3 otherDelta = F3{o.velocity[0], o.velocity[1], o.velocity[2]} * segmentDt;
                    float toi;
                    if (capsuleCapsuleSweep(position, remaining, selfRadius, halfHeight,
                                            F3{o.position[0], o.position[1], o.position[2]}, otherDelta, o.radius,
                                            o.halfHeight, toi, nrm)) {
                        // toi >= 0 by construction (bestT in [0,1], moveLen >= 0): float bits order as integers
                        key = ((unsigned long long)__float_as_uint(toi) << 32) | (unsigned)other;
                    }
                }
            }
            unsigned long long k = waveMinU64(key);
            if (k < bestKey) {
                bestKey = k;
                int src = __ffsll((long long)__ballot(key == k)) - 1;
                bestN = F3{__shfl(nrm.x, src, kWave), __shfl(nrm.y, src, kWave), __shfl(nrm.z, src, kWave)};
            }
        }
    }
    if (bestKey == ~0ull) return false;
    toiOut = __uint_as_float((unsigned)(bestKey >> 32));
    normalOut = bestN;
    return true;
}

// PlatformCarry.computeDelta (Systems.swift:644-732) over the step's platform list; every lane computes the same value.
__device__ __forceinline__ F3 platformCarryDelta(F3 position, const sge_controller_params& P, const sge_platform_state* platforms, int count) {
    const float capsuleHalf = P.halfHeight + P.radius;
    const float baseY = position.y - capsuleHalf;
    const F3 capMin{position.x - P.radius, position.y - capsuleHalf, position.z - P.radius};
    const F3 capMax{position.x + P.radius, position.y + capsuleHalf, position.z + P.radius};
    const float sideTol = smax(P.skinWidth, P.groundSnapSkin);
    F3 bestCarry{0, 0, 0}, pushDelta{0, 0, 0};
    for (int k = 0; k < count; ++k) {
        const sge_platform_state pf = platforms[k];
        if (!pf.kinematic) continue;
        const F3 pDelta{pf.delta[0], pf.delta[1], pf.delta[2]};
        if (lengthSq(pDelta) < 1e-8f) continue;
        if (!pf.hasAABB) continue;
        const F3 amin{pf.aabbMin[0], pf.aabbMin[1], pf.aabbMin[2]}, amax{pf.aabbMax[0], pf.aabbMax[1], pf.aabbMax[2]};
        const F3 tol{sideTol, sideTol, sideTol};
        const F3 emin = amin - tol, emax = amax + tol;
        const bool overlap = capMin.x <= emax.x && capMax.x >= emin.x && capMin.y <= emax.y && capMax.y >= emin.y &&
                             capMin.z <= emax.z && capMax.z >= emin.z;
        if (!overlap) continue;
        const bool withinXZ = position.x >= amin.x - P.radius && position.x <= amax.x + P.radius &&
                              position.z >= amin.z - P.radius && position.z <= amax.z + P.radius;
        const float topY = amax.y;
        const float topTol = P.snapDistance + smax(P.skinWidth, P.groundSnapSkin) + 0.05f;
        const bool onTop = withinXZ && baseY >= topY - topTol && baseY <= topY + topTol;
        if (onTop) {
            if (lengthSq(pDelta) > lengthSq(bestCarry)) bestCarry = pDelta;
        } else {
            const float yMin = amin.y - capsuleHalf, yMax = amax.y + capsuleHalf;
            if (position.y >= yMin && position.y <= yMax) {
                const bool outsideX = position.x < amin.x - P.radius || position.x > amax.x + P.radius;
                const bool outsideZ = position.z < amin.z - P.radius || position.z > amax.z + P.radius;
                if (!outsideX && !outsideZ) continue;
                const float cx = smax(amin.x, smin(position.x, amax.x));
                const float cz = smax(amin.z, smin(position.z, amax.z));
                const float dx = position.x - cx, dz = position.z - cz;
                const float sideDistSq = dx * dx + dz * dz;
                const float sidePushTol = P.radius + sideTol;
                if (sideDistSq <= sidePushTol * sidePushTol) {
                    const float dirLen = sqrtf(smax(sideDistSq, 0.0f));
                    if (dirLen > 1e-5f) {
                        const F3 dir{dx / dirLen, 0, dz / dirLen};
                        const float moveToward = dot(F3{pDelta.x, 0, pDelta.z}, dir);
                        if (moveToward > 0) pushDelta = pushDelta + F3{pDelta.x, 0, pDelta.z};
                    }
                }
            }
        }
    }
    if (lengthSq(bestCarry) > 1e-8f) return bestCarry;
    if (lengthSq(pushDelta) > 1e-8f) return pushDelta;
    return F3{0, 0, 0};
}

// ---------------------------------------------------------------------------
// the per-character step
// ---------------------------------------------------------------------------
__device__ __forceinline__ D3 approachVecD(D3 current, D3 target, double maxDelta) { // Systems.swift:419
    D3 delta = target - current;
    double len = length(delta);
    if (len <= maxDelta || len < 0.00001) return target;
    return current + delta / len * maxDelta;
}

struct SlideHit { bool isStatic; CastRec s; float aToi; F3 aNormal; };
// SlideResolver.SlideOptions (Systems.swift:1208-1222): .kinematicMove / .agentSeparation
struct SlideOpts { bool allowHorizontalGroundPass, adjustVelocity, useGroundSnapSkinForStatic, allowTriangleNormalGroundLike; };
__device__ constexpr SlideOpts kKinematicMove{false, true, true, true};
__device__ constexpr SlideOpts kAgentSeparation{true, false, false, false};

// SlideResolver.resolveHit (Systems.swift:1229-1375)
__device__ __forceinline__ bool resolveHit(F3& remaining, float len, const SlideHit& hit, const sge_controller_params& P,
                                           const sge_controller_state& C, bool wasGrounded, bool wasGroundedNear,
                                           D3& velocity, F3& position, bool hasCachedSide, F3 cachedSide,
                                           const SlideOpts options = kKinematicMove) {
    if (options.allowHorizontalGroundPass && hit.isStatic && fabsf(remaining.y) < 1e-5f && hit.s.normal.y >= P.minGroundDot) { // :1240-1247
        position = position + remaining;
        remaining = F3{0, 0, 0};
        return true;
    }
    float contactSkin, hitToi;
    F3 slideNormal, hitTriNormal{0, 0, 0};
    bool hitIsStatic = false, hitIsGroundLike = false;
    if (hit.isStatic) {
        hitToi = hit.s.toi;
        slideNormal = hit.s.normal;
        hitIsGroundLike = hit.s.triNormal.y >= P.minGroundDot;
        contactSkin = (options.useGroundSnapSkinForStatic && hitIsGroundLike) ? P.groundSnapSkin : P.skinWidth;
        hitTriNormal = hit.s.triNormal;
        hitIsStatic = true;
    } else {
        hitToi = hit.aToi;
        slideNormal = hit.aNormal;
        contactSkin = 0;
    }
    if (hitIsStatic && slideNormal.y < P.minGroundDot && C.sideContactFrames > 0) {
        if (hasCachedSide) {
            F3 cachedN = cachedSide;
            if (dot(cachedN, slideNormal) < 0) cachedN = -cachedN;
            slideNormal = cachedN;
        } else {
            F3 cached = ld3(C.sideContactNormal);
            float cachedLen = lengthSq(cached);
            if (cachedLen > 1e-6f) {
                F3 cachedN = cached / sqrtf(cachedLen);
                float dotC = dot(cachedN, slideNormal);
                if (fabsf(dotC) > 0.5f) slideNormal = dotC >= 0 ? cachedN : -cachedN;
            }
        }
    }
    if (slideNormal.y < P.minGroundDot) {
        if (hitIsStatic && hitIsGroundLike && options.allowTriangleNormalGroundLike) slideNormal = hitTriNormal;
        if (slideNormal.y < P.minGroundDot) {
            slideNormal.y = 0;
            float nLen = length(slideNormal);
            if (nLen > 1e-5f) {
                slideNormal = slideNormal / nLen;
            } else {
                position = position + remaining;
                remaining = F3{0, 0, 0};
                return true;
            }
        }
    }
    float into = dot(remaining, slideNormal);
    float intoEps = 1e-4f * len;
    float effectiveSkin;
    if (hitToi <= contactSkin && into < -intoEps) effectiveSkin = smin(contactSkin, hitToi * 0.5f);
    else effectiveSkin = contactSkin;
    float stickyThreshold = contactSkin * 0.1f;
    if (hitToi <= stickyThreshold && into < -intoEps) {
        remaining = remaining - slideNormal * into;
        return false;
    }
    if (into >= -intoEps) {
        if (wasGroundedNear && hitIsStatic && !hitIsGroundLike && remaining.y < 0) remaining.y = 0;
        position = position + remaining;
        remaining = F3{0, 0, 0};
        return true;
    }
    if (hitToi <= effectiveSkin && fabsf(into) <= intoEps) {
        position = position + remaining;
        remaining = F3{0, 0, 0};
        return true;
    }
    if (into >= 0) {
        position = position + remaining;
        remaining = F3{0, 0, 0};
        return true;
    }
    float rawMoveDist = smax(hitToi - effectiveSkin, 0.0f);
    float moveDist = rawMoveDist;
    if (slideNormal.y >= P.minGroundDot && remaining.y < 0 && moveDist > P.groundSweepMaxStep) moveDist = P.groundSweepMaxStep;
    F3 dir = remaining / len;
    position = position + dir * moveDist;
    F3 leftover = remaining - dir * moveDist;
    leftover = leftover - slideNormal * dot(leftover, slideNormal);
    if (wasGrounded && wasGroundedNear && leftover.y < 0) leftover.y = 0;
    float residual = dot(leftover, slideNormal);
    if (fabsf(residual) < 1e-5f) leftover = leftover - slideNormal * residual;
    if (lengthSq(leftover) < 1e-8f) {
        remaining = F3{0, 0, 0};
        return true;
    }
    remaining = leftover;
    if (options.adjustVelocity) {
        D3 snD = toD(slideNormal);
        double vInto = dot(velocity, snD);
        if (vInto < 0) velocity = velocity - snD * vInto;
    }
    return false;
}

// Wave-uniform working set of one character step, kept in LDS between queries (every lane
// reads and writes the same values in lockstep) so that the two big query loops, each
// instantiated exactly once below, do not have to carry it in registers.
struct MoveState {
    F3 position, remaining;
    D3 velocity;
    int phase, it;
    // depenetration
    int didResolve; F3 normalSum; float normalWeight;
    // slide
    float baseMoveLen; int haveLast; F3 lastSlideNormal;
    // ground probe
    CastRec centerHit; int haveCenter; float gDistance; F3 gNormalSum; int sampleK, sampled;
    int nearGround, canSnap, gGrounded, gNear;
    int wasGrounded, wasGroundedNear;
    // constants of the launch + the agent sweep result of this slide iteration
    float dt; F3 gravity; const DevMaterial* materials;
    int aHave; float aToi; F3 aNormal;
    int sideContactCacheOnly; // SGE_STAGE_SIDE_CONTACT_CACHE: SideContactOnlyCachePolicy (Systems.swift:1136-1157)
};
__shared__ MoveState msA[kGroup];

enum { MP_DEPEN = 0, MP_SLIDE = 1, MP_GROUND_CENTER = 2, MP_GROUND_EVAL = 4, MP_GROUND_SAMPLE = 5,
       MP_FINISH = 6, MP_DONE = 7 };

// ---- consume steps of the per-character state machine. They work on the LDS-resident state only and are
// deliberately NOT inlined: the kernel body then holds just the two query loops within its register budget. ----
__device__ __noinline__ void consumeDepen(int g, int nOverlap) { // DepenetrationResolver.resolve :734-808, one iteration
    sge_body_state& body = sBodyA[g]; (void)body;
    const sge_controller_params& P = sParamsA[g]; (void)P;
    sge_controller_state& C = sCtrlA[g]; (void)C;
    MoveState& ms = msA[g];
    const int rb = g * kMaxRays; (void)rb; // first ray slot of this character
    const float dt = ms.dt; (void)dt;
    bool stop = nOverlap == 0;
    if (!stop) {
        const int n = nOverlap;
        // stable sort by depth descending: deepest and second deepest (first occurrence wins ties)
        int i0 = 0;
        for (int k = 1; k < n; ++k) if (sOvl[k].depth > sOvl[i0].depth) i0 = k;
        int i1 = -1;
        for (int k = 0; k < n; ++k) {
            if (k == i0) continue;
            if (i1 < 0 || sOvl[k].depth > sOvl[i1].depth) i1 = k;
        }
        OverlapRec deepest = sOvl[i0];
        OverlapRec second = i1 >= 0 ? sOvl[i1] : deepest;
        __syncthreads();
        const float slop = smax(P.skinWidth * 0.5f, 0.001f);
        bool sideContact = deepest.normal.y < P.minGroundDot;
        int useCount = sideContact ? 1 : (n < 2 ? n : 2);
        float maxDepth = deepest.depth;
        F3 frameNormal{0, 0, 0};
        for (int k = 0; k < useCount; ++k) {
            const OverlapRec& hit = k == 0 ? deepest : second;
            maxDepth = smax(maxDepth, hit.depth);
            F3 nn = hit.normal, cached;
            if (cachedNormal(C, hit.triIndex, cached)) nn = cached;
            frameNormal = frameNormal + nn * hit.depth;
            const bool isSide = hit.normal.y < P.minGroundDot;
            if (isSide || !ms.sideContactCacheOnly) cacheRecord(C, hit.triIndex, nn, isSide); // (:1150 `guard isSideContact else { return }`)
        }
        float frameNormalLen = length(frameNormal);
        F3 depenNormal = frameNormalLen > 1e-6f ? frameNormal / frameNormalLen : frameNormal;
        float push = sideContact ? smax(maxDepth, 0.0f) : smax(maxDepth + slop, 0.0f);
        if (sideContact) push = smin(push, P.skinWidth);
        if (push <= 1e-6f) {
            stop = true;
        } else {
            ms.position = ms.position + depenNormal * push;
            D3 dn = toD(depenNormal);
            D3 velocity = ms.velocity;
            double vInto = dot(velocity, dn);
            if (vInto < 0) velocity = velocity - dn * vInto;
            ms.velocity = velocity;
            ms.didResolve = 1;
            ms.normalSum = ms.normalSum + depenNormal * maxDepth;
            ms.normalWeight += maxDepth;
            ms.it += 1;
            if (ms.it >= 4) stop = true;
        }
    }
    if (stop) {
        if (ms.didResolve) { // applyPreSweepDepenetration :1651-1654
            F3 depenNormal = ms.normalWeight > 1e-6f ? normalize(ms.normalSum / ms.normalWeight) : normalize(ms.normalSum);
            float into = dot(ms.remaining, depenNormal);
            if (into < 0) ms.remaining = ms.remaining - depenNormal * into;
        }
        // resolveKinematicSweep prologue :1671-1673
        F3 baseMove = toF(ms.velocity) * dt;
        ms.baseMoveLen = length(baseMove);
        ms.it = 0;
        ms.phase = MP_SLIDE;
    }
}

__device__ __noinline__ void consumeSlide(int g) { // one iteration of resolveKinematicSweep :1674-1764
    sge_body_state& body = sBodyA[g]; (void)body;
    const sge_controller_params& P = sParamsA[g]; (void)P;
    sge_controller_state& C = sCtrlA[g]; (void)C;
    MoveState& ms = msA[g];
    const int rb = g * kMaxRays; (void)rb; // first ray slot of this character
    const float dt = ms.dt; (void)dt;
    F3 remaining = ms.remaining, position = ms.position;
    D3 velocity = ms.velocity;
    const float len = length(remaining);
    SlideHit hit;
    hit.s = sh.rayRec[rb]; hit.aToi = 0; hit.aNormal = F3{0, 0, 0}; hit.isStatic = true;
    bool haveStatic = rayHit(rb);
    if (haveStatic && hit.s.normal.y < P.minGroundDot && C.sideContactFrames > 0) {
        F3 cached;
        if (cachedNormal(C, hit.s.triIndex, cached)) {
            if (dot(cached, hit.s.normal) < 0) cached = -cached;
            hit.s.normal = cached;
        }
    }
    const bool haveAgent = ms.aHave != 0;
    if (haveAgent) { hit.aToi = ms.aToi; hit.aNormal = ms.aNormal; }
    bool endSlide = false;
    if (haveStatic || haveAgent) {
        if (haveStatic && haveAgent) { // HitSelector.selectBestHit :1382-1390
            float staticSkin = hit.s.normal.y >= P.minGroundDot ? P.groundSnapSkin : P.skinWidth;
            float staticStop = smax(hit.s.toi - staticSkin, 0.0f);
            float agentStop = smax(hit.aToi, 0.0f);
            hit.isStatic = staticStop <= agentStop;
        } else {
            hit.isStatic = haveStatic;
        }
        F3 hitNormal = hit.isStatic ? hit.s.normal : hit.aNormal;
        bool hasCachedSide = false;
        F3 cachedSide{0, 0, 0};
        if (hit.isStatic && hit.s.normal.y < P.minGroundDot && C.sideContactFrames > 0)
            hasCachedSide = cachedNormal(C, hit.s.triIndex, cachedSide);
        bool shouldBreak = resolveHit(remaining, len, hit, P, C, ms.wasGrounded != 0, ms.wasGroundedNear != 0, velocity,
                                      position, hasCachedSide, cachedSide);
        if (hit.isStatic && hit.s.normal.y < P.minGroundDot) cacheRecord(C, hit.s.triIndex, hit.s.normal, true);
        if (ms.haveLast) {
            F3 last = ms.lastSlideNormal;
            float dotN = dot(last, hitNormal);
            if (fabsf(dotN) < 0.98f) {
                F3 axis = cross(last, hitNormal);
                float axisLen = length(axis);
                if (axisLen > 1e-5f) {
                    F3 axisN = axis / axisLen;
                    remaining = axisN * dot(remaining, axisN);
                }
            }
        }
        ms.lastSlideNormal = hitNormal;
        ms.haveLast = 1;
        endSlide = shouldBreak;
    } else {
        position = position + remaining;
        remaining = F3{0, 0, 0};
        endSlide = true;
    }
    ms.remaining = remaining; ms.position = position; ms.velocity = velocity;
    ms.it += 1;
    if (endSlide) ms.phase = MP_GROUND_CENTER;
}

__device__ __noinline__ void consumeGroundCenter(int g) { // GroundProbe.resolve :844-866
    sge_body_state& body = sBodyA[g]; (void)body;
    const sge_controller_params& P = sParamsA[g]; (void)P;
    sge_controller_state& C = sCtrlA[g]; (void)C;
    MoveState& ms = msA[g];
    const int rb = g * kMaxRays; (void)rb; // first ray slot of this character
    const float dt = ms.dt; (void)dt;
    ms.haveCenter = rayHit(rb) ? 1 : 0;
    if (rayHit(rb)) ms.centerHit = sh.rayRec[rb];
    if (rayHit(rb + 1)) ms.gDistance = sh.rayRec[rb + 1].toi;
    ms.phase = MP_GROUND_EVAL;
}

__device__ __noinline__ void consumeGroundSample(int g) { // :906-921, in the reference's sample order
    sge_body_state& body = sBodyA[g]; (void)body;
    const sge_controller_params& P = sParamsA[g]; (void)P;
    sge_controller_state& C = sCtrlA[g]; (void)C;
    MoveState& ms = msA[g];
    const int rb = g * kMaxRays; (void)rb; // first ray slot of this character
    const float dt = ms.dt; (void)dt;
    const CastRec c = ms.centerHit;
    float combineTol = smax(smax(P.groundSnapSkin, P.skinWidth), 0.05f);
    F3 normalSum = ms.gNormalSum;
    const int base = rb + ms.sampleK; // sampleK: 2 when the samples rode along with the centre pass, 0 after a pass of their own
    for (int k = 0; k < 4; ++k) {
        if (rayHit(base + k) && sh.rayRec[base + k].toi <= c.toi + combineTol) {
            if (dot(sh.rayRec[base + k].triNormal, c.triNormal) > 0.98f) normalSum = normalSum + sh.rayRec[base + k].triNormal;
        }
    }
    ms.gNormalSum = normalSum;
    ms.phase = MP_FINISH;
}

__device__ __noinline__ void groundEval(int g) { // :868-894 — decides whether the four offset casts are needed
    sge_body_state& body = sBodyA[g]; (void)body;
    const sge_controller_params& P = sParamsA[g]; (void)P;
    sge_controller_state& C = sCtrlA[g]; (void)C;
    MoveState& ms = msA[g];
    const int rb = g * kMaxRays; (void)rb; // first ray slot of this character
    const float dt = ms.dt; (void)dt;
    const CastRec centerHit = ms.centerHit;
    if (ms.haveCenter && centerHit.toi <= P.snapDistance) {
        const F3 position = ms.position;
        const D3 velocity = ms.velocity;
        float baseCenterY = position.y - P.halfHeight;
        float bottomY = baseCenterY - P.radius;
        float groundTol = smax(P.skinWidth, P.groundSnapSkin);
        bool validGroundPoint = centerHit.position.y <= bottomY + groundTol;
        float groundNearThreshold = smax(P.groundSnapSkin, P.skinWidth);
        bool nearGround = centerHit.toi <= groundNearThreshold;
        bool groundGateVel = velocity.y <= 0;
        double vInto = dot(velocity, toD(centerHit.normal));
        bool groundGateSpeed = vInto >= -(double)P.groundSnapMaxSpeed;
        bool groundGateToi = centerHit.toi <= P.groundSnapMaxToi;
        bool canSnap = validGroundPoint && groundGateVel && (nearGround || groundGateSpeed || groundGateToi);
        if (ms.wasGroundedNear && centerHit.toi <= P.snapDistance) canSnap = validGroundPoint;
        ms.nearGround = nearGround; ms.gNear = nearGround; ms.canSnap = canSnap;
        ms.gDistance = centerHit.toi;
        ms.phase = MP_FINISH;
        if (validGroundPoint && (nearGround || canSnap)) {
            ms.gGrounded = 1;
            ms.gNormalSum = centerHit.triNormal;
            if (centerHit.triNormal.y < 0.98f && (ms.wasGroundedNear || nearGround)) { ms.phase = MP_GROUND_SAMPLE; ms.sampled = 1; }
        }
    } else {
        ms.haveCenter = 0; // guard failed: GroundProbeResult(hit: nil), canSnap false
        ms.phase = MP_FINISH;
    }
    __syncthreads();
}

__device__ __noinline__ void finishStep(int g) { // ground state, GroundSnap, SlopeFriction, writeBack
    sge_body_state& body = sBodyA[g]; (void)body;
    const sge_controller_params& P = sParamsA[g]; (void)P;
    sge_controller_state& C = sCtrlA[g]; (void)C;
    MoveState& ms = msA[g];
    const int rb = g * kMaxRays; (void)rb; // first ray slot of this character
    const float dt = ms.dt; (void)dt;
    const F3 gravity = ms.gravity;
    F3 position = ms.position;
    D3 velocity = ms.velocity;
    const CastRec centerHit = ms.centerHit;
    const bool gGrounded = ms.gGrounded != 0;
    F3 gNormal{0, 1, 0};
    DevMaterial gMat{0.8f, 0.6f, 0};
    int gTri = -1;
    if (gGrounded) { // :890-937
        gMat = ms.materials[centerHit.triIndex];
        gTri = centerHit.triIndex;
        F3 normalSum = ms.gNormalSum;
        float nLen = length(normalSum);
        gNormal = nLen > 1e-6f ? normalSum / nLen : centerHit.triNormal;
        if (ms.wasGroundedNear) {
            const F3 prevNormal = ld3(C.groundNormal);
            float dotN = dot(prevNormal, gNormal);
            if (dotN > 0.9f) {
                const float blend = 0.2f;
                gNormal = normalize(prevNormal * (1 - blend) + gNormal * blend);
            }
        }
        if (gMat.flatten) gNormal = F3{0, 1, 0};
    }
    // GroundSnap.apply :945-963
    if (ms.canSnap && ms.haveCenter) {
        float rawMove = smax(centerHit.toi - P.groundSnapSkin, 0.0f);
        float moveDist = rawMove;
        if (ms.nearGround && moveDist > P.groundSnapMaxStep) moveDist = P.groundSnapMaxStep;
        position = position + F3{0, -1, 0} * moveDist;
        D3 nD = toD(centerHit.normal);
        double vIntoSnap = dot(velocity, nD);
        if (vIntoSnap < 0) velocity = velocity - nD * vIntoSnap;
    }
    if (gGrounded) { // resolveGroundContact :1787-1792
        float normalUpDelta = gNormal.y - C.groundNormal[1];
        if (gTri != C.groundTriangleIndex && normalUpDelta > 0.02f) C.groundTransitionFrames = 3;
    }
    // SlopeFriction.apply :965-1021
    if (!gGrounded) {
        C.flags &= ~(uint32_t)SGE_CTRL_GROUND_SLIDING;
    } else {
        F3 normal = normalize(gNormal);
        if (normal.y > 0.98f) {
            C.groundTransitionFrames = 0;
            C.flags &= ~(uint32_t)SGE_CTRL_GROUND_SLIDING;
        } else if (C.groundTransitionFrames > 0) {
            C.groundTransitionFrames -= 1;
            C.flags &= ~(uint32_t)SGE_CTRL_GROUND_SLIDING;
        } else {
            float gN = dot(gravity, normal);
            F3 gTan = gravity - normal * gN;
            float gTanLen = length(gTan);
            if (gTanLen > 0.5f) {
                float gNMag = fabsf(gN);
                F3 gTanDir = gTan / gTanLen;
                D3 gTanDirD = toD(gTanDir), normalD = toD(normal);
                float stickLimit = gMat.muS * gNMag;
                bool enterSlide = gTanLen > stickLimit * 1.05f;
                bool exitSlide = gTanLen < stickLimit * 0.9f;
                bool sliding = (C.flags & SGE_CTRL_GROUND_SLIDING) != 0;
                if (sliding) { if (exitSlide) sliding = false; }
                else if (enterSlide) sliding = true;
                if (sliding) C.flags |= SGE_CTRL_GROUND_SLIDING; else C.flags &= ~(uint32_t)SGE_CTRL_GROUND_SLIDING;
                if (!sliding && gTanLen <= stickLimit) {
                    D3 vTan = velocity - normalD * dot(velocity, normalD);
                    double downhillSpeed = dot(vTan, gTanDirD);
                    if (downhillSpeed > 0) velocity = velocity - gTanDirD * downhillSpeed;
                } else {
                    float slideAccelMag = smax(gTanLen - gMat.muK * gNMag, 0.0f);
                    if (slideAccelMag > 0) velocity = velocity + gTanDirD * (double)slideAccelMag * (double)dt;
                }
            }
        }
    }
    // writeBack :1802-1821
    D3 pd = toD(position);
    body.position[0] = pd.x; body.position[1] = pd.y; body.position[2] = pd.z;
    C.flags &= ~(uint32_t)(SGE_CTRL_GROUNDED | SGE_CTRL_GROUNDED_NEAR);
    if (gGrounded) C.flags |= SGE_CTRL_GROUNDED;
    if (ms.gNear) C.flags |= SGE_CTRL_GROUNDED_NEAR;
    st3(C.groundNormal, gGrounded ? gNormal : F3{0, 1, 0});
    C.groundDistance = ms.gDistance;
    if (gGrounded) C.groundTriangleIndex = gTri;
    ms.velocity = velocity;
    ms.phase = MP_DONE;
    __syncthreads();
}

static_assert(sizeof(MoveState) <= kMoveScratchBytes, "MoveLaunch::scratch stride");

// The step runs as two launches so that each keeps its own register budget:
//   PART 0  intent, gravity, VelocityGate, contact-cache decay and the pre-sweep depenetration (overlap queries only)
//   PART 1  the slide iterations, the ground probe and the write-back (cast passes only)
// The LDS-resident MoveState crosses the boundary through K.scratch (256 B per character).
// XCD-aware character order (workgroup b runs on XCD b % 8; give each XCD a contiguous eighth of the range so that
// neighbouring characters share an L2): measured -0.6 % on the synthetic terrain, whose BVH fits every L2 anyway, and +3 % on
// the merged real scene, where it also piles the expensive neighbours onto one XCD. Off.
#ifndef SGE_XCD_REMAP
#define SGE_XCD_REMAP 0
#endif
__device__ __forceinline__ int xcdRemap(int b, int n) {
#if SGE_XCD_REMAP
    const int x = b & 7, j = b >> 3, q = n >> 3, r = n & 7;
    return x * q + (x < r ? x : r) + j;
#else
    return b;
#endif
}

// PART 0 of one character's step on LDS slot g (body / params / controller already there): intent, gravity, VelocityGate, contact-cache
// decay, platform carry and the pre-sweep depenetration. Leaves the step's working set in msA[g] (phase MP_SLIDE, or MP_DONE when the
// body does not move). Shared by move_kernel<0> (slot 0; the result crosses to part 1 through K.scratch) and by move_group_kernel,
// which runs it for its own members at its head: the grouped launch then needs no launch in front of it.
__device__ __forceinline__ void movePart0(const MoveLaunch& K, const int g, const int e, WaveStats& st, long long& pQuery) {
    const DevCollision& col = K.col;
    sge_body_state& body = sBodyA[g];
    const sge_controller_params& P = sParamsA[g];
    sge_controller_state& C = sCtrlA[g];
    MoveState& ms = msA[g];
    const float dt = K.dt;
    const F3 gravity{K.gx, K.gy, K.gz};
    {
        D3 velocity{body.linearVelocity[0], body.linearVelocity[1], body.linearVelocity[2]};
        // ---- PhysicsIntentSystem, controller branch (Systems.swift:217-247) ----
        if (K.stages & SGE_STAGE_INTENT) {
            const sge_move_intent in = K.crowd.intents[e];
            if ((in.flags & SGE_INTENT_PRESENT) && (body.bodyType == SGE_BODY_DYNAMIC || body.bodyType == SGE_BODY_KINEMATIC)) {
                if (in.flags & SGE_INTENT_DODGE_ACTIVE) {
                    velocity.x = (double)in.desiredVelocity[0];
                    velocity.z = (double)in.desiredVelocity[2];
                } else {
                    D3 target{(double)in.desiredVelocity[0], 0.0, (double)in.desiredVelocity[2]};
                    D3 current{velocity.x, 0.0, velocity.z};
                    float accel = length(target) >= length(current) ? in.maxAcceleration : in.maxDeceleration;
                    D3 next = approachVecD(current, target, (double)accel * (double)dt);
                    velocity.x = next.x;
                    velocity.z = next.z;
                }
                if (in.flags & SGE_INTENT_HAS_FACING_YAW) {
                    Quat q = quatAngleAxis(in.desiredFacingYaw, F3{0, 1, 0});
                    body.rotation[0] = q.x; body.rotation[1] = q.y; body.rotation[2] = q.z; body.rotation[3] = q.w;
                }
            }
        }
        // ---- GravitySystem (Systems.swift:609-618) ----
        if ((K.stages & SGE_STAGE_GRAVITY) && body.bodyType == SGE_BODY_DYNAMIC &&
            !((C.flags & SGE_CTRL_GROUNDED) && (C.flags & SGE_CTRL_GROUNDED_NEAR))) {
            velocity = velocity + toD(gravity) * (double)dt;
        }
        ms.velocity = velocity;
    }
    const bool doMove = (K.stages & SGE_STAGE_MOVE) && body.bodyType != SGE_BODY_STATIC;
    ms.position = toF(D3{body.position[0], body.position[1], body.position[2]});
    ms.phase = MP_DONE;
    ms.dt = dt; ms.gravity = gravity; ms.materials = col.materials; ms.aHave = 0; ms.aToi = 0; ms.aNormal = F3{0, 0, 0};
    ms.sideContactCacheOnly = (K.stages & SGE_STAGE_SIDE_CONTACT_CACHE) ? 1 : 0;
    if (doMove) {
        cacheDecay(C);
        if (K.platformCount > 0) { // applyPlatformDelta :1619-1633
            F3 platformDelta = platformCarryDelta(ms.position, P, K.platforms, K.platformCount);
            if (lengthSq(platformDelta) > 1e-8f) ms.position = ms.position + platformDelta;
        }
        ms.wasGrounded = (C.flags & SGE_CTRL_GROUNDED) != 0;
        ms.wasGroundedNear = (C.flags & SGE_CTRL_GROUNDED_NEAR) != 0;
        // VelocityGate.apply :1037-1051
        D3 velocity = ms.velocity;
        const bool wg = ms.wasGrounded && ms.wasGroundedNear;
        if (wg && velocity.y < 0) velocity.y = 0;
        D3 remD = velocity * (double)dt;
        if (wg && remD.y < 0) remD.y = 0;
        ms.velocity = velocity;
        ms.remaining = toF(remD);
        ms.phase = MP_DEPEN;
        ms.it = 0;
        ms.didResolve = 0; ms.normalSum = F3{0, 0, 0}; ms.normalWeight = 0;
        ms.haveLast = 0; ms.lastSlideNormal = F3{0, 0, 0}; ms.baseMoveLen = 0;
        ms.haveCenter = 0; ms.gDistance = kFloatMax; ms.gNormalSum = F3{0, 0, 0}; ms.sampleK = 0; ms.sampled = 0;
        ms.nearGround = 0; ms.canSnap = 0; ms.gGrounded = 0; ms.gNear = 0;
    }
    __syncthreads();
    // applyPreSweepDepenetration :1635-1656 — one overlap query per trip
    while (ms.phase == MP_DEPEN) {
        __syncthreads();
        const long long pq = K.waveProf ? (long long)__builtin_amdgcn_s_memtime() : 0;
        const int nOverlap = waveCapsuleOverlapAll(col, ms.position, P.radius, P.halfHeight, 8, P.collisionMask, st);
        if (K.waveProf) pQuery += (long long)__builtin_amdgcn_s_memtime() - pq;
        consumeDepen(g, nOverlap);
        __syncthreads();
    }
}

template <int PART, bool AGENTS, bool HEAVY = false>
#ifndef SGE_MOVE_WAVES1
#define SGE_MOVE_WAVES1 4
#endif
// (the multi-wave form: three waves per SIMD = at most 168 VGPRs, so that its 2 waves per SIMD fit into the 336 registers two resident
// LBS wavefronts leave of a SIMD's 512; at the 176 the compiler takes when asked for less, the workgroup found no CU beside a resident
// LBS launch and ran only after it: DESIGN.md 3.5)
#ifndef SGE_HEAVY_WAVES_EU
#define SGE_HEAVY_WAVES_EU 3
#endif
__global__ __launch_bounds__(HEAVY ? kWave * kHeavyWaves : kWave, PART == 0 ? (SGE_CCD_EXCLUSIVE ? 2 : 4) : (HEAVY ? SGE_HEAVY_WAVES_EU : SGE_MOVE_WAVES1)) void move_kernel(MoveLaunch K) {
    if (PART == 0) SGE_PAD_VGPRS();
    if (HEAVY) SGE_HEAVY_PRIO();
    // PART 1 may run over an index list (light / heavy characters of this step, see classify_kernel)
    // (so may PART 0: the multi-wave launch's characters take their part 0 in a launch of their own, see launch_move)
    if (K.list && (int)blockIdx.x >= *K.listCount) return;
    const int e = K.list ? K.list[blockIdx.x] : K.first + xcdRemap((int)blockIdx.x, K.count);
    if (PART == 1 && !K.list && K.heavyFlags && K.heavyFlags[e]) return; // this character runs in the multi-wave launch
    const int lane = laneId();
    WaveStats st{0, 0, 0, 0, 0, 0, 0};
    const DevCollision& col = K.col;
    const long long pT0 = (PART == 0 && K.waveProf) ? (long long)__builtin_amdgcn_s_memtime() : 0; // diagnostics (SGE_WAVE_PROF)
    long long pQuery = 0;
    if (HEAVY) {
        if (threadIdx.x == 0) { hv.cmd = HCMD_NONE; hv.evalSum = 0; }
        __syncthreads();
        if (threadIdx.x >= kWave) { // helper waves
            heavyHelperLoop(col, st);
            if (K.stats && lane == 0 && st.trips) atomicAdd(&statShard(K.stats)[5], (unsigned long long)st.trips);
            return;
        }
    }
#ifdef SGE_CCD_TIMING
    const long long tStart = (long long)__builtin_amdgcn_s_memtime();
#endif
    {   // 288 B of state: lanes copy dwords
        const uint32_t* gb = reinterpret_cast<const uint32_t*>(K.crowd.bodies + e);
        const uint32_t* gp = reinterpret_cast<const uint32_t*>(K.crowd.params + e);
        const uint32_t* gc = reinterpret_cast<const uint32_t*>(K.crowd.controllers + e);
        if (lane < 24) reinterpret_cast<uint32_t*>(&sBodyA[0])[lane] = gb[lane];
        if (lane < 16) reinterpret_cast<uint32_t*>(&sParamsA[0])[lane] = gp[lane];
        if (lane < 32) reinterpret_cast<uint32_t*>(&sCtrlA[0])[lane] = gc[lane];
        __syncthreads();
    }
    sge_body_state& body = sBodyA[0];
    const sge_controller_params& P = sParamsA[0];
    sge_controller_state& C = sCtrlA[0]; (void)C;
    MoveState& ms = msA[0];
    const float dt = K.dt;
    const F3 gravity{K.gx, K.gy, K.gz}; (void)gravity;
    uint32_t* const scratch = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(K.scratch) + (size_t)e * kMoveScratchBytes);
    if (PART == 1) {
        if (lane < (int)(sizeof(MoveState) / 4)) reinterpret_cast<uint32_t*>(&ms)[lane] = scratch[lane];
        __syncthreads();
    }
    if (PART == 0) movePart0(K, 0, e, st, pQuery);
    const bool hasAgent = (P.agentFlags & SGE_AGENT_PRESENT) != 0;
    const bool selfSolid = hasAgent && (P.agentFlags & SGE_AGENT_SOLID);
    const bool useAgents = AGENTS && (K.stages & SGE_STAGE_AGENTS) && selfSolid && K.agents.all != nullptr;
    __syncthreads();

    // Each trip issues at most one BVH query; each query routine is inlined exactly once (one per PART).
    while (PART == 1 && ms.phase != MP_DONE) {
        const int phase = ms.phase;
        // ---------------- 1. which query does this phase need? ----------------
        bool doCast = false, blocking = false;
        if (phase == MP_SLIDE) {
            // head of the slide loop :1674-1676
            float len = length(ms.remaining);
            if (ms.it >= P.maxSlideIterations || len < 1e-6f) { ms.phase = MP_GROUND_CENTER; __syncthreads(); continue; }
            doCast = true; blocking = true;
            sh.rayCount = 1; sh.rayFrom[0] = ms.position; sh.rayDelta[0] = ms.remaining;
        } else if (phase == MP_GROUND_CENTER) {
            // GroundProbe's snap cast (:844-853) and fall probe (:855-866) share origin, capsule and filters:
            // two rays of one pass. A disabled probe becomes a zero-length ray (= nil, like :987-988).
            doCast = true;
            sh.rayFrom[0] = ms.position; sh.rayFrom[1] = ms.position;
            sh.rayDelta[0] = P.snapDistance > 0 ? F3{0, -1, 0} * P.snapDistance : F3{0, 0, 0};
            sh.rayDelta[1] = P.fallProbeDistance > 0 ? F3{0, -1, 0} * P.fallProbeDistance : F3{0, 0, 0};
            // The four offset casts (:898-921) are issued only when the centre hit turns out to be a slope under a character
            // near the ground (groundEval). A character that stood on a slope last step will almost surely need them again,
            // so they ride along as rays 2..5 of this pass (casts are pure: unused results are simply dropped); everyone
            // else casts them in a pass of their own if the condition comes true.
            const bool spec = __builtin_amdgcn_readfirstlane((int)(ms.wasGroundedNear && P.snapDistance > 0 && K.hint && K.hint[e] != 0)) != 0;
            sh.rayCount = spec ? 6 : 2;
            ms.sampleK = spec ? 2 : 0;
            if (spec) {
                const float offset = P.radius * 0.6f;
                const F3 snapDelta = F3{0, -1, 0} * P.snapDistance;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float ox = k == 0 ? offset : (k == 1 ? -offset : 0.0f);
                    float oz = k == 2 ? offset : (k == 3 ? -offset : 0.0f);
                    sh.rayFrom[2 + k] = ms.position + F3{ox, 0, oz};
                    sh.rayDelta[2 + k] = snapDelta;
                }
            }
        } else if (phase == MP_GROUND_SAMPLE && ms.sampleK == 0) {
            // the four offset casts of :898-921, one pass
            doCast = true;
            sh.rayCount = 4;
            const float offset = P.radius * 0.6f;
            const F3 snapDelta = F3{0, -1, 0} * P.snapDistance;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float ox = k == 0 ? offset : (k == 1 ? -offset : 0.0f);
                float oz = k == 2 ? offset : (k == 3 ? -offset : 0.0f);
                sh.rayFrom[k] = ms.position + F3{ox, 0, oz};
                sh.rayDelta[k] = snapDelta;
            }
        }
        __syncthreads();
        // ---------------- 2. the query ----------------
        if (PART == 1 && doCast) waveCastRays<HEAVY>(col, P.radius, P.halfHeight, blocking, !blocking, P.minGroundDot, P.collisionMask, st,
                                                     HEAVY && phase == MP_GROUND_CENTER ? 1 : -1);
        // ---------------- 3. consume ----------------
        if (phase == MP_SLIDE) {
            if (AGENTS && useAgents) { // AgentSweepSolver.bestHit :1053-1091 (independent of the static hit)
                const F3 remaining = ms.remaining;
                const float selfRadius = (hasAgent && (P.agentFlags & SGE_AGENT_RADIUS_OVERRIDE)) ? P.agentRadiusOverride : P.radius;
                float aToi = 0; F3 aNormal{0, 0, 0};
                bool have = waveAgentBestHit(K.agents, ms.position, remaining, length(remaining), ms.baseMoveLen, dt, K.agents.selfOffset + e,
                                             selfRadius, P.halfHeight, aToi, aNormal);
                ms.aHave = have ? 1 : 0; ms.aToi = aToi; ms.aNormal = aNormal;
            }
            consumeSlide(0);
        } else if (phase == MP_GROUND_CENTER) consumeGroundCenter(0);
        else if (phase == MP_GROUND_SAMPLE) consumeGroundSample(0);
        __syncthreads();
        if (PART == 1 && ms.phase == MP_GROUND_EVAL) { groundEval(0); __syncthreads(); }
        if (PART == 1 && ms.phase == MP_FINISH) { finishStep(0); __syncthreads(); }
    }

    if (ms.phase == MP_DONE) { // the step (or, without SGE_STAGE_MOVE, the velocity update) is complete
        D3 velocity = ms.velocity;
        body.linearVelocity[0] = velocity.x; body.linearVelocity[1] = velocity.y; body.linearVelocity[2] = velocity.z;
    }
    __syncthreads();
    if (PART == 0 && lane < (int)(sizeof(MoveState) / 4)) scratch[lane] = reinterpret_cast<const uint32_t*>(&ms)[lane];
    storeCharacter(K.crowd, e, 0, lane);
#ifdef SGE_CCD_TIMING
    if (lane == 0) {
        atomicAdd(&g_cycTotal, (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - tStart));
    }
#endif
    if (PART == 0 && K.waveProf && lane == 0) { // region 1 of the diagnostics buffer: one row per character
        unsigned long long* w = K.waveProf + ((size_t)K.crowd.count + (size_t)e) * 8;
        w[0] = (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - pT0); w[1] = (unsigned long long)pQuery;
        w[2] = st.steps; w[3] = st.trips; w[7] = (unsigned long long)pT0;
    }
    if (HEAVY) { // release the helper waves
        if (lane == 0) hv.cmd = HCMD_EXIT;
        __syncthreads();
    }
    {
        // evals are counted per lane; sum over the wave
        unsigned v = st.evals;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
        // this step's sweep cost decides which kernel takes the character next step
        if (PART == 1 && K.cost && lane == 0) K.cost[e] = HEAVY ? (int)hv.evalSum : (int)v;
        // did this step need the four offset ground casts? (next step's centre pass then carries them along)
        if (PART == 1 && K.hint && lane == 0) K.hint[e] = (uint8_t)(ms.sampled != 0);
    }
    if (K.stats) {
        // evals are counted per lane; sum over the wave
        unsigned v = st.evals, pr = st.pruned; // per-lane counts
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o, kWave); pr += __shfl_xor(pr, o, kWave); }
        (void)pr;
        if (lane == 0) {
            unsigned long long* sp = statShard(K.stats);
            if (st.queries) atomicAdd(&sp[0], (unsigned long long)st.queries);
            if (st.candidates) atomicAdd(&sp[1], (unsigned long long)st.candidates);
            if (st.overflow) atomicAdd(&sp[3], (unsigned long long)st.overflow);
#ifdef SGE_CCD_TIMING
            // diagnostic build: the three counters below carry cycle sums instead (shard 0 only)
            K.stats[2] = g_cycTotal; K.stats[4] = g_cycTraverse; K.stats[5] = g_cycSweep;
#else
            if (v) atomicAdd(&sp[2], (unsigned long long)v);
            if (st.steps) atomicAdd(&sp[4], (unsigned long long)st.steps);
            if (st.trips) atomicAdd(&sp[5], (unsigned long long)st.trips);
            if (pr) atomicAdd(&sp[6], (unsigned long long)pr);
#endif
        }
    }
}

// ---------------------------------------------------------------------------
// move_group_kernel: part 1 of the step for kGroup characters per wavefront
// ---------------------------------------------------------------------------
// With one character per wavefront the sweep trips — two thirds of the kernel's vector instructions — run with ~14 of 64
// lanes active: a cast pass meets ~10 (ray, triangle) pairs once the exact rejects have run (rocprofv3: SQ_THREAD_CYCLES_VALU /
// (64 x SQ_INSTS_VALU) = 0.30, profiles/r2_move_pmc_*.json). Here a wavefront steps kGroup characters together: per round every
// unfinished character sets up its next cast pass (whatever its phase: slide, ground centre, ground samples) and traverses for
// its work items, all items go into ONE queue tagged with their ray slot, the 64 lanes sweep that queue together, then every
// character consumes its rays' results. Each character still performs exactly its own sequence of queries and arithmetic
// (ray slots [g * kMaxRays, +kMaxRays) and the LDS state arrays carry its place g), so results stay bit-identical.
__shared__ int sGroupCost[kGroup]; // distance evaluations of each character's casts this step
__shared__ int sGroupE[kGroup];    // character index of every member (-1: none)
// Near pass / far pass of the ground-centre cast: the 200-unit fall probe (Systems.swift:855-866) shares its origin with the 0.8-unit
// snap cast, so the triangles around the capsule are gathered and swept first (box of the snap ray and the offset rays; the fall
// probe gets its items from the same candidates), and only then is the rest of the fall probe's box traversed — clipped to the
// best hit found so far. A grounded character's clipped box lies inside the near box and the far pass disappears, with it the walk
// through every wall triangle below a ledge. Exact: a triangle entirely below the capsule's lowest point at t = bestToi (+ margin)
// cannot be touched before bestToi, and a later or equal hit with a higher visit rank never wins.
__shared__ F3 sNearMin[kGroup], sNearMax[kGroup];
// speculative bisection (groupSweep): lane of the refining item a helper works for, and the helpers' answers per item
#ifndef SGE_BISECT_SPEC
#define SGE_BISECT_SPEC 1
#endif
// (at most 31 items are served per trip: every served item has at least two of the <= 63 idle lanes)
constexpr int kSpecSlots = kWave / 2;
__shared__ unsigned char sSpecOwner[kSpecSlots];
__shared__ unsigned sSpecBits[kSpecSlots];
// The bounds these LDS arrays rest on, checked where they are declared (an out-of-allocation LDS access aborts the queue; one working-tree
// build of round 2 did): a served item's rank is < nServe <= nHelp / per with per >= 2 and nHelp <= kWave - 1 idle lanes
// (the refining owner is not idle); a helper's node number is <= 14 + 1, a bit of one 32-bit word; a lane index fits the owner byte.
static_assert((kWave - 1) / 2 <= kSpecSlots, "speculative bisection: ranks of served items");
static_assert(14 + 2 <= 32, "speculative bisection: one bit per tree node in sSpecBits");
static_assert(kWave <= 256, "speculative bisection: owner lane stored in a byte");
// a traversal expands only while fewer than kWave candidates are pending and one expansion adds at most kWave: kWave - 1 + kWave;
// the item queue is swept before a batch of kWave candidates x kMaxRays rays might not fit
static_assert(kCandCap >= 2 * kWave - 1 && kRangeCap >= 2 * kWave - 1, "candidate / range lists");
static_assert(kItemCap >= kWave + kWave * kMaxRays, "item queue: one gathered batch always fits an empty queue");
static_assert(kRaySlots <= (1 << (32 - kItemRayShift)), "work item word: the ray slot has 32 - kItemRayShift bits");

// One cast pass of character g, rays rb .. rb + R - 1 already in sh.rayFrom / sh.rayDelta: the per-ray setup of waveCastRays
// (CollisionQuery.swift:1021-1035). Returns the number of valid rays and their union box.
__device__ __forceinline__ int groupSetupRays(const DevCollision& col, int rb, int R, float radius, float halfHeight, bool blockingOnly,
                                              bool hasMinNormalY, float minNormalY, F3& minP, F3& maxP) {
    const int lane = laneId();
    if (lane < R) {
        const int r = rb + lane;
        F3 from = sh.rayFrom[r], delta = sh.rayDelta[r];
        float len = length(delta);
        bool valid = !(len < 1e-6f) && col.root >= 0; // :987-988, :1020
        F3 dir = delta / len;
        F3 up{0, 1, 0};
        F3 a0 = from + up * halfHeight, b0 = from - up * halfHeight;
        F3 a1 = a0 + delta, b1 = b0 + delta;
        F3 mn = vmin(vmin(a0, b0), vmin(a1, b1));
        F3 mx = vmax(vmax(a0, b0), vmax(a1, b1));
        F3 ext{radius, radius, radius};
        sh.rayMin[r] = mn - ext; sh.rayMax[r] = mx + ext;
        sh.rayDir[r] = dir; sh.rayLen[r] = len; sh.rayValid[r] = valid ? 1 : 0;
        sh.rayVertical[r] = (delta.x == 0.0f && delta.z == 0.0f) ? 1 : 0;
        const float minAdv = smax(radius * 0.02f, 1e-4f);
        int maxIter = (int)ceilf(len / minAdv) + 1; // :1296
        sh.rayMaxIter[r] = maxIter < 256 ? maxIter : 256;
        sh.rayKey[r] = ((unsigned long long)__float_as_uint(len) << 32) | 0xffffffffull;
        sh.rayRadius[r] = radius; sh.rayHalfHeight[r] = halfHeight; sh.rayMinNormalY[r] = minNormalY;
        sh.rayFilter[r] = (blockingOnly ? 1 : 0) | (hasMinNormalY ? 2 : 0);
    }
    __syncthreads();
    minP = F3{kFloatMax, kFloatMax, kFloatMax}; maxP = F3{-kFloatMax, -kFloatMax, -kFloatMax};
    int nValid = 0;
    for (int r = rb; r < rb + R; ++r) {
        if (!sh.rayValid[r]) continue;
        minP = vmin(minP, sh.rayMin[r]); maxP = vmax(maxP, sh.rayMax[r]);
        nValid += 1;
    }
    return nValid;
}

// Sweeps every queued (ray slot, triangle) item: the streaming lane state machine of waveCastRays with the capsule and the
// acceptance filters taken from the item's ray slot. Leaves the queue empty.
__device__ __forceinline__ void groupSweep(const DevCollision& col, int& itemCount, WaveStats& st) {
    const int lane = laneId();
    const float contactEps = 1e-5f;
    int phase = PH_DONE, myRay = 0, iter = 0, refineK = 0, maxIter = 0;
    Tri tri;
    tri.v0 = tri.v1 = tri.v2 = F3{0, 0, 0}; tri.triIndex = -1; tri.rank = 0x7fffffff;
    F3 triNormal{0, 0, 0};
    float t = 0, lastSafeT = 0, lo = 0, hi = 0, tEval = 0, len = 0, bestToi = 0, radius = 0, halfHeight = 0, minAdvance = 0;
    int crawlRun = 0;    // consecutive march steps that advanced by exactly minAdvance
    unsigned evals = 0;  // of this lane's current item
    while (true) {
        const unsigned long long idleMask = __ballot(phase == PH_DONE);
        int nIdle = __popcll(idleMask);
        unsigned long long takenMask = 0;
        if (nIdle > 0 && itemCount > 0) {
            const int take = nIdle < itemCount ? nIdle : itemCount;
            const int p = prefixCount(idleMask);
            if (phase == PH_DONE && p < take) {
                if (evals) { atomicAdd(&sGroupCost[myRay / kMaxRays], (int)evals); st.evals += evals; evals = 0; }
                const int it = sh.items[itemCount - 1 - p];
                myRay = (unsigned)it >> kItemRayShift;
                tri = loadTri(col, it & kItemSlotMask);
                triNormal = normalize(cross(tri.v1 - tri.v0, tri.v2 - tri.v0));
                len = sh.rayLen[myRay];
                maxIter = sh.rayMaxIter[myRay];
                radius = sh.rayRadius[myRay]; halfHeight = sh.rayHalfHeight[myRay];
                minAdvance = smax(radius * 0.02f, 1e-4f); // :1295
                bestToi = __uint_as_float((unsigned)(sh.rayKey[myRay] >> 32));
                phase = PH_MARCH;
                t = 0; lastSafeT = 0; lo = 0; hi = 0; tEval = 0; iter = 0; refineK = 0; crawlRun = 0;
            }
            takenMask = __ballot(phase == PH_MARCH) & idleMask;
            itemCount -= take;
            nIdle -= take;
            __syncthreads();
        } else if (nIdle == kWave) {
            break;
        }
        st.trips += 1;
        if (phase == PH_MARCH) {
            // loop head of :1303-1307 — iteration budget, then `if t > maxDistance return nil`
            if (iter >= maxIter || t > len || lastSafeT > bestToi) phase = PH_DONE;
            else { iter += 1; tEval = t; }
        } else if (phase == PH_REFINE) {
            if (lo > bestToi) phase = PH_DONE;
            else tEval = 0.5f * (lo + hi);
        }
        // ---- speculative crawl ------------------------------------------------------------------------------------------
        // A march that creeps along a surface advances by exactly minAdvance per evaluation (advance = max(dist - r, minAdvance),
        // :1316-1321), up to the reference's 256-iteration cap: hundreds of dependent evaluations on one lane while the wavefront
        // idles — the tail of the whole launch. While advance == minAdvance the evaluation points do not depend on the distances:
        // t_{k+1} = t_k + minAdvance. When at least half the lanes are idle and nothing is queued, the idle lanes evaluate the
        // creeping item ("owner") at t_{k+1}, t_{k+2}, ... (the same sequential float additions the march makes) in the trip in
        // which the owner evaluates t_k, and the owner then takes over the prefix up to the first evaluation that ends the loop,
        // makes contact, or advances by more than minAdvance — exactly the states the sequential march goes through.
        const unsigned long long crawlMask = __ballot(phase == PH_MARCH && crawlRun >= 3);
        const unsigned long long helpMask = idleMask & __ballot(phase == PH_DONE) & ~takenMask;
        const int nHelp = __popcll(helpMask);
        const bool spec = itemCount == 0 && crawlMask != 0 && nHelp >= kWave / 2;
        const int owner = spec ? __ffsll((long long)crawlMask) - 1 : 0;
        const bool helper = spec && ((helpMask >> lane) & 1);
        const int h = prefixCount(helpMask); // rank among the helpers: helper h evaluates t_{k+1+h}
        // what this lane evaluates in this trip
        bool cEval = phase != PH_DONE;
        float cT = tEval, cPrev = 0, cHalf = halfHeight, bMinAdv = 0, bRadius = 0;
        F3 cFrom = sh.rayFrom[myRay], cDir = sh.rayDir[myRay], c0 = tri.v0, c1 = tri.v1, c2 = tri.v2;
        if (spec) {
            const int bRay = __shfl(myRay, owner, kWave), bIter = __shfl(iter, owner, kWave), bMaxIter = __shfl(maxIter, owner, kWave);
            const float bT = __shfl(t, owner, kWave), bLen = __shfl(len, owner, kWave), bHalf = __shfl(halfHeight, owner, kWave);
            bMinAdv = __shfl(minAdvance, owner, kWave); bRadius = __shfl(radius, owner, kWave);
            const F3 b0{__shfl(tri.v0.x, owner, kWave), __shfl(tri.v0.y, owner, kWave), __shfl(tri.v0.z, owner, kWave)};
            const F3 b1{__shfl(tri.v1.x, owner, kWave), __shfl(tri.v1.y, owner, kWave), __shfl(tri.v1.z, owner, kWave)};
            const F3 b2{__shfl(tri.v2.x, owner, kWave), __shfl(tri.v2.y, owner, kWave), __shfl(tri.v2.z, owner, kWave)};
            if (helper) {
                cHalf = bHalf; c0 = b0; c1 = b1; c2 = b2;
                cFrom = sh.rayFrom[bRay]; cDir = sh.rayDir[bRay];
                cT = bT;
            }
            for (int k = 0; k < nHelp; ++k) if (helper && k <= h) { cPrev = cT; cT += bMinAdv; } // sequential additions, as the march makes them
            // loop head of :1303-1307 for the helper's evaluation: the owner's count already includes its own evaluation of this trip
            if (helper) cEval = (bIter + h < bMaxIter) && !(cT > bLen);
        }
        // ---- speculative bisection -------------------------------------------------------------------------------------
        // refineTOI (:1361-1377) is ten DEPENDENT evaluations, and every contact goes through it — the longest chain of an ordinary
        // cast pass (two march steps, ten bisection steps, the final evaluation). The points of the next steps are functions of the
        // interval only: mid = 0.5 (lo + hi), then 0.5 (lo + mid) or 0.5 (mid + hi), ... Idle lanes evaluate the two (six, fourteen)
        // descendants of a refining item's midpoint in the trip in which its owner evaluates the midpoint itself — the same float
        // operations on the same operands — and the owner then walks two (three, four) steps of the bisection at once.
        // Node numbering: root 1 = the owner's midpoint; 2n = the midpoint after "contact at n" (hi = mid), 2n + 1 after "clear" (lo = mid).
        const unsigned long long refMask = __ballot(phase == PH_REFINE);
        const int nOwn = __popcll(refMask);
        int per = 0, levels = 1;
        if (SGE_BISECT_SPEC && !spec && nOwn > 0 && nHelp >= 2) {
            per = nOwn * 14 <= nHelp ? 14 : (nOwn * 6 <= nHelp ? 6 : 2);
            levels = per == 14 ? 4 : (per == 6 ? 3 : 2);
        }
        const bool bspec = per != 0;
        const int nServe = bspec ? (nOwn < nHelp / per ? nOwn : nHelp / per) : 0;
        const int oRank = prefixCount(refMask);
        const bool served = bspec && phase == PH_REFINE && oRank < nServe;
        const bool bHelper = bspec && ((helpMask >> lane) & 1) && h < nServe * per;
        int node = 0, bOwnerRank = 0;
        float bRad = 0;
        bool ownContact = false;
        if (bspec) {
            if (served) { sSpecOwner[oRank] = (unsigned char)lane; sSpecBits[oRank] = 0; }
            __syncthreads();
            bOwnerRank = bHelper ? h / per : 0;
            node = bHelper ? h - bOwnerRank * per + 2 : 0;
            const int src = bHelper ? (int)sSpecOwner[bOwnerRank] : lane;
            const int oRay = __shfl(myRay, src, kWave);
            float oLo = __shfl(lo, src, kWave), oHi = __shfl(hi, src, kWave);
            const float oHalf = __shfl(halfHeight, src, kWave);
            bRad = __shfl(radius, src, kWave);
            const F3 o0{__shfl(tri.v0.x, src, kWave), __shfl(tri.v0.y, src, kWave), __shfl(tri.v0.z, src, kWave)};
            const F3 o1{__shfl(tri.v1.x, src, kWave), __shfl(tri.v1.y, src, kWave), __shfl(tri.v1.z, src, kWave)};
            const F3 o2{__shfl(tri.v2.x, src, kWave), __shfl(tri.v2.y, src, kWave), __shfl(tri.v2.z, src, kWave)};
            if (bHelper) {
                // walk from the root to this node: the bits of `node` below its leading one, most significant first
                const int depth = 31 - __clz(node); // 1 .. 3
#pragma unroll
                for (int i = 2; i >= 0; --i) {
                    if (i < depth) {
                        const float mid = 0.5f * (oLo + oHi);
                        if ((node >> i) & 1) oLo = mid; else oHi = mid;
                    }
                }
                cT = 0.5f * (oLo + oHi);
                cHalf = oHalf; c0 = o0; c1 = o1; c2 = o2;
                cFrom = sh.rayFrom[oRay]; cDir = sh.rayDir[oRay];
                cEval = true;
            }
        }
        bool finished = false;
        unsigned long long myKey = ~0ull;
        CastRec rec;
        rec.toi = 0; rec.position = rec.normal = rec.triNormal = F3{0, 0, 0}; rec.triIndex = -1;
        float dist = 0;
        F3 segP{0, 0, 0}, triP{0, 0, 0};
        if (cEval) dist = segmentTriangleDistance(cFrom + cDir * cT, cHalf, c0, c1, c2, segP, triP);
        if (phase != PH_DONE) {
            evals += 1;
            const F3 dir = cDir;
            if (phase == PH_MARCH) {
                if (dist <= radius + contactEps) {
                    // refineTOI(t0: lastSafeT, t1: t) :1361-1377
                    float k0 = smax(0.0f, smin(lastSafeT, len));
                    float k1 = smax(0.0f, smin(t, len));
                    lo = smin(k0, k1);
                    hi = smax(k0, k1);
                    if (hi - lo < 1e-5f) { phase = PH_FINAL; tEval = hi; }
                    else { phase = PH_REFINE; refineK = 0; }
                    crawlRun = 0;
                } else {
                    lastSafeT = t;
                    float advance = smax(dist - radius, minAdvance);
                    crawlRun = advance == minAdvance ? crawlRun + 1 : 0;
                    if (advance <= 0) t += minAdvance; else t += advance;
                }
            } else if (phase == PH_REFINE) {
                ownContact = dist <= radius;
                if (ownContact) hi = tEval; else lo = tEval;
                refineK += 1;
                if (refineK == 10) { phase = PH_FINAL; tEval = hi; }
            } else { // PH_FINAL :1325-1346
                float tHit = tEval;
                F3 nrm;
                if (dist < 1e-6f) nrm = dot(triNormal, dir) > 0 ? -triNormal : triNormal;
                else nrm = normalize(segP - triP);
                F3 triN = triNormal;
                if (dot(triN, nrm) < 0) triN = -triN;
                phase = PH_DONE;
                const int filter = sh.rayFilter[myRay];
                bool ok = tHit < len;
                if (ok && (filter & 1)) {
                    F3 delta = sh.rayDelta[myRay];
                    ok = !(dot(delta, nrm) >= 0) && !(dot(delta, triN) >= 0);
                }
                if (ok && (filter & 2)) ok = !(triN.y < sh.rayMinNormalY[myRay]);
                if (ok) {
                    rec = CastRec{tHit, triP, nrm, triN, tri.triIndex};
                    myKey = ((unsigned long long)__float_as_uint(tHit) << 32) | (unsigned)tri.rank;
                    atomicMin(&sh.rayKey[myRay], myKey);
                    finished = true;
                }
            }
        }
        if (bspec) {
            if (bHelper && dist <= bRad) atomicOr(&sSpecBits[bOwnerRank], 1u << node);
            __syncthreads();
            if (served && phase == PH_REFINE) { // (its own step did not finish the bisection)
                const unsigned bits = sSpecBits[oRank];
                int n = ownContact ? 2 : 3;
                for (int k = 1; k < levels; ++k) {
                    const float mid = 0.5f * (lo + hi);
                    const bool c = (bits >> n) & 1;
                    if (c) hi = mid; else lo = mid;
                    refineK += 1; evals += 1;
                    n = 2 * n + (c ? 0 : 1);
                    if (refineK == 10) { phase = PH_FINAL; tEval = hi; break; }
                }
            }
        }
        if (spec) {
            // did the owner's own evaluation of this trip creep (still marching, advance == minAdvance)? then the helpers' points are its
            const int ownerCrept = __shfl((int)(phase == PH_MARCH && crawlRun > 0), owner, kWave);
            if (ownerCrept) {
                const bool contact = helper && cEval && dist <= bRadius + contactEps;
                const float adv = smax(dist - bRadius, bMinAdv);
                const bool breaks = helper && cEval && !contact && adv != bMinAdv;
                const unsigned long long ev = __ballot(helper && (!cEval || contact || breaks));
                const int lastHelper = 63 - __clzll((long long)helpMask);
                const int jl = ev ? __ffsll((long long)ev) - 1 : lastHelper; // the helper whose evaluation decides
                const int rJ = __shfl(h, jl, kWave);
                const float tJ = __shfl(cT, jl, kWave), tPrev = __shfl(cPrev, jl, kWave), advJ = __shfl(adv, jl, kWave);
                const int evalJ = __shfl((int)cEval, jl, kWave), contactJ = __shfl((int)contact, jl, kWave);
                if (lane == owner) {
                    if (!ev) {                    // every helper found another creeping step
                        iter += nHelp; evals += nHelp; crawlRun += nHelp; lastSafeT = tJ; t = tJ + bMinAdv;
                    } else if (!evalJ) {          // the loop ends at helper rJ's evaluation (budget or t > maxDistance): the next head sees it
                        iter += rJ; evals += rJ; lastSafeT = tPrev; t = tJ; crawlRun = 0;
                    } else if (contactJ) {        // contact: refineTOI(t0: lastSafeT, t1: t) :1361-1377
                        iter += rJ + 1; evals += rJ + 1; lastSafeT = tPrev; t = tJ; crawlRun = 0;
                        float k0 = smax(0.0f, smin(lastSafeT, len));
                        float k1 = smax(0.0f, smin(t, len));
                        lo = smin(k0, k1);
                        hi = smax(k0, k1);
                        if (hi - lo < 1e-5f) { phase = PH_FINAL; tEval = hi; }
                        else { phase = PH_REFINE; refineK = 0; }
                    } else {                      // this evaluation advances further than minAdvance: back to the ordinary march
                        iter += rJ + 1; evals += rJ + 1; lastSafeT = tJ; t = tJ + advJ; crawlRun = 0;
                    }
                }
            }
        }
        if (__any(finished)) {
            __syncthreads();
            if (finished && sh.rayKey[myRay] == myKey) sh.rayRec[myRay] = rec;
            bestToi = __uint_as_float((unsigned)(sh.rayKey[myRay] >> 32));
        }
    }
    if (evals) { atomicAdd(&sGroupCost[myRay / kMaxRays], (int)evals); st.evals += evals; }
    __syncthreads();
}

// Traversal of one cast pass (rays rb .. rb + R - 1, union box minP / maxP): its (ray, triangle) items join the shared queue;
// the queue is swept whenever the next batch might not fit.
// skipBox: triangles whose AABB overlaps [skipMin, skipMax] were already handled by an earlier pass of the same rays.
__device__ __forceinline__ void groupGather(const DevCollision& col, int rb, int R, F3 minP, F3 maxP, float radius, bool hasMinNormalY,
                                            float minNormalY, uint32_t mask, int& itemCount, WaveStats& st, bool skipBox = false,
                                            F3 skipMin = F3{0, 0, 0}, F3 skipMax = F3{0, 0, 0}) {
    const int lane = laneId();
    int stackSize = initTraversal(col), rangeCount = 0, candCount = 0;
    __syncthreads();
    while (stackSize > 0 || rangeCount > 0 || candCount > 0) {
        while ((stackSize > 0 || rangeCount > 0) && candCount < kWave) expandNodes(col, minP, maxP, mask, stackSize, rangeCount, candCount, st);
        const int n = candCount < kWave ? candCount : kWave;
        candCount -= n;
        st.candidates += n;
        if (itemCount + kWave * R > kItemCap) groupSweep(col, itemCount, st);
        int slot = -1;
        F3 v0{0, 0, 0}, v1{0, 0, 0}, v2{0, 0, 0};
        if (lane < n) {
            slot = sh.cand[candCount + lane];
            const float4* tp = reinterpret_cast<const float4*>(col.tris + slot);
            float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
            v0 = F3{t0.x, t0.y, t0.z}; v1 = F3{t0.w, t1.x, t1.y}; v2 = F3{t1.z, t1.w, t2.x};
            if (hasMinNormalY && tooSteepForGroundCast(v0, v1, v2, minNormalY)) { slot = -1; st.pruned += R; }
        }
        const F3 bmin = vmin(v0, vmin(v1, v2)), bmax = vmax(v0, vmax(v1, v2));
        if (skipBox && !boxDisjoint(bmin, bmax, skipMin, skipMax)) slot = -1;
        for (int r = rb; r < rb + R; ++r) {
            bool c = slot >= 0 && sh.rayValid[r] && !boxDisjoint(bmin, bmax, sh.rayMin[r], sh.rayMax[r]);
            if (c && sh.rayVertical[r]) { c = !verticalSweepMisses(sh.rayFrom[r].x, sh.rayFrom[r].z, radius, v0, v1, v2); st.pruned += c ? 0 : 1; }
            unsigned long long mc = __ballot(c);
            if (c) sh.items[itemCount + prefixCount(mc)] = (int)(((unsigned)r << kItemRayShift) | (unsigned)slot);
            itemCount += __popcll(mc);
        }
        __syncthreads();
    }
}

template <bool AGENTS>
__global__ __launch_bounds__(kWave, SGE_CCD_EXCLUSIVE ? 2 : SGE_GROUP_WAVES) void move_group_kernel(MoveLaunch K) {
    SGE_PAD_VGPRS();
#ifdef SGE_GROUP_TAIL_PRIO // experiment: issue priority for the launch's first wavefronts only (they hold the most expensive characters)
    if ((int)blockIdx.x * SGE_GROUP_TAIL_PRIO < (int)gridDim.x) __builtin_amdgcn_s_setprio(3);
#endif
    const int lane = laneId();
    WaveStats st{0, 0, 0, 0, 0, 0, 0};
    const DevCollision& col = K.col;
    const float dt = K.dt;
    // Members of this wavefront. With an order list (characters sorted by last step's cost, most expensive first; the multi-wave
    // launch's characters left out) wavefront w of W takes ranks w, 2W-1-w, 2W+w, 4W-1-w, ...: one character from every cost
    // stratum, the expensive end paired with the cheap end, so that no wavefront collects a cluster of expensive neighbours and
    // the launch's first wavefronts hold the most expensive characters. Without a list: kGroup consecutive characters.
    unsigned actMask = 0;
    {
        // The first K.solo wavefronts take ONE character each, the K.solo most expensive of the list: a character of 3,000-4,000
        // evaluations is ~0.33 ms of dependent cast passes by itself, and three companions whose traversals it would have to wait
        // for made the wavefront that held it the launch's last (0.94 M cycles against a median of 0.38 M).
        const int S = K.order ? K.solo : 0, W = (int)gridDim.x - S, w = (int)blockIdx.x - S;
        // (agent-scope loads: the list was written by the kernels just before this one)
        const int n = K.order ? __hip_atomic_load(K.orderCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : K.count;
        for (int g = 0; g < kGroup; ++g) {
            int rank = w < 0 ? (g == 0 ? w + S : n) : S + ((g & 1) ? (g + 1) * W - 1 - w : g * W + w);
            if (W <= 0) rank = g == 0 ? (int)blockIdx.x : n;
            int e = -1;
            if (rank < n) e = K.order ? __hip_atomic_load(K.order + rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : K.first + rank;
            if (e >= 0) actMask |= 1u << g;
            if (lane == 0) sGroupE[g] = e;
        }
    }
    if (actMask == 0) return;
    __syncthreads();
    long long pT0 = (long long)__builtin_amdgcn_s_memtime(), pGather = 0, pSweep = 0, pConsume = 0, pRounds = 0, pT;
    for (int g = 0; g < kGroup; ++g) {
        if (!((actMask >> g) & 1)) continue;
        const int e = sGroupE[g];
        const uint32_t* gb = reinterpret_cast<const uint32_t*>(K.crowd.bodies + e);
        const uint32_t* gp = reinterpret_cast<const uint32_t*>(K.crowd.params + e);
        const uint32_t* gc = reinterpret_cast<const uint32_t*>(K.crowd.controllers + e);
        const uint32_t* gs = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(K.scratch) + (size_t)e * kMoveScratchBytes);
        if (lane < 24) reinterpret_cast<uint32_t*>(&sBodyA[g])[lane] = gb[lane];
        if (lane < 16) reinterpret_cast<uint32_t*>(&sParamsA[g])[lane] = gp[lane];
        if (lane < 32) reinterpret_cast<uint32_t*>(&sCtrlA[g])[lane] = gc[lane];
        if (lane < (int)(sizeof(MoveState) / 4)) reinterpret_cast<uint32_t*>(&msA[g])[lane] = gs[lane];
    }
    if (lane < kGroup) sGroupCost[lane] = 0;
    __syncthreads();

    while (true) {
        unsigned phasePack = 0; // 4 bits per character: its phase at the head of this round (0xF: nothing to do)
        unsigned farMask = 0;   // characters whose fall probe still has its far pass to do this round
        bool any = false;
        int itemCount = 0;
        pT = (long long)__builtin_amdgcn_s_memtime();
        // ---------------- 1. every unfinished character sets up its next cast pass and gathers its work items ----------------
        for (int g = 0; g < kGroup; ++g) {
            unsigned myPhase = 0xF;
            if ((actMask >> g) & 1) {
                MoveState& ms = msA[g];
                const sge_controller_params& P = sParamsA[g];
                const int e = sGroupE[g], rb = g * kMaxRays;
                if (ms.phase == MP_SLIDE) { // head of the slide loop :1674-1676
                    float len = length(ms.remaining);
                    if (ms.it >= P.maxSlideIterations || len < 1e-6f) { __syncthreads(); ms.phase = MP_GROUND_CENTER; __syncthreads(); }
                }
                const int phase = ms.phase;
                if (phase != MP_DONE) {
                    myPhase = (unsigned)phase;
                    any = true;
                    int R = 0;
                    bool blocking = false;
                    if (phase == MP_SLIDE) {
                        blocking = true; R = 1;
                        sh.rayFrom[rb] = ms.position; sh.rayDelta[rb] = ms.remaining;
                    } else if (phase == MP_GROUND_CENTER) { // snap cast + fall probe (+ speculative offset casts), see move_kernel
                        sh.rayFrom[rb] = ms.position; sh.rayFrom[rb + 1] = ms.position;
                        sh.rayDelta[rb] = P.snapDistance > 0 ? F3{0, -1, 0} * P.snapDistance : F3{0, 0, 0};
                        sh.rayDelta[rb + 1] = P.fallProbeDistance > 0 ? F3{0, -1, 0} * P.fallProbeDistance : F3{0, 0, 0};
                        // (the flag goes through readfirstlane: hipcc 7.2 otherwise selects `spec ? 2 : 0` with s_cselect on a stale SCC
                        // after comparing the loaded hint byte in VCC)
                        const bool spec = __builtin_amdgcn_readfirstlane((int)(ms.wasGroundedNear && P.snapDistance > 0 && K.hint && K.hint[e] != 0)) != 0;
                        R = spec ? 6 : 2;
                        ms.sampleK = spec ? 2 : 0;
                        if (spec) {
                            const float offset = P.radius * 0.6f;
                            const F3 snapDelta = F3{0, -1, 0} * P.snapDistance;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float ox = k == 0 ? offset : (k == 1 ? -offset : 0.0f);
                                float oz = k == 2 ? offset : (k == 3 ? -offset : 0.0f);
                                sh.rayFrom[rb + 2 + k] = ms.position + F3{ox, 0, oz};
                                sh.rayDelta[rb + 2 + k] = snapDelta;
                            }
                        }
                    } else if (phase == MP_GROUND_SAMPLE && ms.sampleK == 0) { // the four offset casts of :898-921
                        R = 4;
                        const float offset = P.radius * 0.6f;
                        const F3 snapDelta = F3{0, -1, 0} * P.snapDistance;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float ox = k == 0 ? offset : (k == 1 ? -offset : 0.0f);
                            float oz = k == 2 ? offset : (k == 3 ? -offset : 0.0f);
                            sh.rayFrom[rb + k] = ms.position + F3{ox, 0, oz};
                            sh.rayDelta[rb + k] = snapDelta;
                        }
                    }
                    __syncthreads();
                    if (R > 0) {
                        F3 minP, maxP;
                        const int nValid = groupSetupRays(col, rb, R, P.radius, P.halfHeight, blocking, !blocking, P.minGroundDot, minP, maxP);
                        if (nValid > 0) {
                            st.queries += nValid;
                            if (phase == MP_GROUND_CENTER && sh.rayValid[rb] && sh.rayValid[rb + 1]) {
                                // near pass: everything but the fall probe decides the traversal box
                                F3 nMin = sh.rayMin[rb], nMax = sh.rayMax[rb];
                                for (int r = rb + 2; r < rb + R; ++r)
                                    if (sh.rayValid[r]) { nMin = vmin(nMin, sh.rayMin[r]); nMax = vmax(nMax, sh.rayMax[r]); }
                                __syncthreads();
                                sNearMin[g] = nMin; sNearMax[g] = nMax;
                                farMask |= 1u << g;
                                minP = nMin; maxP = nMax;
                            }
                            groupGather(col, rb, R, minP, maxP, P.radius, !blocking, P.minGroundDot, P.collisionMask, itemCount, st);
                        }
                    }
                }
            }
            phasePack |= myPhase << (4 * g);
        }
        if (!any) break;
        pRounds += 1;
        pGather += (long long)__builtin_amdgcn_s_memtime() - pT; pT = (long long)__builtin_amdgcn_s_memtime();
        // ---------------- 2. all lanes sweep all characters' items ----------------
        groupSweep(col, itemCount, st);
        pSweep += (long long)__builtin_amdgcn_s_memtime() - pT; pT = (long long)__builtin_amdgcn_s_memtime();
        // ---------------- 2b. far pass of the fall probes, clipped to what the near pass found ----------------
        if (farMask) {
            for (int g = 0; g < kGroup; ++g) {
                if (!((farMask >> g) & 1)) continue;
                const sge_controller_params& P = sParamsA[g];
                const int r = g * kMaxRays + 1;
                const F3 from = sh.rayFrom[r];
                const float len = sh.rayLen[r];
                float reach = len;
                if (rayHit(r)) reach = smin(len, __uint_as_float((unsigned)(sh.rayKey[r] >> 32)) + (0.05f + 1e-4f * fabsf(from.y)));
                // the capsule's AABB swept over [0, reach] along -Y
                const F3 ext{P.radius, P.radius, P.radius};
                const F3 fMin = F3{from.x, from.y - P.halfHeight - reach, from.z} - ext;
                const F3 fMax = F3{from.x, from.y + P.halfHeight, from.z} + ext;
                const F3 nMin = sNearMin[g], nMax = sNearMax[g];
                if (fMin.y >= nMin.y) continue; // inside the near box (same x / z extent: same origin and capsule)
                __syncthreads();
                sh.rayMin[r] = vmax(sh.rayMin[r], fMin); sh.rayMax[r] = vmin(sh.rayMax[r], fMax);
                __syncthreads();
                groupGather(col, r, 1, sh.rayMin[r], sh.rayMax[r], P.radius, true, P.minGroundDot, P.collisionMask, itemCount, st, true, nMin, nMax);
            }
            groupSweep(col, itemCount, st);
        }
        // ---------------- 3. every character consumes its rays ----------------
        for (int g = 0; g < kGroup; ++g) {
            const int phase = (int)((phasePack >> (4 * g)) & 0xF);
            if (phase == 0xF) continue;
            MoveState& ms = msA[g];
            const sge_controller_params& P = sParamsA[g];
            if (phase == MP_SLIDE) {
                if (AGENTS) { // AgentSweepSolver.bestHit :1053-1091 (independent of the static hit)
                    const bool hasAgent = (P.agentFlags & SGE_AGENT_PRESENT) != 0;
                    const bool selfSolid = hasAgent && (P.agentFlags & SGE_AGENT_SOLID);
                    if ((K.stages & SGE_STAGE_AGENTS) && selfSolid && K.agents.all != nullptr) {
                        const F3 remaining = ms.remaining;
                        const float selfRadius = (hasAgent && (P.agentFlags & SGE_AGENT_RADIUS_OVERRIDE)) ? P.agentRadiusOverride : P.radius;
                        float aToi = 0; F3 aNormal{0, 0, 0};
                        bool have = waveAgentBestHit(K.agents, ms.position, remaining, length(remaining), ms.baseMoveLen, dt,
                                                     K.agents.selfOffset + sGroupE[g], selfRadius, P.halfHeight, aToi, aNormal);
                        __syncthreads();
                        ms.aHave = have ? 1 : 0; ms.aToi = aToi; ms.aNormal = aNormal;
                        __syncthreads();
                    }
                }
                consumeSlide(g);
            } else if (phase == MP_GROUND_CENTER) consumeGroundCenter(g);
            else if (phase == MP_GROUND_SAMPLE) consumeGroundSample(g);
            __syncthreads();
            if (ms.phase == MP_GROUND_EVAL) { groundEval(g); __syncthreads(); }
            // the offset casts rode along with the centre pass: their results are already there
            if (ms.phase == MP_GROUND_SAMPLE && ms.sampleK != 0) { consumeGroundSample(g); __syncthreads(); }
            if (ms.phase == MP_FINISH) { finishStep(g); __syncthreads(); }
        }
        pConsume += (long long)__builtin_amdgcn_s_memtime() - pT;
    }
    if (K.waveProf && lane == 0) {
        unsigned long long* w = K.waveProf + (size_t)blockIdx.x * 8;
        w[0] = (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - pT0); w[1] = (unsigned long long)pGather; w[2] = (unsigned long long)pSweep;
        w[3] = (unsigned long long)pConsume; w[4] = (unsigned long long)pRounds; w[5] = st.trips; w[6] = st.steps; w[7] = (unsigned long long)pT0;
    }

    // ---------------- write-back ----------------
    for (int g = 0; g < kGroup; ++g) {
        if (!((actMask >> g) & 1)) continue;
        const int e = sGroupE[g];
        MoveState& ms = msA[g];
        if (ms.phase == MP_DONE) {
            D3 velocity = ms.velocity;
            __syncthreads();
            sBodyA[g].linearVelocity[0] = velocity.x; sBodyA[g].linearVelocity[1] = velocity.y; sBodyA[g].linearVelocity[2] = velocity.z;
        }
        __syncthreads();
        storeCharacter(K.crowd, e, g, lane);
        if (lane == 0) {
            if (K.cost) K.cost[e] = sGroupCost[g];
            if (K.hint) K.hint[e] = (uint8_t)(ms.sampled != 0);
        }
    }
    if (K.stats) {
        unsigned v = st.evals, pr = st.pruned; // per-lane counts
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o, kWave); pr += __shfl_xor(pr, o, kWave); }
        if (lane == 0) {
            unsigned long long* sp = statShard(K.stats);
            if (st.queries) atomicAdd(&sp[0], (unsigned long long)st.queries);
            if (st.candidates) atomicAdd(&sp[1], (unsigned long long)st.candidates);
            if (v) atomicAdd(&sp[2], (unsigned long long)v);
            if (st.overflow) atomicAdd(&sp[3], (unsigned long long)st.overflow);
            if (st.steps) atomicAdd(&sp[4], (unsigned long long)st.steps);
            if (st.trips) atomicAdd(&sp[5], (unsigned long long)st.trips);
            if (pr) atomicAdd(&sp[6], (unsigned long long)pr);
        }
    }
}

// ---------------------------------------------------------------------------
// AgentSeparationSystem (Systems.swift:1906-2210), canonical order = character index
// ---------------------------------------------------------------------------
// The pair loop is sequential by definition — pair (i, j) reads the positions and velocities the pairs before it wrote, and agent
// i's stale copy `a` taken at the head of loop i (:1952) — and in a crowd the chain of dependent pairs runs through every agent, so
// there is nothing to spread over the chip: ONE wavefront walks the loop in the reference's order (per agent i: the 3 x 3 cells
// around it, dz outer, dx inner, the cell's agents in ascending index, j > i) with the live positions / velocities in LDS. The
// grid of :1915-1944 is only a filter, so it is kept as each agent's cell at rebuild time and a neighbour cell's members are found
// by a 64-wide scan of those; the two capsuleCastBlocking calls of a pair (:2004-2027) run as one two-ray pass (each ray with its
// own capsule). The post-process (:2047-2140) is independent per agent: one wavefront each.
struct SepAgentDev { float position[3], velocity[3], start[3]; float radius, halfHeight, invWeight; int entity; int pad; };
struct SepLaunch {
    DevCrowd crowd; DevCollision col;
    int iterations; float separationMargin, heightMargin;
    SepAgentDev* agents; int* count; // count[0]: listed agents (0: nothing to post-process)
};
__shared__ F3 sSepPos[SGE_MAX_SEPARATION_AGENTS], sSepVel[SGE_MAX_SEPARATION_AGENTS];
__shared__ int sSepCellX[SGE_MAX_SEPARATION_AGENTS], sSepCellZ[SGE_MAX_SEPARATION_AGENTS];

__global__ __launch_bounds__(kWave) void separation_resolve_kernel(SepLaunch K) {
    const int lane = laneId();
    const DevCollision& col = K.col;
    WaveStats st{0, 0, 0, 0, 0, 0, 0};
    const int N = K.crowd.count;
    // ---- the agent list (:2166-2187), in character order ----
    int n = 0;
    float maxRadius = 0;
    for (int base = 0; base < N; base += kWave) {
        const int e = base + lane;
        bool solid = false;
        float radius = 0, invWeight = 0;
        if (e < N) {
            const sge_controller_params& P = K.crowd.params[e];
            const bool present = (P.agentFlags & SGE_AGENT_PRESENT) != 0;
            solid = present ? (P.agentFlags & SGE_AGENT_SOLID) != 0 : true; // aStore[e] ?? AgentCollisionComponent()
            radius = (present && (P.agentFlags & SGE_AGENT_RADIUS_OVERRIDE)) ? P.agentRadiusOverride : P.radius;
            const float massWeight = present ? P.agentMassWeight : 1.0f;
            invWeight = massWeight > 0 ? 1.0f / massWeight : 0.0f;
        }
        const unsigned long long m = __ballot(solid);
        const int idx = n + prefixCount(m);
        if (solid && idx < SGE_MAX_SEPARATION_AGENTS) {
            const sge_body_state& b = K.crowd.bodies[e];
            SepAgentDev a;
            for (int k = 0; k < 3; ++k) { a.position[k] = (float)b.position[k]; a.velocity[k] = (float)b.linearVelocity[k]; a.start[k] = a.position[k]; }
            a.radius = radius; a.halfHeight = K.crowd.params[e].halfHeight; a.invWeight = invWeight; a.entity = e; a.pad = 0;
            K.agents[idx] = a;
            sSepPos[idx] = F3{a.position[0], a.position[1], a.position[2]};
            sSepVel[idx] = F3{a.velocity[0], a.velocity[1], a.velocity[2]};
        }
        float r = solid ? radius : 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) r = smax(r, __shfl_xor(r, o, kWave));
        maxRadius = smax(maxRadius, r);
        n += __popcll(m);
    }
    __syncthreads();
    if (N <= 1 || n <= 1 || n > SGE_MAX_SEPARATION_AGENTS) { if (lane == 0) K.count[0] = 0; return; } // guards :2153, :2189 (capacity: sge_tick)
    const float cellSize = smax(maxRadius * 2 + K.separationMargin, 0.001f);
    const float separationMargin = K.separationMargin, heightMargin = K.heightMargin;
    for (int it = 0; it < K.iterations; ++it) {
        // grid.rebuild (:1930-1936): every agent's cell at this moment
        for (int i = lane; i < n; i += kWave) {
            const F3 p = sSepPos[i];
            sSepCellX[i] = (int)floorf(p.x / cellSize);
            sSepCellZ[i] = (int)floorf(p.z / cellSize);
        }
        __syncthreads();
        for (int i = 0; i < n; ++i) { // AgentSeparationResolver.resolve :1947-2046
            const F3 aPos = sSepPos[i], aVel = sSepVel[i]; // the stale copy `a`
            const SepAgentDev A = K.agents[i];
            const sge_controller_params& Pa = K.crowd.params[A.entity];
            const float aSkin = Pa.skinWidth, aMinGroundDot = Pa.minGroundDot;
            const uint32_t aMask = Pa.collisionMask;
            const int cx = (int)floorf(aPos.x / cellSize), cz = (int)floorf(aPos.z / cellSize);
            for (int dz = -1; dz <= 1; ++dz) {
                for (int dx = -1; dx <= 1; ++dx) {
                    const int tx = cx + dx, tz = cz + dz;
                    for (int base = (i + 1) & ~(kWave - 1); base < n; base += kWave) {
                        const int jl = base + lane;
                        unsigned long long members = __ballot(jl > i && jl < n && sSepCellX[jl] == tx && sSepCellZ[jl] == tz);
                        while (members) {
                            const int j = base + __ffsll((long long)members) - 1;
                            members &= members - 1;
                            const F3 bPos = sSepPos[j], bVel = sSepVel[j];
                            const SepAgentDev B = K.agents[j];
                            const sge_controller_params& Pb = K.crowd.params[B.entity];
                            const float aMin = aPos.y - A.halfHeight, aMax = aPos.y + A.halfHeight;
                            const float bMin = bPos.y - B.halfHeight, bMax = bPos.y + B.halfHeight;
                            const float ddx = aPos.x - bPos.x, ddz = aPos.z - bPos.z;
                            const float distSq = ddx * ddx + ddz * ddz;
                            const float skinAllowance = smin(aSkin, Pb.skinWidth);
                            const float margin = smin(separationMargin, skinAllowance);
                            const float minDist = A.radius + B.radius + margin;
                            const bool heightSeparated = aMax < bMin - heightMargin || aMin > bMax + heightMargin;
                            if (heightSeparated) continue;
                            if (distSq >= minDist * minDist) continue;
                            const float dist = sqrtf(smax(distSq, 1e-8f));
                            const float nx = ddx / dist, nz = ddz / dist;
                            const float penetration = minDist - dist;
                            const float wSum = A.invWeight + B.invWeight;
                            if (wSum <= 0) continue;
                            const float corr = penetration / wSum;
                            F3 moveA{nx * corr * A.invWeight, 0, nz * corr * A.invWeight};
                            F3 moveB{-nx * corr * B.invWeight, 0, -nz * corr * B.invWeight};
                            const F3 relV = aVel - bVel;
                            const float vn = relV.x * nx + relV.z * nz;
                            F3 velI = sSepVel[i], velJ = bVel;
                            if (vn < 0) {
                                const float impulse = -vn;
                                const float scaleA = A.invWeight / wSum, scaleB = B.invWeight / wSum;
                                velI.x += nx * impulse * scaleA; velI.z += nz * impulse * scaleA;
                                velJ.x -= nx * impulse * scaleB; velJ.z -= nz * impulse * scaleB;
                            }
                            const F3 posI = sSepPos[i]; // the live position of agent i (:2008)
                            __syncthreads();
                            sSepVel[i] = velI; sSepVel[j] = velJ;
                            // the two blocking casts of :2004-2027 as one pass: ray 0 = agent i along moveA, ray 1 = agent j along moveB
                            const float eps = 1e-6f;
                            const bool castA = length(moveA) > eps, castB = length(moveB) > eps;
                            bool blockedA = false, blockedB = false;
                            if (castA || castB) {
                                sh.rayFrom[0] = posI; sh.rayDelta[0] = castA ? moveA : F3{0, 0, 0};
                                sh.rayFrom[1] = bPos; sh.rayDelta[1] = castB ? moveB : F3{0, 0, 0};
                                __syncthreads();
                                int itemCount = 0;
                                F3 minP, maxP;
                                if (groupSetupRays(col, 0, 1, A.radius, A.halfHeight, true, false, 0.0f, minP, maxP) > 0)
                                    groupGather(col, 0, 1, minP, maxP, A.radius, false, 0.0f, aMask, itemCount, st);
                                if (groupSetupRays(col, 1, 1, B.radius, B.halfHeight, true, false, 0.0f, minP, maxP) > 0)
                                    groupGather(col, 1, 1, minP, maxP, B.radius, false, 0.0f, Pb.collisionMask, itemCount, st);
                                groupSweep(col, itemCount, st);
                                blockedA = castA && rayHit(0) && sh.rayRec[0].toi <= aSkin && sh.rayRec[0].normal.y < aMinGroundDot;
                                blockedB = castB && rayHit(1) && sh.rayRec[1].toi <= Pb.skinWidth && sh.rayRec[1].normal.y < Pb.minGroundDot;
                            }
                            if (blockedA && !blockedB) {
                                moveA = F3{0, 0, 0};
                                moveB = F3{-nx * penetration, 0, -nz * penetration};
                            } else if (blockedB && !blockedA) {
                                moveB = F3{0, 0, 0};
                                moveA = F3{nx * penetration, 0, nz * penetration};
                            } else if (blockedA && blockedB) {
                                continue;
                            }
                            __syncthreads();
                            sSepPos[i] = posI + moveA;
                            sSepPos[j] = bPos + moveB;
                            __syncthreads();
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < n; i += kWave) {
        const F3 p = sSepPos[i], v = sSepVel[i];
        K.agents[i].position[0] = p.x; K.agents[i].position[1] = p.y; K.agents[i].position[2] = p.z;
        K.agents[i].velocity[0] = v.x; K.agents[i].velocity[1] = v.y; K.agents[i].velocity[2] = v.z;
    }
    if (lane == 0) K.count[0] = n;
}

// AgentSeparationPostProcessor.apply + the write-back of :2201-2215, one wavefront per listed agent
__global__ __launch_bounds__(kWave, 3) void separation_post_kernel(SepLaunch K) {
    const int idx = blockIdx.x;
    if (idx >= K.count[0]) return;
    const int lane = laneId();
    const DevCollision& col = K.col;
    WaveStats st{0, 0, 0, 0, 0, 0, 0};
    const SepAgentDev A = K.agents[idx];
    const int e = A.entity;
    {
        const uint32_t* gb = reinterpret_cast<const uint32_t*>(K.crowd.bodies + e);
        const uint32_t* gp = reinterpret_cast<const uint32_t*>(K.crowd.params + e);
        const uint32_t* gc = reinterpret_cast<const uint32_t*>(K.crowd.controllers + e);
        if (lane < 24) reinterpret_cast<uint32_t*>(&sBodyA[0])[lane] = gb[lane];
        if (lane < 16) reinterpret_cast<uint32_t*>(&sParamsA[0])[lane] = gp[lane];
        if (lane < 32) reinterpret_cast<uint32_t*>(&sCtrlA[0])[lane] = gc[lane];
        __syncthreads();
    }
    sge_body_state& body = sBodyA[0];
    const sge_controller_params& P = sParamsA[0];
    sge_controller_state& C = sCtrlA[0];
    const F3 start{A.start[0], A.start[1], A.start[2]};
    F3 position{A.position[0], A.position[1], A.position[2]};
    D3 bodyVelocity{body.linearVelocity[0], body.linearVelocity[1], body.linearVelocity[2]};
    const F3 delta = position - start;
    const float len = length(delta);
    bool moved = false;
    if (len > 1e-6f) {
        moved = true;
        F3 remaining = delta;
        position = start;
        for (int s = 0; s < 2; ++s) { // slideIterations :2063
            const float segLen = length(remaining);
            if (segLen < 1e-6f) break;
            __syncthreads();
            sh.rayCount = 1; sh.rayFrom[0] = position; sh.rayDelta[0] = remaining;
            __syncthreads();
            waveCastRays(col, A.radius, A.halfHeight, true, false, 0.0f, P.collisionMask, st);
            if (rayHit(0)) {
                SlideHit hit;
                hit.isStatic = true; hit.s = sh.rayRec[0]; hit.aToi = 0; hit.aNormal = F3{0, 0, 0};
                const bool done = resolveHit(remaining, segLen, hit, P, C, false, false, bodyVelocity, position, false, F3{0, 0, 0}, kAgentSeparation);
                if (done) break;
            } else {
                position = position + remaining;
                remaining = F3{0, 0, 0};
                break;
            }
        }
    }
    __syncthreads();
    if (moved && bodyVelocity.y <= 0 && P.snapDistance > 0) { // :2108-2136
        const F3 down{0, -1, 0};
        sh.rayCount = 1; sh.rayFrom[0] = position; sh.rayDelta[0] = down * P.snapDistance;
        __syncthreads();
        waveCastRays(col, A.radius, A.halfHeight, false, true, P.minGroundDot, P.collisionMask, st);
        if (rayHit(0) && sh.rayRec[0].toi <= P.snapDistance) {
            const CastRec hit = sh.rayRec[0];
            const float rawMove = smax(hit.toi - P.groundSnapSkin, 0.0f);
            const float moveDist = smin(rawMove, P.groundSnapMaxStep);
            position = position + down * moveDist;
            __syncthreads();
            uint32_t flags = C.flags | SGE_CTRL_GROUNDED;
            if (hit.toi <= smax(P.groundSnapSkin, P.skinWidth)) flags |= SGE_CTRL_GROUNDED_NEAR; else flags &= ~(uint32_t)SGE_CTRL_GROUNDED_NEAR;
            C.flags = flags;
            const F3 gn = col.materials[hit.triIndex].flatten ? F3{0, 1, 0} : hit.triNormal;
            st3(C.groundNormal, gn);
            C.groundTriangleIndex = hit.triIndex;
        }
    }
    __syncthreads();
    const D3 pd = toD(position), vd = toD(F3{A.velocity[0], A.velocity[1], A.velocity[2]});
    body.position[0] = pd.x; body.position[1] = pd.y; body.position[2] = pd.z;
    body.linearVelocity[0] = vd.x; body.linearVelocity[1] = vd.y; body.linearVelocity[2] = vd.z;
    __syncthreads();
    storeCharacter(K.crowd, e, 0, lane);
}


// ---------------------------------------------------------------------------
// AgentSeparationSystem for crowds (more solid agents than one wavefront's LDS holds): the pair loop as a dataflow over agents
// ---------------------------------------------------------------------------
// What the reference's loop fixes is, per agent, the ORDER in which the loops of other agents read and write it:
//   - loop i takes its copy `a` of agent i when every loop i' < i that can pair with i has done so (:1954), and after that nobody
//     touches agent i but loop i itself (pairs are (i, j > i));
//   - pair (i, j) reads and writes agent j after every pair (i', j) with i' < i (:1961, :1999-2000, :2040).
// Loops that share no agent commute. So every agent c carries a counter ver[c] of the earlier loops that have PASSED it, and every
// loop k knows, for each agent c it may touch, its rank among c's passers: it waits for ver[c] == rank, pairs with c (or not),
// and stores ver[c] = rank + 1; loop c starts at ver[c] == need[c]. Who "may touch" whom is decided from the cells at the head of
// the pass (the grid is fixed for the pass, :2187): loop k pairs with the agents in the 3 x 3 cells around its LIVE cell (:1955-1960),
// which is at most one cell away from its cell at the head of the pass unless it has been pushed further than a cell (checked; the
// pass is then redone serially) — so the agents c > k in the 5 x 5 cells around k's pass-start cell are its candidates, and it passes
// all of them: the ones it pairs with in the reference's order, the others afterwards. (Ordering only agents that start the pass
// close enough to meet — one cell + what two agents may move — shortens the chains but needs a bound on every agent's movement
// in the pass; a quarter cell and half a cell were both exceeded in crowded scenes, 3 of 45 and 1 of 45 steps of the 192-agent
// test, every step of an 8,192-agent crowd without character-vs-character sweeps, and each miss costs a serial pass.) Wavefronts draw loops from a ticket counter in index order, so the lowest
// unfinished loop never waits for an unfinished one: no deadlock, whatever the number of resident wavefronts.
// Positions and velocities cross CUs (and XCDs) through agent-scope atomic loads / stores. In the first form of the loops
// (sep_flow_kernel, kept behind SGE_SEPARATION_BVH_CASTS=1 as the reference point of tests/test_separation.py) a release is the data
// stores, a wait for them, then the counter store (MI355X_MICROARCH.md, "Valid forms"); the second form (sep_flow2_kernel, the
// default) carries the version inside the data. The result is the reference's, bit for bit, for any crowd; what varies is the depth
// of the dependency graph (tools/separation_depth.py prints it).
constexpr int kSepTriCap = 128;       // triangles cached per agent; an agent with more in reach casts through the BVH
constexpr float kSepTriReach = 1.5f;  // the cached box reaches this far beyond the capsule at the head of the pass (0.5 / 1 / 1.5 / 2.5: 38.9 / 34.4 / 33.7 / 35.0 ms per step)
constexpr int kSepMaxCand = 1024; // candidates a loop tracks (agents of higher index in its 5 x 5 cells); more: the pass runs serially.
                                  // (256 until round 3: a crowd spawned at 1.6 units' spacing — 31,250 agents on the benchmark scene, one
                                  // GPU's share of configs[3] — has 300-500 per agent and every pass fell back to one wavefront, 2.1 s per
                                  // step; 1,024 are 16 KB of LDS per loop and 8 KB of list per agent)
struct SepStaticDev { float radius, halfHeight, invWeight, skinWidth, minGroundDot; uint32_t mask; float posY, velY; };
struct SepFlow {
    int* cell;          // [n][2] cell of every agent at the head of the pass (:1940-1944)
    int* bucketStart;   // [H + 1] hashed cells: agents sorted by (bucket, index)
    int* bucketCursor;  // [H]
    int* bucketItems;   // [n]
    int* need;          // [n] agents of lower index in the 5 x 5 cells around the agent
    int* ver;           // [n] how many of them have passed it
    int* candCount;     // [n]
    int2* cand;         // [n][kSepMaxCand] (agent of higher index in the 5 x 5 cells, this loop's rank among that agent's passers)
    int* control;       // [0] listed agents, [1] ticket, [2] redo flag of the pass, [3] cell size (float bits), [4] depth diagnostics
    float* backup;      // [n][6] position, velocity at the head of the pass
    SepStaticDev* stat; // [n] what a pair reads of an agent that no pair changes (sep_cells_kernel)
    unsigned long long* live; // [n][4] position x, z and velocity x, z, each an 8-byte granule (version << 32 | float bits): the data is the flag
    float4* triCache;   // [n][kSepTriCap][3] the 48-B records of the triangles in reach of every agent during the pass (sep_tricache_kernel)
    float* triBox;      // [n][8] the box those were gathered for (min xyz, max xyz) and their number (int bits; -1: more than kSepTriCap)
    int H;
    int trace;          // (diagnostics build: this pass writes the timeline)
    int noDefer;        // (experiments: SGE_SEPARATION_NO_DEFER=1, every loop waits for its outer ring as well)
    int retry;          // this launch belongs to the second attempt of a pass (candidates from 7 x 7 cells): it runs only if control[8] says so
    float triReach;     // the cached box reaches this far beyond the capsule at the head of the pass (kSepTriReach; SGE_SEPARATION_REACH)
    int reach;          // candidates come from the (2 reach + 1)^2 cells around an agent's cell at the head of the pass: 2 = one cell of
                        // movement + the 3 x 3 pair list (the rule); 3 after a step in which an agent was pushed further than a cell
};
__device__ __forceinline__ unsigned sepHash(int cx, int cz, int H) { return ((unsigned)cx * 73856093u ^ (unsigned)cz * 19349663u) & (unsigned)(H - 1); }
__device__ __forceinline__ float sepLoad(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sepStore(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ F3 sepLoad3(const float* p) { return F3{sepLoad(p), sepLoad(p + 1), sepLoad(p + 2)}; }
__device__ __forceinline__ void sepStore3(float* p, F3 v) { sepStore(p, v.x); sepStore(p + 1, v.y); sepStore(p + 2, v.z); }
__device__ __forceinline__ void sepWait(const int* counter, int value) {
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != value) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ void sepRelease(int* counter, int value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the data stores of this wavefront have left it
    __hip_atomic_store(counter, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the agent list (:2166-2187) in character order, its size and the cell size: one workgroup, ordered compaction
__global__ __launch_bounds__(1024) void sep_list_kernel(SepLaunch K, SepFlow F) {
    __shared__ int sWaveCount[16];
    __shared__ float sWaveMax[16];
    __shared__ int sBase;
    const int N = K.crowd.count, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) sBase = 0;
    float maxRadius = 0;
    __syncthreads();
    for (int base = 0; base < N; base += 1024) {
        const int e = base + tid;
        bool solid = false;
        float radius = 0, invWeight = 0;
        if (e < N) {
            const sge_controller_params& P = K.crowd.params[e];
            const bool present = (P.agentFlags & SGE_AGENT_PRESENT) != 0;
            solid = present ? (P.agentFlags & SGE_AGENT_SOLID) != 0 : true; // aStore[e] ?? AgentCollisionComponent()
            radius = (present && (P.agentFlags & SGE_AGENT_RADIUS_OVERRIDE)) ? P.agentRadiusOverride : P.radius;
            const float massWeight = present ? P.agentMassWeight : 1.0f;
            invWeight = massWeight > 0 ? 1.0f / massWeight : 0.0f;
        }
        const unsigned long long m = __ballot(solid);
        if (lane == 0) sWaveCount[wv] = __popcll(m);
        __syncthreads();
        int off = sBase;
        for (int w = 0; w < wv; ++w) off += sWaveCount[w];
        if (solid) {
            const int idx = off + prefixCount(m);
            const sge_body_state& b = K.crowd.bodies[e];
            SepAgentDev a;
            for (int k = 0; k < 3; ++k) { a.position[k] = (float)b.position[k]; a.velocity[k] = (float)b.linearVelocity[k]; a.start[k] = a.position[k]; }
            a.radius = radius; a.halfHeight = K.crowd.params[e].halfHeight; a.invWeight = invWeight; a.entity = e; a.pad = 0;
            K.agents[idx] = a;
            maxRadius = smax(maxRadius, radius);
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += sWaveCount[w]; sBase += t; }
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxRadius = smax(maxRadius, __shfl_xor(maxRadius, o, kWave));
    if (lane == 0) sWaveMax[wv] = maxRadius;
    __syncthreads();
    if (tid == 0) {
        float r = 0;
        for (int w = 0; w < 16; ++w) r = smax(r, sWaveMax[w]);
        const int n = sBase;
        const bool run = !(N <= 1 || n <= 1); // guards :2153, :2189
        F.control[0] = run ? n : 0;
        F.control[3] = __float_as_int(smax(r * 2 + K.separationMargin, 0.001f));
        K.count[0] = run ? n : 0;
    }
}

// head of a pass: grid.rebuild (:1930-1936) as hashed buckets; counters back to zero; the pass's start state kept for a serial redo
__global__ void sep_cells_kernel(SepLaunch K, SepFlow F) {
    const int n = F.control[0], i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { F.control[1] = 0; F.control[2] = 0; }
    if (i >= n) return;
    const float cellSize = __int_as_float(F.control[3]);
    const SepAgentDev& a = K.agents[i];
    const int cx = (int)floorf(a.position[0] / cellSize), cz = (int)floorf(a.position[2] / cellSize);
    F.cell[2 * i] = cx; F.cell[2 * i + 1] = cz;
    atomicAdd(&F.bucketCursor[sepHash(cx, cz, F.H)], 1);
    F.ver[i] = 0;
    for (int k = 0; k < 3; ++k) { F.backup[i * 6 + k] = a.position[k]; F.backup[i * 6 + 3 + k] = a.velocity[k]; }
    const sge_controller_params& P = K.crowd.params[a.entity];
    F.stat[i] = SepStaticDev{a.radius, a.halfHeight, a.invWeight, P.skinWidth, P.minGroundDot, P.collisionMask, a.position[1], a.velocity[1]};
    const float lv[4] = {a.position[0], a.position[2], a.velocity[0], a.velocity[2]};
    for (int k = 0; k < 4; ++k) F.live[(size_t)i * 4 + k] = (unsigned long long)__float_as_uint(lv[k]); // version 0
}
// exclusive scan of the bucket counts (one workgroup; every thread a contiguous run), counts -> zero (they become the scatter cursors)
__global__ __launch_bounds__(1024) void sep_scan_kernel(SepFlow F) {
    __shared__ int sPart[1024];
    const int tid = threadIdx.x, per = (F.H + 1023) / 1024, lo = tid * per, hi = min(F.H, lo + per);
    int s = 0;
    for (int b = lo; b < hi; ++b) s += F.bucketCursor[b];
    sPart[tid] = s;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int t = 0; t < 1024; ++t) { const int c = sPart[t]; sPart[t] = run; run += c; } F.bucketStart[F.H] = run; }
    __syncthreads();
    int run = sPart[tid];
    for (int b = lo; b < hi; ++b) { const int c = F.bucketCursor[b]; F.bucketStart[b] = run; F.bucketCursor[b] = 0; run += c; }
}
__global__ void sep_scatter_kernel(SepFlow F) {
    const int n = F.control[0], i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned b = sepHash(F.cell[2 * i], F.cell[2 * i + 1], F.H);
    F.bucketItems[F.bucketStart[b] + atomicAdd(&F.bucketCursor[b], 1)] = i;
}
// a cell's list is in agent order (:1934-1937): sort every bucket by index (a handful of entries each)
__global__ void sep_sort_kernel(SepFlow F) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= F.H) return;
    const int lo = F.bucketStart[b], hi = F.bucketStart[b + 1];
    for (int p = lo + 1; p < hi; ++p) {
        const int v = F.bucketItems[p];
        int q = p - 1;
        while (q >= lo && F.bucketItems[q] > v) { F.bucketItems[q + 1] = F.bucketItems[q]; --q; }
        F.bucketItems[q + 1] = v;
    }
}
// agents in the (2 reach + 1)^2 cells around (cx, cz) with index below `limit`
__device__ __forceinline__ int sepCountBelow(const SepFlow& F, int cx, int cz, int limit) {
    int cnt = 0;
    for (int dz = -F.reach; dz <= F.reach; ++dz)
        for (int dx = -F.reach; dx <= F.reach; ++dx) {
            const int tx = cx + dx, tz = cz + dz;
            const unsigned b = sepHash(tx, tz, F.H);
            for (int p = F.bucketStart[b]; p < F.bucketStart[b + 1]; ++p) {
                const int x = F.bucketItems[p];
                if (x >= limit) break; // sorted by index
                if (F.cell[2 * x] == tx && F.cell[2 * x + 1] == tz) cnt += 1;
            }
        }
    return cnt;
}
// per agent k (one wavefront): need[k], its candidates (agents c > k in the 5 x 5 cells) and k's rank among each candidate's passers
__global__ __launch_bounds__(kWave) void sep_cand_kernel(SepFlow F) {
    __shared__ int sCount;
    __shared__ int sList[kSepMaxCand];
    const int n = F.control[0], k = blockIdx.x, lane = laneId();
    if (k >= n) return;
    if (F.retry && F.control[8] == 0) return;
    if (lane == 0) sCount = 0;
    __syncthreads();
    const int cx = F.cell[2 * k], cz = F.cell[2 * k + 1];
    int below = 0;
    const int side = 2 * F.reach + 1; // 5 or 7: at most 49 cells, one per lane
    if (lane < side * side) {
        const int tx = cx + lane % side - F.reach, tz = cz + lane / side - F.reach;
        const unsigned b = sepHash(tx, tz, F.H);
        for (int p = F.bucketStart[b]; p < F.bucketStart[b + 1]; ++p) {
            const int x = F.bucketItems[p];
            if (F.cell[2 * x] != tx || F.cell[2 * x + 1] != tz) continue;
            if (x < k) below += 1;
            else if (x > k) { const int pos = atomicAdd(&sCount, 1); if (pos < kSepMaxCand) sList[pos] = x; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o, kWave);
    __syncthreads();
    const int total = sCount, nc = total < kSepMaxCand ? total : kSepMaxCand;
    if (lane == 0) {
        F.need[k] = below;
        F.candCount[k] = nc;
        if (total > kSepMaxCand) atomicOr(&F.control[2], 1); // more neighbours than a loop tracks: the pass runs serially
    }
    for (int l = lane; l < nc; l += kWave) {
        const int c = sList[l];
        F.cand[(size_t)k * kSepMaxCand + l] = make_int2(c, sepCountBelow(F, F.cell[2 * c], F.cell[2 * c + 1], k));
    }
}

// Can the capsule touch the triangle anywhere along the cast? A lower bound of the distance between the triangle and the parallelogram
// the capsule's axis sweeps (axis box [lo, hi]: for a cast, the box of its four corners; for "any cast of this agent in this pass", the
// agent's whole cached region): the distance to the triangle's plane is linear in the position, so its minimum over a box is at a
// corner, and the box distance to the triangle's box. Every evaluation of sweepCapsuleTriangle (:1303-1322) lies on that axis sweep; if
// the bound exceeds radius + contactEps by 1e-3 (the rounding of either computation is below 1e-4 at these coordinates) no evaluation
// can report contact and the march returns nil whatever steps it takes.
__device__ __forceinline__ bool sepNeverTouches(F3 lo, F3 hi, float radius, F3 v0, F3 v1, F3 v2) {
    const F3 n = normalize(cross(v1 - v0, v2 - v0));
    float dmin = kFloatMax, dmax = -kFloatMax;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const F3 c{(k & 1) ? hi.x : lo.x, (k & 2) ? hi.y : lo.y, (k & 4) ? hi.z : lo.z};
        const float d = dot(n, c - v0);
        dmin = smin(dmin, d); dmax = smax(dmax, d);
    }
    float lb = dmin > 0 ? dmin : (dmax < 0 ? -dmax : 0.0f);
    const F3 mn = vmin(v0, vmin(v1, v2)), mx = vmax(v0, vmax(v1, v2));
    const float dx = smax(0.0f, smax(mn.x - hi.x, lo.x - mx.x));
    const float dy = smax(0.0f, smax(mn.y - hi.y, lo.y - mx.y));
    const float dz = smax(0.0f, smax(mn.z - hi.z, lo.z - mx.z));
    const float lbBox = sqrtf(dx * dx + dy * dy + dz * dz);
    if (lbBox > lb) lb = lbBox;
    return lb - radius - 1e-5f - 1e-3f > 0;
}
// The same bound for ONE cast: its axis sweeps the parallelogram from +- halfHeight * up, + delta, whose corners — not those of its
// bounding box — carry the plane distance's minimum.
__device__ __forceinline__ bool sepCastNeverTouches(F3 from, F3 delta, float halfHeight, float radius, F3 v0, F3 v1, F3 v2) {
    const F3 n = normalize(cross(v1 - v0, v2 - v0));
    const F3 up{0, halfHeight, 0};
    const F3 c0 = from + up, c1 = from - up, c2 = c0 + delta, c3 = c1 + delta;
    const float d0 = dot(n, c0 - v0), d1 = dot(n, c1 - v0), d2 = dot(n, c2 - v0), d3 = dot(n, c3 - v0);
    const float dmin = smin(smin(d0, d1), smin(d2, d3)), dmax = smax(smax(d0, d1), smax(d2, d3));
    float lb = dmin > 0 ? dmin : (dmax < 0 ? -dmax : 0.0f);
    const F3 lo = vmin(c1, c3), hi = vmax(c0, c2); // (up.y >= 0)
    const F3 mn = vmin(v0, vmin(v1, v2)), mx = vmax(v0, vmax(v1, v2));
    const float dx = smax(0.0f, smax(mn.x - hi.x, lo.x - mx.x));
    const float dy = smax(0.0f, smax(mn.y - hi.y, lo.y - mx.y));
    const float dz = smax(0.0f, smax(mn.z - hi.z, lo.z - mx.z));
    const float lbBox = sqrtf(dx * dx + dy * dy + dz * dz);
    if (lbBox > lb) lb = lbBox;
    return lb - radius - 1e-5f - 1e-3f > 0;
}
// Per agent (one wavefront, all agents in parallel, off the pass's critical path): the triangles whose box overlaps the capsule's box at
// the head of the pass grown by kSepTriReach AND that the capsule could touch from somewhere in there (sepNeverTouches over the whole
// region the axis can be in) — what the casts of this agent's pairs can meet unless it is pushed further than that.
// A cast takes exactly the triangles whose AABB overlaps ITS swept box and whose layer passes the mask (groupGather: the traversal
// only prunes, the triangle test decides), so filtering this list by the cast's box gives the work items of a traversal minus
// triangles whose marches return nil, whenever the cast's box lies inside the cached one. An agent whose list is empty (open
// ground: the capsule rests a snap skin above it) casts into nothing at all.
__global__ __launch_bounds__(kWave) void sep_tricache_kernel(SepLaunch K, SepFlow F) {
    const int lane = laneId();
    const int x = blockIdx.x;
    if (x >= F.control[0]) return;
    WaveStats st{0, 0, 0, 0, 0, 0, 0};
    const DevCollision& col = K.col;
    const SepAgentDev& a = K.agents[x];
    const uint32_t mask = K.crowd.params[a.entity].collisionMask;
    const float triReach = F.triReach;
    const float e = a.radius + triReach, ey = a.halfHeight + a.radius + 0.01f;
    const F3 minP{a.position[0] - e, a.position[1] - ey, a.position[2] - e}, maxP{a.position[0] + e, a.position[1] + ey, a.position[2] + e};
    // where the capsule's axis is during any cast that stays inside the cached box
    const F3 axisLo{a.position[0] - triReach, a.position[1] - a.halfHeight - 0.01f, a.position[2] - triReach};
    const F3 axisHi{a.position[0] + triReach, a.position[1] + a.halfHeight + 0.01f, a.position[2] + triReach};
    int count = 0;
    if (col.root >= 0) {
        int stackSize = initTraversal(col), rangeCount = 0, candCount = 0;
        __syncthreads();
        while (stackSize > 0 || rangeCount > 0 || candCount > 0) {
            while ((stackSize > 0 || rangeCount > 0) && candCount < kWave) expandNodes(col, minP, maxP, mask, stackSize, rangeCount, candCount, st);
            const int m = candCount < kWave ? candCount : kWave;
            candCount -= m;
            bool touchable = false;
            float4 t0{0, 0, 0, 0}, t1{0, 0, 0, 0}, t2{0, 0, 0, 0};
            if (lane < m) {
                const float4* tp = reinterpret_cast<const float4*>(col.tris + sh.cand[candCount + lane]);
                t0 = tp[0]; t1 = tp[1]; t2 = tp[2];
                touchable = !sepNeverTouches(axisLo, axisHi, a.radius, F3{t0.x, t0.y, t0.z}, F3{t0.w, t1.x, t1.y}, F3{t1.z, t1.w, t2.x});
            }
            const unsigned long long mt = __ballot(touchable);
            if (count + __popcll(mt) > kSepTriCap) { count = -1; break; }
            if (touchable) {
                float4* dst = F.triCache + ((size_t)x * kSepTriCap + count + prefixCount(mt)) * 3;
                dst[0] = t0; dst[1] = t1; dst[2] = t2;
            }
            count += __popcll(mt);
            __syncthreads();
        }
    }
    const int clear = count == 0 ? 1 : 0;
    if (lane == 0) {
        float* b = F.triBox + (size_t)x * 8;
        b[0] = minP.x; b[1] = minP.y; b[2] = minP.z; b[3] = maxP.x; b[4] = maxP.y; b[5] = maxP.z;
        b[6] = __int_as_float(count); b[7] = __int_as_float(clear);
    }
}

#ifdef SGE_SEP_TIMING
// diagnostics build (tools/separation_timeline.py): per loop of the last pass — ticket, own turn, candidates all passed, end (100-MHz
// chip-wide clock), then the changing pairs, rounds, pairs sent through the BVH casts and sweep trips of the loop
__device__ unsigned long long g_sepTs[8 * 32768];
#define SEP_TS(i, k) do { if (laneId() == 0 && (i) < 32768 && F.trace) g_sepTs[(i) * 8 + (k)] = wall_clock64(); } while (0)
#define SEP_TSV(i, k, v) do { if (laneId() == 0 && (i) < 32768 && F.trace) g_sepTs[(i) * 8 + (k)] = (unsigned long long)(v); } while (0)
extern "C" int sge_experiment_sep_timeline(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sepTs), sizeof(unsigned long long) * 8 * (size_t)(n < 32768 ? n : 32768));
}
#else
#define SEP_TS(i, k) do {} while (0)
#define SEP_TSV(i, k, v) do {} while (0)
#endif
struct SepPairAgent { float radius, halfHeight, invWeight, skinWidth, minGroundDot; uint32_t mask; };
// per-ray set-up of a blocking cast (groupSetupRays), for the short form of a pair's casts in sep_flow2_kernel
struct SepRay { F3 from, delta, dir, mn, mx; float len; int maxIter; bool valid; };
__device__ __forceinline__ SepRay sepRaySetup(const DevCollision& col, F3 from, F3 delta, float radius, float halfHeight) { // groupSetupRays, :1021-1035
    SepRay r;
    r.from = from; r.delta = delta;
    r.len = length(delta);
    r.valid = !(r.len < 1e-6f) && col.root >= 0;
    r.dir = delta / r.len;
    const F3 up{0, 1, 0};
    const F3 a0 = from + up * halfHeight, b0 = from - up * halfHeight;
    const F3 a1 = a0 + delta, b1 = b0 + delta;
    const F3 mn = vmin(vmin(a0, b0), vmin(a1, b1));
    const F3 mx = vmax(vmax(a0, b0), vmax(a1, b1));
    const F3 ext{radius, radius, radius};
    r.mn = mn - ext; r.mx = mx + ext;
    const float minAdv = smax(radius * 0.02f, 1e-4f);
    const int maxIter = (int)ceilf(r.len / minAdv) + 1; // :1296
    r.maxIter = maxIter < 256 ? maxIter : 256;
    return r;
}
// One pair of AgentSeparationResolver.resolve (:1961-2040): `aPos` / `aVel` are loop i's copy of agent i, posI / velI its live entry,
// bPos / bVel agent j's (live). Returns true when agent j's entry changed.
__device__ __forceinline__ bool sepPair(const DevCollision& col, const SepPairAgent& A, const SepPairAgent& B, F3 aPos, F3 aVel, F3& posI, F3& velI,
                                        F3& bPos, F3& bVel, float separationMargin, float heightMargin, WaveStats& st) {
    const float aMin = aPos.y - A.halfHeight, aMax = aPos.y + A.halfHeight;
    const float bMin = bPos.y - B.halfHeight, bMax = bPos.y + B.halfHeight;
    const float ddx = aPos.x - bPos.x, ddz = aPos.z - bPos.z;
    const float distSq = ddx * ddx + ddz * ddz;
    const float skinAllowance = smin(A.skinWidth, B.skinWidth);
    const float margin = smin(separationMargin, skinAllowance);
    const float minDist = A.radius + B.radius + margin;
    const bool heightSeparated = aMax < bMin - heightMargin || aMin > bMax + heightMargin;
    if (heightSeparated) return false;
    if (distSq >= minDist * minDist) return false;
    const float dist = sqrtf(smax(distSq, 1e-8f));
    const float nx = ddx / dist, nz = ddz / dist;
    const float penetration = minDist - dist;
    const float wSum = A.invWeight + B.invWeight;
    if (wSum <= 0) return false;
    const float corr = penetration / wSum;
    F3 moveA{nx * corr * A.invWeight, 0, nz * corr * A.invWeight};
    F3 moveB{-nx * corr * B.invWeight, 0, -nz * corr * B.invWeight};
    const F3 relV = aVel - bVel;
    const float vn = relV.x * nx + relV.z * nz;
    if (vn < 0) {
        const float impulse = -vn;
        const float scaleA = A.invWeight / wSum, scaleB = B.invWeight / wSum;
        velI.x += nx * impulse * scaleA; velI.z += nz * impulse * scaleA;
        bVel.x -= nx * impulse * scaleB; bVel.z -= nz * impulse * scaleB;
    }
    // the two blocking casts of :2004-2027 as one pass: ray 0 = agent i along moveA, ray 1 = agent j along moveB
    const float eps = 1e-6f;
    const bool castA = length(moveA) > eps, castB = length(moveB) > eps;
    bool blockedA = false, blockedB = false;
    if (castA || castB) {
        __syncthreads();
        sh.rayFrom[0] = posI; sh.rayDelta[0] = castA ? moveA : F3{0, 0, 0};
        sh.rayFrom[1] = bPos; sh.rayDelta[1] = castB ? moveB : F3{0, 0, 0};
        __syncthreads();
        int itemCount = 0;
        F3 minP, maxP;
        if (groupSetupRays(col, 0, 1, A.radius, A.halfHeight, true, false, 0.0f, minP, maxP) > 0)
            groupGather(col, 0, 1, minP, maxP, A.radius, false, 0.0f, A.mask, itemCount, st);
        if (groupSetupRays(col, 1, 1, B.radius, B.halfHeight, true, false, 0.0f, minP, maxP) > 0)
            groupGather(col, 1, 1, minP, maxP, B.radius, false, 0.0f, B.mask, itemCount, st);
        groupSweep(col, itemCount, st);
        blockedA = castA && rayHit(0) && sh.rayRec[0].toi <= A.skinWidth && sh.rayRec[0].normal.y < A.minGroundDot;
        blockedB = castB && rayHit(1) && sh.rayRec[1].toi <= B.skinWidth && sh.rayRec[1].normal.y < B.minGroundDot;
        __syncthreads();
    }
    if (blockedA && !blockedB) {
        moveA = F3{0, 0, 0};
        moveB = F3{-nx * penetration, 0, -nz * penetration};
    } else if (blockedB && !blockedA) {
        moveB = F3{0, 0, 0};
        moveA = F3{nx * penetration, 0, nz * penetration};
    } else if (blockedA && blockedB) {
        return vn < 0; // (:2035 `continue` comes after the velocity update)
    }
    posI = posI + moveA;
    bPos = bPos + moveB;
    return true;
}
__device__ __forceinline__ SepPairAgent sepPairAgent(const SepLaunch& K, const SepAgentDev& a) {
    const sge_controller_params& P = K.crowd.params[a.entity];
    return SepPairAgent{a.radius, a.halfHeight, a.invWeight, P.skinWidth, P.minGroundDot, P.collisionMask};
}

// the pair loops of one pass, one wavefront per loop, loops drawn in index order
// does the pair change anything? The early exits of :1972-1986 (sepPair repeats them, same expressions)
__device__ __forceinline__ bool sepInteracts(const SepPairAgent& A, const SepPairAgent& B, F3 aPos, F3 bPos, float separationMargin, float heightMargin) {
    const float aMin = aPos.y - A.halfHeight, aMax = aPos.y + A.halfHeight;
    const float bMin = bPos.y - B.halfHeight, bMax = bPos.y + B.halfHeight;
    const float ddx = aPos.x - bPos.x, ddz = aPos.z - bPos.z;
    const float distSq = ddx * ddx + ddz * ddz;
    const float skinAllowance = smin(A.skinWidth, B.skinWidth);
    const float margin = smin(separationMargin, skinAllowance);
    const float minDist = A.radius + B.radius + margin;
    const bool heightSeparated = aMax < bMin - heightMargin || aMin > bMax + heightMargin;
    if (heightSeparated) return false;
    if (distSq >= minDist * minDist) return false;
    return A.invWeight + B.invWeight > 0;
}

// the pair loops of one pass, one wavefront per loop, loops drawn in index order
__global__ __launch_bounds__(kWave, 2) void sep_flow_kernel(SepLaunch K, SepFlow F) {
    __shared__ int2 sCand[kSepMaxCand];
    __shared__ unsigned long long sKey[kSepMaxCand]; // pairs that change something, by their place in the reference's order (~0: none)
    __shared__ int sTicket;
    const int lane = laneId();
    const DevCollision& col = K.col;
    WaveStats st{0, 0, 0, 0, 0, 0, 0};
    const int n = F.control[0];
    if (n <= 1) return;
    if (__hip_atomic_load(&F.control[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) return; // the candidate lists overflowed: serial pass
    const float cellSize = __int_as_float(F.control[3]);
    while (true) {
        __syncthreads();
        if (lane == 0) sTicket = atomicAdd(&F.control[1], 1);
        __syncthreads();
        const int i = sTicket;
        if (i >= n) break;
        const int nc = F.candCount[i];
        for (int l = lane; l < nc; l += kWave) { sCand[l] = F.cand[(size_t)i * kSepMaxCand + l]; sKey[l] = ~0ull; }
        __syncthreads();
        sepWait(&F.ver[i], F.need[i]); // every earlier loop that can pair with agent i has passed it
        SepAgentDev* Ai = K.agents + i;
        const F3 aPos = sepLoad3(Ai->position), aVel = sepLoad3(Ai->velocity); // the copy `a` (:1954)
        const SepPairAgent A = sepPairAgent(K, *Ai);
        F3 posI = aPos, velI = aVel;
        const int cx = (int)floorf(aPos.x / cellSize), cz = (int)floorf(aPos.z / cellSize);
        {   // the pair list of :1955-1960 hangs on the LIVE cell: its 3 x 3 cells lie inside the 5 x 5 the candidates were taken from
            // as long as the agent has not been pushed further than one cell since the head of the pass
            const int d0 = cx - F.cell[2 * i], d1 = cz - F.cell[2 * i + 1], slack = F.reach - 1;
            if ((d0 < -slack || d0 > slack || d1 < -slack || d1 > slack) && lane == 0) atomicOr(&F.control[2], 2); // redo serially
            if ((d0 < -1 || d0 > 1 || d1 < -1 || d1 > 1) && lane == 0) atomicOr(&F.control[5], 2);               // (what the rule's reach would have said)
        }
        // 1. every candidate in its turn, a lane each: the ones outside the 3 x 3 cells around the live cell, and the pairs that fail
        //    the tests of :1972-1986 (they read agent j at their turn and change nothing: they commute with everything), are passed
        //    at once; the pairs that DO change something keep their turn and are queued by their place in the reference's order
        //    (cell dz, dx, then index). A loop no longer makes its successors wait for ~30 passes one after the other.
        for (int base = 0; base < nc; base += kWave) {
            const int l = base + lane;
            bool waiting = l < nc;
            const int c = waiting ? sCand[l].x : 0, rank = waiting ? sCand[l].y : 0;
            const int ddx = waiting ? F.cell[2 * c] - cx : 9, ddz = waiting ? F.cell[2 * c + 1] - cz : 9;
            const bool inList = ddx >= -1 && ddx <= 1 && ddz >= -1 && ddz <= 1;
            while (__any(waiting)) {
                if (waiting && __hip_atomic_load(&F.ver[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == rank) {
                    waiting = false;
                    bool hold = false;
                    if (inList) {
                        const SepAgentDev* Aj = K.agents + c;
                        const F3 bPos = sepLoad3(Aj->position);
                        hold = sepInteracts(A, sepPairAgent(K, *Aj), aPos, bPos, K.separationMargin, K.heightMargin);
                    }
                    if (hold) sKey[l] = ((unsigned long long)((ddz + 1) * 3 + (ddx + 1)) << 32) | (unsigned)c;
                    else __hip_atomic_store(&F.ver[c], rank + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (__any(waiting)) __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        // 2. the pairs that change something, one after the other in the reference's order (the casts start from agent i's live position)
        while (true) {
            unsigned long long best = ~0ull;
            int bestSlot = -1;
            for (int base = 0; base < nc; base += kWave) {
                const int l = base + lane;
                const unsigned long long k = l < nc ? sKey[l] : ~0ull;
                const unsigned long long m = waveMinU64(k);
                if (m < best) { best = m; const unsigned long long who = __ballot(k == m); bestSlot = base + __ffsll((long long)who) - 1; }
            }
            if (bestSlot < 0) break;
            const int j = sCand[bestSlot].x, rank = sCand[bestSlot].y;
            SepAgentDev* Aj = K.agents + j;
            F3 bPos = sepLoad3(Aj->position), bVel = sepLoad3(Aj->velocity);
            const SepPairAgent B = sepPairAgent(K, *Aj);
            if (sepPair(col, A, B, aPos, aVel, posI, velI, bPos, bVel, K.separationMargin, K.heightMargin, st)) {
                if (lane == 0) { sepStore3(Aj->position, bPos); sepStore3(Aj->velocity, bVel); }
            }
            __syncthreads();
            if (lane == 0) { sepRelease(&F.ver[j], rank + 1); sKey[bestSlot] = ~0ull; }
            __syncthreads();
        }
        if (lane == 0) { sepStore3(Ai->position, posI); sepStore3(Ai->velocity, velI); } // nobody reads it before the pass ends
    }
}

// ---- the pair loops, second form (round 4) ---------------------------------------------------------------------------------------
// In-kernel stamps of the first form showed a changing pair costing ~25,000 cycles (~12 us), about half of them six dependent round
// trips to the memory side (the agent's record behind its counter, its controller parameters behind its entity index, two BVH
// traversals, the fence in front of the counter store) and the rest distance evaluations, mostly of triangles the capsule cannot
// reach — and a per-loop timeline (tools/separation_timeline.py) that only ~3 loops are inside their pairs at any time: the stage's
// time IS the pairs' latency. This form takes the round trips out of the chain and most of the evaluations out of the casts:
//  - what a pair changes of an agent (position x, z and velocity x, z: the resolver's moves and impulses have no y) travels as four
//    8-byte granules (version << 32 | float bits), stored by ONE 8-byte agent-scope atomic each: a reader that finds its rank in all
//    four tags has the values — no counter behind a fence, no load behind a poll (cdna_hip_programming.md Guideline 16, R2);
//  - what no pair changes (capsule, weight, controller parameters, y) is one 32-byte record per agent, and a candidate's lane loads
//    it — and the box of the candidate's cached triangles — before the candidate's turn comes;
//  - the pairs that change something are kept in LDS with everything a pair reads and with their push and impulse, which follow from
//    loop i's copy of agent i and agent j alone (:1961-2003) and are worked out by the candidate's lane at its turn; only the origin
//    of agent i's cast is its live position, i.e. the sum of the pushes before it — predicted as if no earlier pair were blocked;
//  - per agent and pass, sep_tricache_kernel lists the triangles the capsule could touch from anywhere within kSepTriReach of where it
//    stands (sepNeverTouches: a lower bound of the distance from the triangle's plane and box to the whole region the axis can be
//    in, 1e-3 of margin against rounding). On open ground that list is empty — the capsule rests a snap skin above the ground —
//    and a cast that stays in the region hits nothing: a loop whose agents are all like that is a few sums (step 2a, ~0.9 us
//    against ~12 us per pair);
//  - the other loops go through their casts in ROUNDS (2b): all rays of a round take their work items from the cached triangles
//    (filtered once more against the cast's own sweep) and are swept together, one lane per item, idle lanes bisecting ahead for
//    the items in refineTOI; the pairs are then resolved in the reference's order, and a pair that IS blocked ends the round behind
//    it (its successors' origins were predicted wrong: they go into the next round).
// A pair whose casts do not fit (more than kSepTriCap triangles in reach, cast outside the cached box, more items than lanes) goes
// through sepPair's BVH casts as before. Same results, bit for bit (tests/test_separation.py: against the oracle and against the
// first form). 8,192 agents on the cheese + mirror scene: 79 -> 34 ms per step on the same box, 31,250: 446 -> 194
// (profiles/r4_separation_bench.txt); what is left is real sweeps — a fifth of the loops stand on triangles they touch, ~5 trips
// of ~3 us each — and the order itself: ~3 loops are inside their pairs at any time.
constexpr int kSepHeldCap = 64, kSepRound = 8;
constexpr unsigned kSepSpinLimit = 1u << 22; // polls of one loop before it gives the pass up (a poll is ~0.5 us)
constexpr int kSepPollSleep = 8; // x 64 cycles between two polls of a waiting loop (none / 2 / 8 / 32 / 127: 34.0 / 33.4 / 32.8 / 33.3 / 39.3 ms per step)
enum { SEP_LIVE = 1, SEP_CAST_A = 2, SEP_CAST_B = 4, SEP_VN_NEG = 8, SEP_B_CERTAIN = 16 };
struct SepHeld {
    unsigned long long key; int c, rank; float bx, bz, bvx, bvz; SepStaticDev S; float box[6]; int triCount, clear;
    // the pair's push and impulse (:1961-2003) — functions of loop i's copy of agent i and of agent j alone, computed by the candidate's
    // lane at its turn — and whether agent j's cast is known to hit nothing (SEP_B_CERTAIN)
    float moveAx, moveAz, moveBx, moveBz, nx, nz, penetration, dvIx, dvIz, nbvx, nbvz; int flags;
};
struct SepRayRec { F3 from, delta, dir, mn, mx; float len, radius, halfHeight, ny; int maxIter, valid, src, skip, cnt, pad; unsigned long long key; };
// the swept box of a cast (groupSetupRays, :1021-1035) and whether it lies inside an agent's cached box
__device__ __forceinline__ bool sepCastInside(F3 from, F3 delta, float radius, float halfHeight, const float* box) {
    const F3 up{0, 1, 0};
    const F3 a0 = from + up * halfHeight, b0 = from - up * halfHeight;
    const F3 a1 = a0 + delta, b1 = b0 + delta;
    const F3 ext{radius, radius, radius};
    const F3 mn = vmin(vmin(a0, b0), vmin(a1, b1)) - ext, mx = vmax(vmax(a0, b0), vmax(a1, b1)) + ext;
    return mn.x >= box[0] && mn.y >= box[1] && mn.z >= box[2] && mx.x <= box[3] && mx.y <= box[4] && mx.z <= box[5];
}
__device__ __forceinline__ unsigned long long sepLoadG(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sepStoreG(unsigned long long* p, int tag, float v) {
    __hip_atomic_store(p, ((unsigned long long)(unsigned)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The workgroup is ONE wavefront: its LDS accesses execute in program order, so all a hand-over through LDS needs is that the
// compiler keeps that order and that the writes have been issued — not __syncthreads(), whose s_waitcnt vmcnt(0) would also wait
// for the acknowledgement of every granule store in flight (a round trip to the memory side per round).
#define SEP_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
__global__ __launch_bounds__(kWave) void sep_flow2_kernel(SepLaunch K, SepFlow F) {
    __shared__ int2 sCand[kSepMaxCand];
    __shared__ SepHeld sHeld[kSepHeldCap];
    __shared__ int sOrder[kSepHeldCap];
    __shared__ SepRayRec sRay[2 * kSepRound];
    __shared__ float4 sItemTri[kWave * 3];
    __shared__ float4 sOwnTri[kSepTriCap * 3];
    __shared__ int sItemRay[kWave];
    __shared__ int sTicket, sHeldCount;
    const int lane = laneId();
    const DevCollision& col = K.col;
    WaveStats st{0, 0, 0, 0, 0, 0, 0};
    const int n = F.control[0];
    if (n <= 1) return;
    if (__hip_atomic_load(&F.control[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) return; // the candidate lists overflowed: serial pass
    if (F.retry && F.control[8] == 0) return; // the pass's first attempt stood
    const float cellSize = __int_as_float(F.control[3]);
    const bool sepNoDefer = F.noDefer != 0;
    // Every wait in here is for a lower loop, and the lowest unfinished loop waits for nobody: the waits end. They are bounded all the
    // same (a wavefront that spun forever would take the card with it): after kSepSpinLimit polls — seconds, where the longest
    // legitimate wait is a pass, i.e. tens of milliseconds — a loop raises bit 2 of the pass's redo flags, every other loop sees it
    // within 1,024 polls, the kernel drains, and the pass is redone by the serial kernel from the saved state.
    unsigned spins = 0;
    auto giveUp = [&]() {
        spins += 1;
        if ((spins & 1023u) != 0) return false;
        const bool over = spins > kSepSpinLimit || (__hip_atomic_load(&F.control[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4);
        if (over && lane == 0) atomicOr(&F.control[2], 4);
        return over;
    };
    while (true) {
        SEP_SYNC();
        if (lane == 0) { sTicket = atomicAdd(&F.control[1], 1); sHeldCount = 0; }
        SEP_SYNC();
        const int i = sTicket;
        if (i >= n) break;
        SEP_TS(i, 0);
        const int nc = F.candCount[i];
        for (int l = lane; l < nc; l += kWave) sCand[l] = F.cand[(size_t)i * kSepMaxCand + l];
        // agent i's own unchanging part and its cached triangles, before its turn comes
        const SepStaticDev SA = F.stat[i];
        const SepPairAgent A{SA.radius, SA.halfHeight, SA.invWeight, SA.skinWidth, SA.minGroundDot, SA.mask};
        float boxI[6];
        for (int k = 0; k < 6; ++k) boxI[k] = F.triBox[(size_t)i * 8 + k];
        const int triCountI = __float_as_int(F.triBox[(size_t)i * 8 + 6]), clearI = __float_as_int(F.triBox[(size_t)i * 8 + 7]);
        for (int k = lane; k < triCountI; k += kWave) {
            const float4* tp = F.triCache + ((size_t)i * kSepTriCap + k) * 3;
            sOwnTri[k * 3] = tp[0]; sOwnTri[k * 3 + 1] = tp[1]; sOwnTri[k * 3 + 2] = tp[2];
        }
        const int needI = F.need[i];
        SEP_SYNC();
        // every earlier loop that can pair with agent i has passed it: its four granules carry need[i]
        unsigned long long g0, g1, g2, g3;
        while (true) {
            const unsigned long long* lp = F.live + (size_t)i * 4;
            g0 = sepLoadG(lp); g1 = sepLoadG(lp + 1); g2 = sepLoadG(lp + 2); g3 = sepLoadG(lp + 3);
            if ((int)(g0 >> 32) == needI && (int)(g1 >> 32) == needI && (int)(g2 >> 32) == needI && (int)(g3 >> 32) == needI) break;
            if (giveUp()) return;
            __builtin_amdgcn_s_sleep(kSepPollSleep);
        }
        SEP_TS(i, 1);
        const F3 aPos{__uint_as_float((unsigned)g0), SA.posY, __uint_as_float((unsigned)g1)};  // the copy `a` (:1954)
        const F3 aVel{__uint_as_float((unsigned)g2), SA.velY, __uint_as_float((unsigned)g3)};
        F3 posI = aPos, velI = aVel;
        const int cx = (int)floorf(aPos.x / cellSize), cz = (int)floorf(aPos.z / cellSize);
        {
            const int d0 = cx - F.cell[2 * i], d1 = cz - F.cell[2 * i + 1], slack = F.reach - 1;
            if ((d0 < -slack || d0 > slack || d1 < -slack || d1 > slack) && lane == 0) atomicOr(&F.control[2], 2); // redo serially
            if ((d0 < -1 || d0 > 1 || d1 < -1 || d1 > 1) && lane == 0) atomicOr(&F.control[5], 2);               // (what the rule's reach would have said)
        }
        // 1. every candidate in its turn, a lane each (see sep_flow_kernel): passed at once unless the pair changes something; a pair
        //    that does is kept in LDS with its push and impulse already worked out by that lane.
        //    Only the candidates in the 3 x 3 cells around the live cell can become pairs; the outer ring is passed for the order's sake
        //    alone, and the loop does not wait for it: whoever of the ring is not yet due when the inner ones of its chunk have all
        //    arrived goes back into the list's head (the chunk's own entries are in registers by then) and is passed between the rounds
        //    and behind the pairs — nothing the loop does depends on it, and the ring's own loops see the pass whenever it comes.
        //    Two engines in lockstep in one process (tools/separation_ab.py), 8,192 agents: 48.1 -> 34.8 ms per step.
        const bool mayDefer = !sepNoDefer;
        int deferred = 0; // entries sCand[0 .. deferred) wait to be passed (x < 0: done)
        auto passDeferred = [&]() {
            bool pending = false;
            for (int d0 = 0; d0 < deferred; d0 += kWave) {
                const int d = d0 + lane;
                const int2 e = d < deferred ? sCand[d] : make_int2(-1, 0);
                if (e.x >= 0) {
                    unsigned long long* lp = F.live + (size_t)e.x * 4;
                    const unsigned long long h0 = sepLoadG(lp), h1 = sepLoadG(lp + 1), h2 = sepLoadG(lp + 2), h3 = sepLoadG(lp + 3);
                    if ((int)(h0 >> 32) == e.y && (int)(h1 >> 32) == e.y && (int)(h2 >> 32) == e.y && (int)(h3 >> 32) == e.y) {
                        sepStoreG(lp, e.y + 1, __uint_as_float((unsigned)h0)); sepStoreG(lp + 1, e.y + 1, __uint_as_float((unsigned)h1));
                        sepStoreG(lp + 2, e.y + 1, __uint_as_float((unsigned)h2)); sepStoreG(lp + 3, e.y + 1, __uint_as_float((unsigned)h3));
                        sCand[d].x = -1;
                    } else pending = true;
                }
            }
            return __any(pending);
        };
        for (int base = 0; base < nc; base += kWave) {
            const int l = base + lane;
            bool waiting = l < nc;
            const int c = waiting ? sCand[l].x : 0, rank = waiting ? sCand[l].y : 0;
            const int ddx = waiting ? F.cell[2 * c] - cx : 9, ddz = waiting ? F.cell[2 * c + 1] - cz : 9;
            const bool inList = ddx >= -1 && ddx <= 1 && ddz >= -1 && ddz <= 1;
            SepStaticDev S = SA;
            float box[6] = {0, 0, 0, 0, 0, 0};
            int triCount = -1, clear = 0;
            if (waiting && inList) {
                S = F.stat[c];
                for (int k = 0; k < 6; ++k) box[k] = F.triBox[(size_t)c * 8 + k];
                triCount = __float_as_int(F.triBox[(size_t)c * 8 + 6]);
                clear = __float_as_int(F.triBox[(size_t)c * 8 + 7]);
            }
            unsigned long long* lp = F.live + (size_t)c * 4;
            while (__any(waiting && (inList || !mayDefer))) {
                if (waiting) {
                    const unsigned long long h0 = sepLoadG(lp), h1 = sepLoadG(lp + 1), h2 = sepLoadG(lp + 2), h3 = sepLoadG(lp + 3);
                    if ((int)(h0 >> 32) == rank && (int)(h1 >> 32) == rank && (int)(h2 >> 32) == rank && (int)(h3 >> 32) == rank) {
                        waiting = false;
                        const float bx = __uint_as_float((unsigned)h0), bz = __uint_as_float((unsigned)h1);
                        const float bvx = __uint_as_float((unsigned)h2), bvz = __uint_as_float((unsigned)h3);
                        bool hold = false;
                        if (inList) {
                            const SepPairAgent B{S.radius, S.halfHeight, S.invWeight, S.skinWidth, S.minGroundDot, S.mask};
                            hold = sepInteracts(A, B, aPos, F3{bx, S.posY, bz}, K.separationMargin, K.heightMargin);
                        }
                        int slot = -1;
                        if (hold) slot = atomicAdd(&sHeldCount, 1);
                        if (hold && slot < kSepHeldCap) {
                            SepHeld& H = sHeld[slot];
                            H.key = ((unsigned long long)((ddz + 1) * 3 + (ddx + 1)) << 32) | (unsigned)c;
                            H.c = c; H.rank = rank; H.bx = bx; H.bz = bz; H.bvx = bvx; H.bvz = bvz; H.S = S;
                            for (int k = 0; k < 6; ++k) H.box[k] = box[k];
                            H.triCount = triCount; H.clear = clear;
                            // the pair's push, impulse and cast flags (:1961-2003; sepPair's expressions)
                            int flags = 0;
                            float moveAx = 0, moveAz = 0, moveBx = 0, moveBz = 0, nx = 0, nz = 0, penetration = 0, dvIx = 0, dvIz = 0, nbvx = bvx, nbvz = bvz;
                            const float pdx = aPos.x - bx, pdz = aPos.z - bz;
                            const float distSq = pdx * pdx + pdz * pdz;
                            const float skinAllowance = smin(A.skinWidth, S.skinWidth);
                            const float margin = smin(K.separationMargin, skinAllowance);
                            const float minDist = A.radius + S.radius + margin;
                            const float wSum = A.invWeight + S.invWeight;
                            if (!(distSq >= minDist * minDist) && !(wSum <= 0)) { // (the height test is part of `hold`)
                                const float dist = sqrtf(smax(distSq, 1e-8f));
                                nx = pdx / dist; nz = pdz / dist;
                                penetration = minDist - dist;
                                const float corr = penetration / wSum;
                                flags = SEP_LIVE;
                                moveAx = nx * corr * A.invWeight; moveAz = nz * corr * A.invWeight;
                                moveBx = -nx * corr * S.invWeight; moveBz = -nz * corr * S.invWeight;
                                const F3 relV = aVel - F3{bvx, S.velY, bvz};
                                const float vn = relV.x * nx + relV.z * nz;
                                if (vn < 0) {
                                    const float impulse = -vn;
                                    const float scaleA = A.invWeight / wSum, scaleB = S.invWeight / wSum;
                                    dvIx = nx * impulse * scaleA; dvIz = nz * impulse * scaleA;
                                    nbvx = bvx - nx * impulse * scaleB; nbvz = bvz - nz * impulse * scaleB;
                                    flags |= SEP_VN_NEG;
                                }
                                const float eps = 1e-6f;
                                if (length(F3{moveAx, 0, moveAz}) > eps) flags |= SEP_CAST_A;
                                if (length(F3{moveBx, 0, moveBz}) > eps) flags |= SEP_CAST_B;
                                // agent j's cast: nothing in reach of agent j, and the cast stays where that was established
                                if (!(flags & SEP_CAST_B) || (clear && !(length(F3{moveBx, 0, moveBz}) < 1e-6f) &&
                                                              sepCastInside(F3{bx, S.posY, bz}, F3{moveBx, 0, moveBz}, S.radius, S.halfHeight, box))) flags |= SEP_B_CERTAIN;
                            }
                            H.moveAx = moveAx; H.moveAz = moveAz; H.moveBx = moveBx; H.moveBz = moveBz; H.nx = nx; H.nz = nz; H.penetration = penetration;
                            H.dvIx = dvIx; H.dvIz = dvIz; H.nbvx = nbvx; H.nbvz = nbvz; H.flags = flags;
                        } else {
                            if (hold) atomicOr(&F.control[2], 8); // more changing pairs than this form keeps: the pass runs serially
                            sepStoreG(lp, rank + 1, bx); sepStoreG(lp + 1, rank + 1, bz); sepStoreG(lp + 2, rank + 1, bvx); sepStoreG(lp + 3, rank + 1, bvz);
                        }
                    }
                }
                if (__any(waiting && (inList || !mayDefer))) { if (giveUp()) return; __builtin_amdgcn_s_sleep(kSepPollSleep); }
            }
            if (mayDefer) { // (the chunk's entries are in registers: the list's head is free up to base + kWave)
                const unsigned long long late = __ballot(waiting);
                if (waiting) sCand[deferred + prefixCount(late)] = make_int2(c, rank);
                deferred += __popcll(late);
                SEP_SYNC();
            }
        }
        SEP_SYNC();
        SEP_TS(i, 2);
        const int nh = sHeldCount < kSepHeldCap ? sHeldCount : kSepHeldCap;
        // the reference's order of the pairs that change something (cell dz, dx, then index)
        if (lane < nh) {
            const unsigned long long mine = sHeld[lane].key;
            int before = 0;
            for (int k = 0; k < nh; ++k) before += sHeld[k].key < mine ? 1 : 0;
            sOrder[before] = lane;
        }
        SEP_SYNC();
#ifdef SGE_SEP_TIMING
        int dbgRounds = 0, dbgFall = 0, dbgTrips = 0, dbgMarch = 0, dbgCrawl = 0, dbgContact = 0;
#endif
        int h0 = 0;
        // 2a. the common case, without a cast: lane g takes pair g. Agent i's live position at pair g is the sum of the pushes before it
        //     (as long as none of them is blocked); if nothing is in reach of agent i there and of no agent j, no cast can hit, no pair
        //     is blocked, and the loop is those sums.
        if (nh > 0) {
            int flags = 0, c = 0, rank = 0;
            float mAx = 0, mAz = 0, dvx = 0, dvz = 0, nbx = 0, nbz = 0, nbvx = 0, nbvz = 0;
            if (lane < nh) {
                const SepHeld& H = sHeld[sOrder[lane]];
                flags = H.flags; c = H.c; rank = H.rank; mAx = H.moveAx; mAz = H.moveAz; dvx = H.dvIx; dvz = H.dvIz; nbvx = H.nbvx; nbvz = H.nbvz;
                nbx = H.bx; nbz = H.bz;
                if (flags & SEP_LIVE) { const F3 nb = F3{H.bx, H.S.posY, H.bz} + F3{H.moveBx, 0, H.moveBz}; nbx = nb.x; nbz = nb.z; }
            }
            F3 pp = posI, vv = velI, myFrom = posI;
            for (int k = 0; k < nh; ++k) {
                const int fk = __shfl(flags, k, kWave);
                const float ax = __shfl(mAx, k, kWave), az = __shfl(mAz, k, kWave), vx = __shfl(dvx, k, kWave), vz = __shfl(dvz, k, kWave);
                if (lane == k) myFrom = pp;
                if (fk & SEP_VN_NEG) { vv.x += vx; vv.z += vz; }
                if (fk & SEP_LIVE) pp = pp + F3{ax, 0, az};
            }
            bool ok = true;
            if (lane < nh && (flags & SEP_LIVE)) {
                if (!(flags & SEP_B_CERTAIN)) ok = false;
                if ((flags & SEP_CAST_A) && !(clearI && sepCastInside(myFrom, F3{mAx, 0, mAz}, A.radius, A.halfHeight, boxI))) ok = false;
            }
            if (__all(ok)) {
                if (lane < nh) {
                    unsigned long long* lp = F.live + (size_t)c * 4;
                    sepStoreG(lp, rank + 1, nbx); sepStoreG(lp + 1, rank + 1, nbz); sepStoreG(lp + 2, rank + 1, nbvx); sepStoreG(lp + 3, rank + 1, nbvz);
                }
                posI = pp; velI = vv;
                h0 = nh;
            }
        }
        // 2b. otherwise the pairs go through their casts in rounds
        while (h0 < nh) {
#ifdef SGE_SEP_TIMING
            dbgRounds += 1;
#endif
            if (deferred) (void)passDeferred();
            int G = nh - h0 < kSepRound ? nh - h0 : kSepRound;
            // (b) the rays: 2 g = agent i from its predicted live position along moveA, 2 g + 1 = agent j along moveB
            bool rayOk = true;
            if (lane < 2 * G) {
                const int g = lane >> 1, which = lane & 1;
                const SepHeld& H = sHeld[sOrder[h0 + g]];
                F3 from, delta;
                float radius, halfHeight;
                bool cast;
                const float* box;
                int triCount, clear;
                if (which == 0) {
                    F3 pp = posI;
                    for (int k = 0; k < g; ++k) { const SepHeld& Hk = sHeld[sOrder[h0 + k]]; if (Hk.flags & SEP_LIVE) pp = pp + F3{Hk.moveAx, 0, Hk.moveAz}; } // as if no earlier pair of the round were blocked
                    from = pp; delta = F3{H.moveAx, 0, H.moveAz}; radius = A.radius; halfHeight = A.halfHeight; cast = (H.flags & SEP_CAST_A) != 0;
                    box = boxI; triCount = triCountI; clear = clearI;
                } else {
                    from = F3{H.bx, H.S.posY, H.bz}; delta = F3{H.moveBx, 0, H.moveBz}; radius = H.S.radius; halfHeight = H.S.halfHeight; cast = (H.flags & SEP_CAST_B) != 0;
                    box = H.box; triCount = H.triCount; clear = H.clear;
                }
                const SepRay r = sepRaySetup(col, from, cast ? delta : F3{0, 0, 0}, radius, halfHeight);
                SepRayRec& R = sRay[lane];
                R.from = r.from; R.delta = r.delta; R.dir = r.dir; R.mn = r.mn; R.mx = r.mx; R.len = r.len; R.radius = radius; R.halfHeight = halfHeight;
                R.ny = 0; R.maxIter = r.maxIter; R.valid = r.valid ? 1 : 0; R.src = which ? H.c : -1; R.key = ~0ull;
                R.skip = clear; // (inside the cached box, checked next:) nothing in reach of this agent — the cast hits nothing
                R.cnt = triCount;
                rayOk = !r.valid || (triCount >= 0 && r.mn.x >= box[0] && r.mn.y >= box[1] && r.mn.z >= box[2] &&
                                     r.mx.x <= box[3] && r.mx.y <= box[4] && r.mx.z <= box[5]);
            }
            unsigned needRays = 0;
            {
                const unsigned long long bad = __ballot(!rayOk);
                if (bad) { const int gBad = (__ffsll((long long)bad) - 1) >> 1; if (gBad < G) G = gBad; }
                SEP_SYNC();
                needRays = (unsigned)__ballot(lane < 2 * G && sRay[lane < 2 * kSepRound ? lane : 0].valid && !sRay[lane < 2 * kSepRound ? lane : 0].skip);
            }
            // (c) work items of the round's rays from the cached triangles: two rays of at most 32 cached triangles share the wavefront,
            //     a longer list has it alone, 64 triangles at a time; a round ends in front of the pair whose items no longer fit one per lane
            int total = 0;
            {
                int curPair = -1, pairStart = 0;
                bool cut = false;
                while (needRays && !cut) {
                    const int rA = __ffs((int)needRays) - 1; needRays &= needRays - 1;
                    int rB = needRays ? __ffs((int)needRays) - 1 : -1;
                    const int cntA = sRay[rA].cnt;
                    const bool wide = cntA > 32 || (rB >= 0 && sRay[rB].cnt > 32);
                    if (wide) rB = -1;
                    if (rB >= 0) needRays &= needRays - 1;
                    for (int k0 = 0; k0 < (wide ? cntA : 1) && !cut; k0 += kWave) {
                        const int half = wide ? 0 : lane >> 5, k = wide ? k0 + lane : lane & 31;
                        const int r = half ? rB : rA;
                        bool c = false;
                        float4 t0{0, 0, 0, 0}, t1{0, 0, 0, 0}, t2{0, 0, 0, 0};
                        if (r >= 0) {
                            const SepRayRec& R = sRay[r];
                            if (k < R.cnt) {
                                if (R.src >= 0) { const float4* tp = F.triCache + ((size_t)R.src * kSepTriCap + k) * 3; t0 = tp[0]; t1 = tp[1]; t2 = tp[2]; }
                                else { t0 = sOwnTri[k * 3]; t1 = sOwnTri[k * 3 + 1]; t2 = sOwnTri[k * 3 + 2]; }
                                const F3 v0{t0.x, t0.y, t0.z}, v1{t0.w, t1.x, t1.y}, v2{t1.z, t1.w, t2.x};
                                c = !boxDisjoint(vmin(v0, vmin(v1, v2)), vmax(v0, vmax(v1, v2)), R.mn, R.mx) &&
                                    !sepCastNeverTouches(R.from, R.delta, R.halfHeight, R.radius, v0, v1, v2);
                            }
                        }
                        const unsigned long long m = __ballot(c);
                        const unsigned long long maskA = wide ? ~0ull : 0xffffffffull;
                        const int nA = __popcll(m & maskA), nB = __popcll(m & ~maskA);
                        // ray A, then ray B, in ray order: the items of one pair are contiguous
                        for (int hh = 0; hh < 2; ++hh) {
                            const int rr = hh ? rB : rA, cnt = hh ? nB : nA;
                            if (rr < 0) break;
                            const int g = rr >> 1;
                            if (g != curPair) { curPair = g; pairStart = total; }
                            if (total + cnt > kWave) { G = g; total = pairStart; cut = true; break; }
                            if (c && half == hh) {
                                const int at = total + prefixCount(hh ? m & ~maskA : m & maskA);
                                sItemTri[at * 3] = t0; sItemTri[at * 3 + 1] = t1; sItemTri[at * 3 + 2] = t2; sItemRay[at] = rr;
                            }
                            total += cnt;
                        }
                    }
                }
            }
            SEP_SYNC();
            if (G == 0) {
                // the pair at the head does not fit the short form: its casts go through the BVH (sepPair), alone
#ifdef SGE_SEP_TIMING
                dbgFall += 1;
#endif
                const SepHeld& H = sHeld[sOrder[h0]];
                const SepPairAgent B{H.S.radius, H.S.halfHeight, H.S.invWeight, H.S.skinWidth, H.S.minGroundDot, H.S.mask};
                F3 bPos{H.bx, H.S.posY, H.bz}, bVel{H.bvx, H.S.velY, H.bvz};
                const int j = H.c, rank = H.rank;
                sepPair(col, A, B, aPos, aVel, posI, velI, bPos, bVel, K.separationMargin, K.heightMargin, st);
                SEP_SYNC();
                if (lane == 0) {
                    unsigned long long* lp = F.live + (size_t)j * 4;
                    sepStoreG(lp, rank + 1, bPos.x); sepStoreG(lp + 1, rank + 1, bPos.z); sepStoreG(lp + 2, rank + 1, bVel.x); sepStoreG(lp + 3, rank + 1, bVel.z);
                }
                h0 += 1;
                continue;
            }
            // (d) the sweep: one lane per item through sweepCapsuleTriangle's states (the expressions of groupSweep)
            if (total > 0) {
                const bool mine = lane < total;
                const int r = mine ? sItemRay[lane] : 0;
                const SepRayRec& R = sRay[r];
                const float radius = R.radius, halfHeight = R.halfHeight, len = R.len;
                const float minAdvance = smax(radius * 0.02f, 1e-4f); // :1295
                const float contactEps = 1e-5f;
                const int maxIter = R.maxIter;
                const F3 from = R.from, dir = R.dir, delta = R.delta;
                F3 v0{0, 0, 0}, v1{0, 0, 0}, v2{0, 0, 0};
                int triRank = 0x7fffffff;
                if (mine) {
                    const float4 t0 = sItemTri[lane * 3], t1 = sItemTri[lane * 3 + 1], t2 = sItemTri[lane * 3 + 2];
                    v0 = F3{t0.x, t0.y, t0.z}; v1 = F3{t0.w, t1.x, t1.y}; v2 = F3{t1.z, t1.w, t2.x};
                    triRank = __float_as_int(t2.w);
                }
                const F3 triNormal = normalize(cross(v1 - v0, v2 - v0));
                int phase = mine ? PH_MARCH : PH_DONE, iter = 0, refineK = 0;
                float t = 0, lastSafeT = 0, lo = 0, hi = 0, tEval = 0;
                unsigned long long myKey = ~0ull;
                float hitNy = 0;
                while (__any(phase != PH_DONE)) {
#ifdef SGE_SEP_TIMING
                    dbgTrips += 1;
#endif
                    // (an item that cannot be touched before its ray's best accepted hit so far is dropped, as in groupSweep: a later or
                    // equal hit of higher visit rank never wins)
                    const float bestToi = __uint_as_float((unsigned)(sRay[r].key >> 32));
                    if (phase == PH_MARCH) {
                        if (iter >= maxIter || t > len || lastSafeT > bestToi) phase = PH_DONE; // loop head of :1303-1307
                        else { iter += 1; tEval = t; }
                    } else if (phase == PH_REFINE) {
                        if (lo > bestToi) phase = PH_DONE;
                        else tEval = 0.5f * (lo + hi);
                    }
                    // speculative bisection (see groupSweep): refineTOI (:1361-1377) is ten dependent evaluations whose points are
                    // functions of the interval alone; idle lanes evaluate the 2 (6, 14) descendants of a refining item's midpoint in the
                    // trip in which its lane evaluates the midpoint itself, and that lane then takes 2 (3, 4) steps at once. Node
                    // numbering: 1 = the owner's midpoint; 2 n = the midpoint after "contact at n" (hi = mid), 2 n + 1 after "clear".
                    const unsigned long long refMask = __ballot(phase == PH_REFINE), idleMask = __ballot(phase == PH_DONE);
                    const int nOwn = __popcll(refMask), nHelp = __popcll(idleMask);
                    int per = 0, levels = 1;
                    if (nOwn > 0 && nHelp >= 2) {
                        per = nHelp >= 14 ? 14 : (nHelp >= 6 ? 6 : 2); // the earliest brackets get the helpers: their hits prune the others
                        levels = per == 14 ? 4 : (per == 6 ? 3 : 2);
                    }
                    const int nServe = per ? (nOwn < nHelp / per ? nOwn : nHelp / per) : 0;
                    int oRank = 0; // place of this refining item by (lo, lane)
                    if (per && nOwn > nServe) {
                        for (unsigned long long mm = refMask; mm; mm &= mm - 1) {
                            const int o = __ffsll((long long)mm) - 1;
                            const float loO = __shfl(lo, o, kWave);
                            oRank += (loO < lo || (loO == lo && o < lane)) ? 1 : 0;
                        }
                    } else oRank = prefixCount(refMask);
                    const int hIdx = prefixCount(idleMask);
                    const bool served = per && phase == PH_REFINE && oRank < nServe;
                    const bool helper = per && phase == PH_DONE && hIdx < nServe * per;
                    bool cEval = phase != PH_DONE;
                    float cT = tEval, cHalf = halfHeight, cRadius = radius;
                    F3 cFrom = from, cDir = dir, c0 = v0, c1 = v1, c2 = v2;
                    int node = 0, ownerRank = 0;
                    if (per) {
                        if (served) { sSpecOwner[oRank] = (unsigned char)lane; sSpecBits[oRank] = 0; }
                        SEP_SYNC();
                        ownerRank = helper ? hIdx / per : 0;
                        node = helper ? hIdx - ownerRank * per + 2 : 0;
                        const int src = helper ? (int)sSpecOwner[ownerRank] : lane;
                        float oLo = __shfl(lo, src, kWave), oHi = __shfl(hi, src, kWave);
                        if (helper) {
                            const int depth = 31 - __clz(node); // 1 .. 3: walk from the root along the bits of `node` below its leading one
#pragma unroll
                            for (int b = 2; b >= 0; --b) {
                                if (b < depth) {
                                    const float mid = 0.5f * (oLo + oHi);
                                    if ((node >> b) & 1) oLo = mid; else oHi = mid;
                                }
                            }
                            cT = 0.5f * (oLo + oHi);
                            const float4 t0 = sItemTri[src * 3], t1 = sItemTri[src * 3 + 1], t2 = sItemTri[src * 3 + 2];
                            c0 = F3{t0.x, t0.y, t0.z}; c1 = F3{t0.w, t1.x, t1.y}; c2 = F3{t1.z, t1.w, t2.x};
                            const SepRayRec& OR = sRay[sItemRay[src]];
                            cFrom = OR.from; cDir = OR.dir; cHalf = OR.halfHeight; cRadius = OR.radius;
                            cEval = true;
                        }
                    }
                    bool ownContact = false;
                    F3 segP{0, 0, 0}, triP{0, 0, 0};
                    float dist = 0;
                    if (cEval) dist = segmentTriangleDistance(cFrom + cDir * cT, cHalf, c0, c1, c2, segP, triP);
                    if (phase != PH_DONE) {
                        if (phase == PH_MARCH) {
                            if (dist <= radius + contactEps) { // refineTOI(t0: lastSafeT, t1: t) :1361-1377
                                const float k0 = smax(0.0f, smin(lastSafeT, len));
                                const float k1 = smax(0.0f, smin(t, len));
                                lo = smin(k0, k1);
                                hi = smax(k0, k1);
                                if (hi - lo < 1e-5f) { phase = PH_FINAL; tEval = hi; }
                                else { phase = PH_REFINE; refineK = 0; }
#ifdef SGE_SEP_TIMING
                                dbgContact += 1;
#endif
                            } else {
                                lastSafeT = t;
                                const float advance = smax(dist - radius, minAdvance);
#ifdef SGE_SEP_TIMING
                                dbgMarch += 1; if (advance == minAdvance) dbgCrawl += 1;
#endif
                                if (advance <= 0) t += minAdvance; else t += advance;
                            }
                        } else if (phase == PH_REFINE) {
                            ownContact = dist <= radius;
                            if (ownContact) hi = tEval; else lo = tEval;
                            refineK += 1;
                            if (refineK == 10) { phase = PH_FINAL; tEval = hi; }
                        } else { // PH_FINAL :1325-1346, then the acceptance filters of capsuleCastBVH :1084-1097 (blocking casts)
                            const float tHit = tEval;
                            F3 nrm;
                            if (dist < 1e-6f) nrm = dot(triNormal, dir) > 0 ? -triNormal : triNormal;
                            else nrm = normalize(segP - triP);
                            F3 triN = triNormal;
                            if (dot(triN, nrm) < 0) triN = -triN;
                            phase = PH_DONE;
                            bool ok = tHit < len;
                            if (ok) ok = !(dot(delta, nrm) >= 0) && !(dot(delta, triN) >= 0);
                            if (ok) { myKey = ((unsigned long long)__float_as_uint(tHit) << 32) | (unsigned)triRank; hitNy = nrm.y; atomicMin(&sRay[r].key, myKey); }
                        }
                    }
                    if (per) {
                        if (helper && dist <= cRadius) atomicOr(&sSpecBits[ownerRank], 1u << node);
                        SEP_SYNC();
                        if (served && phase == PH_REFINE) { // (its own step did not finish the bisection)
                            const unsigned bits = sSpecBits[oRank];
                            int nn = ownContact ? 2 : 3;
                            for (int k = 1; k < levels; ++k) {
                                const float mid = 0.5f * (lo + hi);
                                const bool c = (bits >> nn) & 1;
                                if (c) hi = mid; else lo = mid;
                                refineK += 1;
                                nn = 2 * nn + (c ? 0 : 1);
                                if (refineK == 10) { phase = PH_FINAL; tEval = hi; break; }
                            }
                        }
                    }
                }
                SEP_SYNC();
                if (myKey != ~0ull && sRay[r].key == myKey) sRay[r].ny = hitNy; // the hit capsuleCastBlocking returns: minimum (toi, visit rank)
                SEP_SYNC();
            }
            // (e) the pairs of the round in the reference's order; a blocked pair ends the round behind it
            int done = 0;
            for (int g = 0; g < G; ++g) {
                const SepHeld& H = sHeld[sOrder[h0 + g]];
                const int flags = H.flags;
                float nbx = H.bx, nbz = H.bz;
                bool deviates = false;
                if (flags & SEP_LIVE) {
                    if (flags & SEP_VN_NEG) { velI.x += H.dvIx; velI.z += H.dvIz; }
                    const SepRayRec& RA = sRay[2 * g];
                    const SepRayRec& RB = sRay[2 * g + 1];
                    const bool blockedA = (flags & SEP_CAST_A) && RA.key != ~0ull && __uint_as_float((unsigned)(RA.key >> 32)) <= A.skinWidth && RA.ny < A.minGroundDot;
                    const bool blockedB = (flags & SEP_CAST_B) && RB.key != ~0ull && __uint_as_float((unsigned)(RB.key >> 32)) <= H.S.skinWidth && RB.ny < H.S.minGroundDot;
                    F3 moveA{H.moveAx, 0, H.moveAz}, moveB{H.moveBx, 0, H.moveBz};
                    bool apply = true;
                    if (blockedA && !blockedB) { moveA = F3{0, 0, 0}; moveB = F3{-H.nx * H.penetration, 0, -H.nz * H.penetration}; deviates = true; }
                    else if (blockedB && !blockedA) { moveB = F3{0, 0, 0}; moveA = F3{H.nx * H.penetration, 0, H.nz * H.penetration}; deviates = true; }
                    else if (blockedA && blockedB) { apply = false; deviates = true; }
                    if (apply) {
                        posI = posI + moveA;
                        const F3 nb = F3{H.bx, H.S.posY, H.bz} + moveB;
                        nbx = nb.x; nbz = nb.z;
                    }
                }
                if (lane == 0) {
                    unsigned long long* lp = F.live + (size_t)H.c * 4;
                    sepStoreG(lp, H.rank + 1, nbx); sepStoreG(lp + 1, H.rank + 1, nbz); sepStoreG(lp + 2, H.rank + 1, H.nbvx); sepStoreG(lp + 3, H.rank + 1, H.nbvz);
                }
                done = g + 1;
                if (deviates) break;
            }
            h0 += done;
            SEP_SYNC();
        }
        SEP_SYNC();
        while (deferred && passDeferred()) { if (giveUp()) return; __builtin_amdgcn_s_sleep(kSepPollSleep); }
        SEP_TS(i, 3);
#ifdef SGE_SEP_TIMING
        {   // (per-lane counters of the sweep: summed over the wavefront)
            for (int o = 32; o > 0; o >>= 1) { dbgMarch += __shfl_xor(dbgMarch, o, kWave); dbgCrawl += __shfl_xor(dbgCrawl, o, kWave); dbgContact += __shfl_xor(dbgContact, o, kWave); }
        }
        SEP_TSV(i, 4, nh); SEP_TSV(i, 5, dbgRounds | (dbgContact << 8) | (dbgCrawl << 16)); SEP_TSV(i, 6, dbgFall | (dbgMarch << 8)); SEP_TSV(i, 7, dbgTrips);
#endif
        SepAgentDev* Ai = K.agents + i;
        if (lane == 0) { sepStore3(Ai->position, posI); sepStore3(Ai->velocity, velI); } // nobody reads it before the pass ends
    }
}

// A pass in which an agent was pushed further than its candidates' cells allow (control[2] == 2: nothing else went wrong) is run again
// from the saved state with candidates from 7 x 7 cells, on the device's own decision: the serial kernel takes 0.35 s for 8,192
// agents and 2.4 s for 31,250, the second attempt what a pass takes. (The host widens the following steps' reach from the same
// flag; this is for the step that shows it first — a crowd spawned on top of itself, a teleport.)
__global__ void sep_retry_decide_kernel(SepFlow F) {
    const int again = F.control[2] == 2 ? 1 : 0;
    F.control[8] = again;
    if (again) { F.control[1] = 0; F.control[2] = 0; F.control[9] += 1; }
}
__global__ void sep_retry_restore_kernel(SepLaunch K, SepFlow F) {
    const int n = F.control[0], i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || F.control[8] == 0) return;
    SepAgentDev& a = K.agents[i];
    for (int k = 0; k < 3; ++k) { a.position[k] = F.backup[i * 6 + k]; a.velocity[k] = F.backup[i * 6 + 3 + k]; }
    F.ver[i] = 0;
    const float lv[4] = {a.position[0], a.position[2], a.velocity[0], a.velocity[2]};
    for (int k = 0; k < 4; ++k) F.live[(size_t)i * 4 + k] = (unsigned long long)__float_as_uint(lv[k]); // version 0
}

// the same pass by ONE wavefront in the reference's own order, from the state at the head of the pass: runs only when the dataflow
// pass gave up (more than kSepMaxCand candidates for some agent, or an agent pushed further than a cell)
__global__ __launch_bounds__(kWave) void sep_serial_kernel(SepLaunch K, SepFlow F) {
    const int lane = laneId();
    const DevCollision& col = K.col;
    WaveStats st{0, 0, 0, 0, 0, 0, 0};
    const int n = F.control[0];
    if (n <= 1 || F.control[2] == 0) return;
    if (lane == 0) F.control[6] |= F.control[2];
    const float cellSize = __int_as_float(F.control[3]);
    for (int i = lane; i < n; i += kWave)
        for (int k = 0; k < 3; ++k) { sepStore(&K.agents[i].position[k], F.backup[i * 6 + k]); sepStore(&K.agents[i].velocity[k], F.backup[i * 6 + 3 + k]); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        SepAgentDev* Ai = K.agents + i;
        const F3 aPos = sepLoad3(Ai->position), aVel = sepLoad3(Ai->velocity);
        const SepPairAgent A = sepPairAgent(K, *Ai);
        F3 posI = aPos, velI = aVel;
        const int cx = (int)floorf(aPos.x / cellSize), cz = (int)floorf(aPos.z / cellSize);
        for (int dz = -1; dz <= 1; ++dz)
            for (int dx = -1; dx <= 1; ++dx) {
                const int tx = cx + dx, tz = cz + dz;
                const unsigned b = sepHash(tx, tz, F.H);
                const int pEnd = F.bucketStart[b + 1];
                for (int p = F.bucketStart[b]; p < pEnd; ++p) {
                    const int j = F.bucketItems[p];
                    if (j <= i || F.cell[2 * j] != tx || F.cell[2 * j + 1] != tz) continue;
                    SepAgentDev* Aj = K.agents + j;
                    F3 bPos = sepLoad3(Aj->position), bVel = sepLoad3(Aj->velocity);
                    const SepPairAgent B = sepPairAgent(K, *Aj);
                    if (sepPair(col, A, B, aPos, aVel, posI, velI, bPos, bVel, K.separationMargin, K.heightMargin, st)) {
                        if (lane == 0) { sepStore3(Aj->position, bPos); sepStore3(Aj->velocity, bVel); }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
            }
        if (lane == 0) { sepStore3(Ai->position, posI); sepStore3(Ai->velocity, velI); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

int separationFlowBuckets(int count);
// byte offset of SepFlow::control inside the scratch buffer (diagnostics: sge_debug_separation)
size_t separationFlowControlOffset(int count) {
    const size_t n = (size_t)count, H = (size_t)separationFlowBuckets(count);
    auto up = [](size_t b) { return (b + 15) & ~(size_t)15; };
    return up(8 * n) + up(4 * (H + 1)) + up(4 * H) + up(4 * n) * 4 + up(sizeof(int2) * kSepMaxCand * n);
}
size_t separationFlowBytes(int count) {
    const size_t n = (size_t)count, H = (size_t)separationFlowBuckets(count);
    return 8 * n + 4 * (H + 1) + 4 * H + 4 * n * 4 + sizeof(int2) * kSepMaxCand * n + 64 + 24 * n + 48 * (size_t)kSepTriCap * n + 32 * n + 64 * n + 256;
}
int separationFlowBuckets(int count) { int H = 64; while (H < 2 * count) H <<= 1; return H; }

void launch_separation(const DevCrowd& crowd, const DevCollision& col, int iterations, float separationMargin, float heightMargin,
                       void* agentScratch, int* counts, void* flowScratch, hipStream_t s, int reach, int* flagsHost) {
    if (crowd.count <= 1) return;
    SepLaunch K{crowd, col, iterations < 1 ? 1 : iterations, separationMargin, heightMargin, reinterpret_cast<SepAgentDev*>(agentScratch), counts};
    // One wavefront with everything in LDS walks the reference's loop as it stands; from a few dozen agents on the dataflow over agents
    // is faster although it is nine small launches per pass (16 agents: 0.39 against 0.54 ms per step, 32: 0.22 / 0.25, 64: 0.56 / 0.42,
    // 128: 1.36 / 0.63, 1,024: 23-49 / 3-10). SGE_SEPARATION_FLOW=0 / 1 forces either form (tests).
    constexpr int kSepOneWaveAgents = 48;
    const int forced = getenv("SGE_SEPARATION_FLOW") ? atoi(getenv("SGE_SEPARATION_FLOW")) : -1;
    if (crowd.count <= SGE_MAX_SEPARATION_AGENTS && (forced == 0 || (forced < 0 && crowd.count <= kSepOneWaveAgents))) {
        hipLaunchKernelGGL(separation_resolve_kernel, dim3(1), dim3(kWave), 0, s, K);
        hipLaunchKernelGGL(separation_post_kernel, dim3(crowd.count), dim3(kWave), 0, s, K);
        return;
    }
    const int n = crowd.count, H = separationFlowBuckets(n);
    SepFlow F;
    char* p = reinterpret_cast<char*>(flowScratch);
    auto carve = [&](size_t bytes) { char* q = p; p += (bytes + 15) & ~(size_t)15; return q; };
    F.cell = reinterpret_cast<int*>(carve(8 * (size_t)n));
    F.bucketStart = reinterpret_cast<int*>(carve(4 * ((size_t)H + 1)));
    F.bucketCursor = reinterpret_cast<int*>(carve(4 * (size_t)H));
    F.bucketItems = reinterpret_cast<int*>(carve(4 * (size_t)n));
    F.need = reinterpret_cast<int*>(carve(4 * (size_t)n));
    F.ver = reinterpret_cast<int*>(carve(4 * (size_t)n));
    F.candCount = reinterpret_cast<int*>(carve(4 * (size_t)n));
    F.cand = reinterpret_cast<int2*>(carve(sizeof(int2) * kSepMaxCand * (size_t)n));
    F.control = reinterpret_cast<int*>(carve(64));
    F.backup = reinterpret_cast<float*>(carve(24 * (size_t)n));
    F.stat = reinterpret_cast<SepStaticDev*>(carve(sizeof(SepStaticDev) * (size_t)n));
    F.live = reinterpret_cast<unsigned long long*>(carve(32 * (size_t)n));
    F.triCache = reinterpret_cast<float4*>(carve(48 * (size_t)kSepTriCap * n));
    F.triBox = reinterpret_cast<float*>(carve(32 * (size_t)n));
    if (getenv("SGE_SEPARATION_BVH_CASTS") && atoi(getenv("SGE_SEPARATION_BVH_CASTS")) != 0) F.triCache = nullptr; // tests: every pair casts through the BVH
    F.H = H;
    F.reach = reach < 2 ? 2 : (reach > 3 ? 3 : reach);
    const int blocks = (n + 255) / 256;
    (void)hipMemsetAsync(F.control + 5, 0, 8, s); // [5] "pushed further than a cell" over the whole step, [6] redo flags of any pass of the step
    hipLaunchKernelGGL(sep_list_kernel, dim3(1), dim3(1024), 0, s, K, F);
    F.triReach = getenv("SGE_SEPARATION_REACH") ? (float)atof(getenv("SGE_SEPARATION_REACH")) : kSepTriReach;
    F.retry = 0;
    const bool noRetry = getenv("SGE_SEPARATION_NO_RETRY") && atoi(getenv("SGE_SEPARATION_NO_RETRY")) != 0; // experiments / tests
    F.noDefer = getenv("SGE_SEPARATION_NO_DEFER") && atoi(getenv("SGE_SEPARATION_NO_DEFER")) != 0 ? 1 : 0;
    const int tracePass = getenv("SGE_SEPARATION_TRACE_PASS") ? atoi(getenv("SGE_SEPARATION_TRACE_PASS")) : K.iterations - 1;
    for (int it = 0; it < K.iterations; ++it) {
        F.trace = it == tracePass ? 1 : 0;
        (void)hipMemsetAsync(F.bucketCursor, 0, 4 * (size_t)H, s);
        hipLaunchKernelGGL(sep_cells_kernel, dim3(blocks), dim3(256), 0, s, K, F);
        hipLaunchKernelGGL(sep_scan_kernel, dim3(1), dim3(1024), 0, s, F);
        hipLaunchKernelGGL(sep_scatter_kernel, dim3(blocks), dim3(256), 0, s, F);
        hipLaunchKernelGGL(sep_sort_kernel, dim3((H + 255) / 256), dim3(256), 0, s, F);
        hipLaunchKernelGGL(sep_cand_kernel, dim3(n), dim3(kWave), 0, s, F);
        if (F.triCache) hipLaunchKernelGGL(sep_tricache_kernel, dim3(n), dim3(kWave), 0, s, K, F);
        // as many wavefronts as stay resident together do useful work; more would only queue behind them
        int waves = std::min(n, currentDeviceCUs() * (F.triCache ? 4 : 8)); // (37 KB of LDS per loop in the second form)
        if (getenv("SGE_SEPARATION_WAVES") && atoi(getenv("SGE_SEPARATION_WAVES")) > 0) waves = std::min(n, atoi(getenv("SGE_SEPARATION_WAVES")));
        if (F.triCache) hipLaunchKernelGGL(sep_flow2_kernel, dim3(waves), dim3(kWave), 0, s, K, F);
        else hipLaunchKernelGGL(sep_flow_kernel, dim3(waves), dim3(kWave), 0, s, K, F);
        if (F.reach < 3 && F.triCache && !noRetry) { // second attempt with wider candidate cells, if the device finds the first one short
            SepFlow R = F;
            R.reach = 3; R.retry = 1;
            hipLaunchKernelGGL(sep_retry_decide_kernel, dim3(1), dim3(1), 0, s, F);
            hipLaunchKernelGGL(sep_retry_restore_kernel, dim3(blocks), dim3(256), 0, s, K, F);
            hipLaunchKernelGGL(sep_cand_kernel, dim3(n), dim3(kWave), 0, s, R);
            hipLaunchKernelGGL(sep_flow2_kernel, dim3(waves), dim3(kWave), 0, s, K, R);
        }
        hipLaunchKernelGGL(sep_serial_kernel, dim3(1), dim3(kWave), 0, s, K, F);
    }
    hipLaunchKernelGGL(separation_post_kernel, dim3(n), dim3(kWave), 0, s, K);
    // what the host sizes the NEXT step's reach from: [0] an agent was pushed further than a cell in some pass, [1] a pass was redone
    if (flagsHost) (void)hipMemcpyAsync(flagsHost, F.control + 5, 8, hipMemcpyDeviceToHost, s);
}

// Picks this step's heavy characters by last step's sweep cost: flags[e] = 1 and an entry in the heavy list (at most
// heavyCap; the rest stay with the grouped launch). counts[1] is zeroed before the launch; list order is arbitrary
// (atomics) and results do not depend on it. Everybody else is counted into a histogram of cost classes for the order list.
constexpr int kCostBuckets = 32;
__device__ __forceinline__ int costBucket(int cost) { const int b = cost >> 7; return b < 0 ? 0 : (b >= kCostBuckets ? kCostBuckets - 1 : b); }
__global__ void classify_kernel(const int* cost, int first, int count, int threshold, int heavyCap, int* lists, int* counts,
                                uint8_t* flags, int* hist) {
    __shared__ int h[kCostBuckets];
    __shared__ int hTotal;
    if (threadIdx.x < kCostBuckets) h[threadIdx.x] = 0;
    if (threadIdx.x == 0) hTotal = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        const int e = first + i;
        uint8_t heavy = 0;
        if (threshold >= 0 && cost[e] > threshold) {
            atomicAdd(&counts[2], 1); // demand, whatever the cap: sizes the multi-wave grid of the coming steps (sge_tick)
            int pos = atomicAdd(&counts[1], 1);
            if (pos < heavyCap) { lists[count + pos] = e; heavy = 1; }
            else atomicSub(&counts[1], 1);
        }
        flags[e] = heavy;
        if (!heavy) atomicAdd(&h[costBucket(cost[e])], 1);
        atomicAdd(&hTotal, cost[e] < 0 ? 0 : cost[e]);
    }
    __syncthreads();
    if (threadIdx.x == 0 && hTotal) atomicAdd(&counts[3], hTotal); // the crowd's distance evaluations of the last step (schedule choice, sge_tick)
    if (threadIdx.x < kCostBuckets && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// hist[0..32) -> start of every class in the order list, most expensive class first (hist[32..64) = scatter cursors); zeroes the
// histogram for the next step; counts[0] = number of listed characters. One lane per class, loads with agent scope: written as a
// one-thread loop hipcc 7.2 turns the reads into scalar loads (s_load_dwordx16) and issues the zeroing vector stores of the same
// words before those loads have returned.
__global__ void order_scan_kernel(int* hist, int* counts) {
    const int lane = threadIdx.x;
    const int b = kCostBuckets - 1 - lane; // lane 0 = most expensive class
    int c = 0;
    if (lane < kCostBuckets) c = __hip_atomic_load(&hist[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane < kCostBuckets) { hist[kCostBuckets + b] = incl - c; hist[b] = 0; }
    if (lane == kCostBuckets - 1) counts[0] = incl;
}
// Scatter with one global atomic per (workgroup, class): 10k single atomics on ~10 hot cursors took 76 us per step.
__global__ void order_scatter_kernel(const int* cost, int first, int count, const uint8_t* flags, int* hist, int* order) {
    __shared__ int h[kCostBuckets], base[kCostBuckets];
    if (threadIdx.x < kCostBuckets) h[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int e = -1, b = 0, local = 0;
    if (i < count && !flags[first + i]) {
        e = first + i;
        b = costBucket(cost[e]);
        local = atomicAdd(&h[b], 1);
    }
    __syncthreads();
    if (threadIdx.x < kCostBuckets && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&hist[kCostBuckets + threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (e >= 0) order[base[b] + local] = e;
}

bool launch_move(const MoveLaunch& L, hipStream_t s) {
    if (L.count <= 0) return false;
    static const bool grouped = !(getenv("SGE_MOVE_GROUP") && atoi(getenv("SGE_MOVE_GROUP")) == 0);
    static const bool pipelineSetting = !(getenv("SGE_MOVE_PIPELINE_LISTS") && atoi(getenv("SGE_MOVE_PIPELINE_LISTS")) == 0); // experiments
    const bool move = (L.stages & SGE_STAGE_MOVE) != 0;
    const bool agents = (L.stages & SGE_STAGE_AGENTS) && L.agents.all;
    const bool heavy = L.heavyThreshold >= 0;
    static const int ldsPad = getenv("SGE_MOVE_LDS_PAD") ? atoi(getenv("SGE_MOVE_LDS_PAD")) : 0; // experiments: caps workgroups per CU
    static const int soloSetting = getenv("SGE_GROUP_SOLO") ? atoi(getenv("SGE_GROUP_SOLO")) : 128; // experiments
    const int solo = grouped ? std::max(0, std::min(soloSetting, L.count / 16)) : 0;
    const int groups = solo + (L.count - solo + kGroup - 1) / kGroup;
    const int blocks = (L.count + 255) / 256;
    // The scheduling lists — last step's costs -> heavy list (multi-wave launch) + order list of everybody else (grouped launch) —
    // depend on nothing of the step they schedule, so they are built BEHIND the move stage of the step before, on the second stream,
    // while that step's pose kernel runs: the memset and three small kernels (and the cross-stream event in front of the grouped
    // launch) were 50 us between the end of move_kernel<0> and the start of the grouped launch, on the step's critical chain.
    const bool pipelined = move && grouped && heavy && pipelineSetting;
    auto buildLists = [&](hipStream_t q, int cap) {
        (void)hipMemsetAsync(L.listCounts, 0, 4 * sizeof(int), q);
        hipLaunchKernelGGL(classify_kernel, dim3(blocks), dim3(256), 0, q, L.cost, L.first, L.count, L.heavyThreshold, cap, L.lists,
                           L.listCounts, L.heavyFlags, L.orderHist);
        if (grouped) {
            hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(64), 0, q, L.orderHist, L.listCounts);
            hipLaunchKernelGGL(order_scatter_kernel, dim3(blocks), dim3(256), 0, q, L.cost, L.first, L.count, L.heavyFlags, L.orderHist, L.lists);
        }
    };
    // how many characters asked for the multi-wave launch, and the crowd's evaluations: read by the host when it enqueues a later
    // step (pinned memory; the copy takes ~45 us, so it goes behind the event that says the lists are ready)
    auto copyDemand = [&](hipStream_t q) {
        if (heavy && L.heavyDemandHost) (void)hipMemcpyAsync(L.heavyDemandHost, L.listCounts + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, q);
    };
    const bool listsFromLastStep = move && pipelined && L.listsReady; // built behind the previous step, on the second stream
    // A build of the previous stage that this one cannot use (another range, another threshold, the option switched off) may still be
    // running on the second stream: it writes the same lists, counts, flags and histogram as the build below and as the kernels of
    // this stage read, so the main stream joins it first (the stage that CAN use it waits further down, in front of its first reader).
    if (move && L.listsPending && !listsFromLastStep) (void)hipStreamWaitEvent(s, L.evListsReady, 0);
    if (move && !listsFromLastStep) { buildLists(s, L.heavyCap); copyDemand(s); } // first step, or something changed
    // (part 0 at the head of the grouped launch instead — no launch in front of the stage, no round trip of the working set — was
    // built and measured: the grouped kernel then needs 166 registers instead of 155, one wavefront fewer fits beside two resident
    // LBS wavefronts, and the step is 1-6 % slower: 0.998 against 0.987 ms, against 0.937 with 155)
    // (four characters per wavefront of part 0, one after the other — 2,500 workgroups instead of 10,000 — measured 0.963 against
    // 0.946 ms per step: the launch is short either way and a wavefront's four overlap queries in a row are not)
    hipLaunchKernelGGL((move_kernel<0, false>), dim3(L.count), dim3(kWave), 0, s, L);
    if (!move) return false;
    MoveLaunch G = L;
    // The multi-wave launch goes FIRST and stays on the main stream, right behind part 0: its 512-thread workgroups need
    // two free wavefront places on every SIMD of one CU at the same moment, which they find while the chip holds nothing but the
    // skin launch of the previous step — and do not find for several hundred microseconds once the one-wave workgroups of the
    // grouped launch have taken every place that comes free (beside resident LBS workgroups the move stage took 0.94 ms instead of
    // 0.52 for exactly this reason). So the grouped launch runs on the second stream,
    // behind an event: the cross-stream hand-over is what gives the multi-wave workgroups their head start.
    // Queue packets between two dependent launches cost ~5 us each (tools/step_gaps.py), so nothing sits on the way from part 0 to
    // the grouped launch but that one event, and nothing between the end of the grouped launch and the main stream's next kernel but
    // the wait for it: the wait for the lists (part 0 reads none of them; the order list is the second stream's own earlier work)
    // stands in front of the multi-wave launch only, and "the multi-wave launch is done" is recorded right behind that launch.
    hipStream_t gs = heavy ? L.heavyStream : s;
    if (heavy) {
        (void)hipEventRecord(L.evClassified, s);
        (void)hipStreamWaitEvent(gs, L.evClassified, 0);
        if (listsFromLastStep) (void)hipStreamWaitEvent(s, L.evListsReady, 0);
        MoveLaunch H = L;
        H.list = L.lists + L.count; H.listCount = L.listCounts + 1;
        const int heavyGrid = L.count < L.heavyCap ? L.count : L.heavyCap;
        if (agents) hipLaunchKernelGGL((move_kernel<1, true, true>), dim3(heavyGrid), dim3(kWave * kHeavyWaves), 0, s, H);
        else hipLaunchKernelGGL((move_kernel<1, false, true>), dim3(heavyGrid), dim3(kWave * kHeavyWaves), 0, s, H);
        (void)hipEventRecord(L.evClassified, s); // from here on: "part 0 and the multi-wave launch are done" (sge_tick orders the pose stream behind it)
    } else if (listsFromLastStep) {
        (void)hipStreamWaitEvent(s, L.evListsReady, 0);
    }
    if (grouped) {
        G.order = L.lists; G.orderCount = L.listCounts; G.solo = solo;
        // kGroup characters per wavefront, members drawn from the order list
        if (agents) hipLaunchKernelGGL((move_group_kernel<true>), dim3(groups), dim3(kWave), ldsPad, gs, G);
        else hipLaunchKernelGGL((move_group_kernel<false>), dim3(groups), dim3(kWave), ldsPad, gs, G);
    } else if (agents) hipLaunchKernelGGL((move_kernel<1, true>), dim3(L.count), dim3(kWave), 0, gs, L); // skips flagged characters
    else hipLaunchKernelGGL((move_kernel<1, false>), dim3(L.count), dim3(kWave), ldsPad, gs, L);
    if (heavy) (void)hipEventRecord(L.evHeavyDone, gs); // the second stream's launch is done
    if (pipelined) { // the NEXT step's lists, behind both launches of this one, beside whatever the main stream does next
        (void)hipStreamWaitEvent(gs, L.evClassified, 0);
        buildLists(gs, L.nextHeavyCap);
        (void)hipEventRecord(L.evListsReady, gs);
        copyDemand(gs);
    }
    if (heavy) (void)hipStreamWaitEvent(s, L.evHeavyDone, 0);
    return pipelined;
}

// ---- batched single queries (the CollisionQuery facade) ---------------------
__global__ __launch_bounds__(kWave, 3) void cast_query_kernel(DevCollision col, const sge_capsule_query* q, int n,
                                                           sge_capsule_cast_hit* out, unsigned long long* stats) {
    const int i = blockIdx.x;
    WaveStats st{0, 0, 0, 0, 0, 0, 0};
    sge_capsule_query Q = q[i];
    sh.rayCount = 1;
    sh.rayFrom[0] = F3{Q.from[0], Q.from[1], Q.from[2]};
    sh.rayDelta[0] = F3{Q.delta[0], Q.delta[1], Q.delta[2]};
    __syncthreads();
    waveCastRays(col, Q.radius, Q.halfHeight, Q.mode == SGE_CAST_BLOCKING, Q.mode == SGE_CAST_GROUND, Q.minNormalY, Q.mask, st);
    const bool got = rayHit(0);
    const CastRec r = sh.rayRec[0];
    if (laneId() == 0) {
        sge_capsule_cast_hit h;
        h.hit = got ? 1 : 0;
        h.toi = got ? r.toi : 0;
        F3 z{0, 0, 0};
        F3 p = got ? r.position : z, nn = got ? r.normal : z, tn = got ? r.triNormal : z;
        h.position[0] = p.x; h.position[1] = p.y; h.position[2] = p.z;
        h.normal[0] = nn.x; h.normal[1] = nn.y; h.normal[2] = nn.z;
        h.triangleNormal[0] = tn.x; h.triangleNormal[1] = tn.y; h.triangleNormal[2] = tn.z;
        h.triangleIndex = got ? r.triIndex : -1;
        DevMaterial m = got ? col.materials[r.triIndex] : DevMaterial{0, 0, 0};
        h.material.muS = m.muS; h.material.muK = m.muK; h.material.flattenGround = m.flatten;
        out[i] = h;
        if (stats) {
            unsigned long long* sp = statShard(stats);
            atomicAdd(&sp[0], (unsigned long long)st.queries);
            atomicAdd(&sp[1], (unsigned long long)st.candidates);
            if (st.overflow) atomicAdd(&sp[3], (unsigned long long)st.overflow);
        }
    }
}

__global__ __launch_bounds__(kWave, 3) void overlap_query_kernel(DevCollision col, const sge_capsule_query* q, int n, int maxHits,
                                                              sge_capsule_overlap_hit* out, int32_t* counts,
                                                              unsigned long long* stats) {
    const int i = blockIdx.x;
    const int lane = laneId();
    WaveStats st{0, 0, 0, 0, 0, 0, 0};
    sge_capsule_query Q = q[i];
    int cnt = waveCapsuleOverlapAll(col, F3{Q.from[0], Q.from[1], Q.from[2]}, Q.radius, Q.halfHeight, maxHits, Q.mask, st);
    if (lane < maxHits) {
        sge_capsule_overlap_hit h;
        if (lane < cnt) {
            OverlapRec r = sOvl[lane];
            h.depth = r.depth;
            h.position[0] = r.position.x; h.position[1] = r.position.y; h.position[2] = r.position.z;
            h.normal[0] = r.normal.x; h.normal[1] = r.normal.y; h.normal[2] = r.normal.z;
            h.triangleNormal[0] = r.triNormal.x; h.triangleNormal[1] = r.triNormal.y; h.triangleNormal[2] = r.triNormal.z;
            h.triangleIndex = r.triIndex;
            DevMaterial m = col.materials[r.triIndex];
            h.material.muS = m.muS; h.material.muK = m.muK; h.material.flattenGround = m.flatten;
        } else {
            h.depth = 0;
            for (int k = 0; k < 3; ++k) { h.position[k] = 0; h.normal[k] = 0; h.triangleNormal[k] = 0; }
            h.triangleIndex = -1;
            h.material.muS = 0; h.material.muK = 0; h.material.flattenGround = 0;
        }
        out[(size_t)i * maxHits + lane] = h;
    }
    if (lane == 0) {
        counts[i] = cnt;
        if (stats && st.overflow) atomicAdd(&statShard(stats)[3], (unsigned long long)st.overflow);
    }
}

// CollisionQuery.capsuleOverlap (:1119-1199): deepest hit, first visited wins equal depths
__global__ __launch_bounds__(kWave, 3) void overlap_deepest_kernel(DevCollision col, const sge_capsule_query* q, int n,
                                                                   sge_capsule_overlap_hit* out, int32_t* found,
                                                                   unsigned long long* stats) {
    const int i = blockIdx.x;
    const int lane = laneId();
    WaveStats st{0, 0, 0, 0, 0, 0, 0};
    const sge_capsule_query Q = q[i];
    const F3 from{Q.from[0], Q.from[1], Q.from[2]};
    unsigned long long bestKey = ~0ull; // (~depth bits << 32) | rank: deeper first, then earlier visit
    if (col.root >= 0) {
        F3 up{0, 1, 0};
        F3 a0 = from + up * Q.halfHeight, b0 = from - up * Q.halfHeight;
        F3 ext{Q.radius, Q.radius, Q.radius};
        F3 minP = vmin(a0, b0) - ext, maxP = vmax(a0, b0) + ext;
        int stackSize = initTraversal(col), rangeCount = 0, candCount = 0;
        __syncthreads();
        while (true) {
            while ((stackSize > 0 || rangeCount > 0) && candCount < kWave) expandNodes(col, minP, maxP, Q.mask, stackSize, rangeCount, candCount, st);
            if (candCount == 0) break;
            int nb = candCount < kWave ? candCount : kWave;
            candCount -= nb;
            unsigned long long key = ~0ull;
            OverlapRec rec;
            rec.depth = 0; rec.position = rec.normal = rec.triNormal = F3{0, 0, 0}; rec.triIndex = -1; rec.rank = 0x7fffffff;
            if (lane < nb) {
                Tri tri = loadTri(col, sh.cand[candCount + lane]);
                F3 segP, triP;
                float dist = segmentTriangleDistance(from, Q.halfHeight, tri.v0, tri.v1, tri.v2, segP, triP);
                float depth = Q.radius - dist;
                if (!(dist >= Q.radius) && !(depth <= 0.0f)) { // :1170-1172 with bestDepth starting at 0
                    F3 triNormal = normalize(cross(tri.v1 - tri.v0, tri.v2 - tri.v0));
                    F3 nn = dist < 1e-6f ? triNormal : normalize(segP - triP);
                    F3 triN = triNormal;
                    if (dot(triN, nn) < 0) triN = -triN;
                    rec.depth = depth; rec.position = triP; rec.normal = nn; rec.triNormal = triN; rec.triIndex = tri.triIndex; rec.rank = tri.rank;
                    key = ((unsigned long long)(~__float_as_uint(depth)) << 32) | (unsigned)tri.rank;
                }
            }
            unsigned long long k = waveMinU64(key);
            if (k < bestKey) {
                bestKey = k;
                if (key == k) sOvl[0] = rec;
            }
            __syncthreads();
        }
    }
    if (lane == 0) {
        sge_capsule_overlap_hit h;
        const bool got = bestKey != ~0ull;
        found[i] = got ? 1 : 0;
        OverlapRec r = sOvl[0];
        F3 z{0, 0, 0};
        F3 p = got ? r.position : z, nn = got ? r.normal : z, tn = got ? r.triNormal : z;
        h.depth = got ? r.depth : 0;
        h.position[0] = p.x; h.position[1] = p.y; h.position[2] = p.z;
        h.normal[0] = nn.x; h.normal[1] = nn.y; h.normal[2] = nn.z;
        h.triangleNormal[0] = tn.x; h.triangleNormal[1] = tn.y; h.triangleNormal[2] = tn.z;
        h.triangleIndex = got ? r.triIndex : -1;
        DevMaterial m = got ? col.materials[r.triIndex] : DevMaterial{0, 0, 0};
        h.material.muS = m.muS; h.material.muK = m.muK; h.material.flattenGround = m.flatten;
        out[i] = h;
        if (stats && st.overflow) atomicAdd(&statShard(stats)[3], (unsigned long long)st.overflow);
    }
}

// ---------------------------------------------------------------------------
// CollisionQuery.raycast (CollisionQuery.swift:768-785, 916-978, 1575-1631). Not on the tick path. Its result depends
// on the order nodes are visited in (`range.0 > closestT` prunes with the running closest hit, and equal distances
// keep the first found), so one thread follows the reference's own stack traversal of the binary BVH per ray.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool rayTriangle(F3 origin, F3 direction, F3 v0, F3 v1, F3 v2, float eps, float& tOut) { // :1575-1601
    F3 e1 = v1 - v0, e2 = v2 - v0;
    F3 pvec = cross(direction, e2);
    float det = dot(e1, pvec);
    if (fabsf(det) < eps) return false;
    float invDet = 1.0f / det;
    F3 tvec = origin - v0;
    float u = dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return false;
    F3 qvec = cross(tvec, e1);
    float v = dot(direction, qvec) * invDet;
    if (v < 0 || (u + v) > 1) return false;
    float t = dot(e2, qvec) * invDet;
    if (!(t >= 0)) return false;
    tOut = t;
    return true;
}
__device__ __forceinline__ bool rayAABB(F3 origin, F3 inv, const DevNode& n, float& tminOut) { // :1603-1630
    float tmin = (n.mnx - origin.x) * inv.x, tmax = (n.mxx - origin.x) * inv.x;
    if (tmin > tmax) { float w = tmin; tmin = tmax; tmax = w; }
    float tymin = (n.mny - origin.y) * inv.y, tymax = (n.mxy - origin.y) * inv.y;
    if (tymin > tymax) { float w = tymin; tymin = tymax; tymax = w; }
    if (tmin > tymax || tymin > tmax) return false;
    tmin = smax(tmin, tymin);
    tmax = smin(tmax, tymax);
    float tzmin = (n.mnz - origin.z) * inv.z, tzmax = (n.mxz - origin.z) * inv.z;
    if (tzmin > tzmax) { float w = tzmin; tzmin = tzmax; tzmax = w; }
    if (tmin > tzmax || tzmin > tmax) return false;
    tminOut = smax(tmin, tzmin);
    return true;
}
constexpr int kRayStack = 192; // binary BVH depth <= 120 (checked at build), + one sibling per level
__global__ __launch_bounds__(kWave) void raycast_query_kernel(DevCollision col, const sge_ray_query* q, int n, sge_raycast_hit* out) {
    const int i = blockIdx.x * kWave + threadIdx.x;
    if (i >= n) return;
    const sge_ray_query Q = q[i];
    const F3 origin{Q.origin[0], Q.origin[1], Q.origin[2]}, direction{Q.direction[0], Q.direction[1], Q.direction[2]};
    const F3 inv{direction.x != 0 ? 1.0f / direction.x : kFloatMax, direction.y != 0 ? 1.0f / direction.y : kFloatMax,
                 direction.z != 0 ? 1.0f / direction.z : kFloatMax};
    sge_raycast_hit best{};
    best.triangleIndex = -1;
    int stack[kRayStack];
    for (int set = 0; set < 2; ++set) {
        if (col.binRoot[set] < 0) continue;
        const DevNode* nodes = col.binNodes[set];
        float closestT = Q.maxDistance;
        sge_raycast_hit hit{};
        int sp = 0;
        stack[sp++] = col.binRoot[set];
        while (sp > 0) {
            const DevNode node = nodes[stack[--sp]];
            float rangeMin;
            if (!rayAABB(origin, inv, node, rangeMin)) continue;
            if (rangeMin > closestT) continue;
            if (node.a < 0) { // leaf: slots [~a, ~a + b)
                const int first = col.binSlotBase[set] + (~node.a);
                for (int s = first; s < first + node.b; ++s) {
                    Tri tri = loadTri(col, s);
                    if ((col.tris[s].layer & Q.mask) == 0) continue;
                    float t;
                    if (rayTriangle(origin, direction, tri.v0, tri.v1, tri.v2, 1e-6f, t) && t < closestT) {
                        F3 nrm = normalize(cross(tri.v1 - tri.v0, tri.v2 - tri.v0));
                        if (dot(nrm, direction) > 0) nrm = -nrm;
                        F3 pos = origin + direction * t;
                        closestT = t;
                        hit.hit = 1; hit.distance = t;
                        hit.position[0] = pos.x; hit.position[1] = pos.y; hit.position[2] = pos.z;
                        hit.normal[0] = nrm.x; hit.normal[1] = nrm.y; hit.normal[2] = nrm.z;
                        hit.triangleIndex = tri.triIndex;
                    }
                }
            } else if (sp + 2 <= kRayStack) {
                stack[sp++] = node.a;
                stack[sp++] = node.b;
            }
        }
        // chooseNearest (:902-907): the static hit wins ties
        if (hit.hit && (!best.hit || !(best.distance <= hit.distance))) best = hit;
    }
    if (best.hit) {
        DevMaterial m = col.materials[best.triangleIndex];
        best.material = sge_surface_material{m.muS, m.muK, m.flatten};
    }
    out[i] = best;
}

void launch_raycast_queries(const DevCollision& col, const sge_ray_query* d_q, int n, sge_raycast_hit* d_out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(raycast_query_kernel, dim3((n + kWave - 1) / kWave), dim3(kWave), 0, s, col, d_q, n, d_out);
}

void launch_overlap_deepest_queries(const DevCollision& col, const sge_capsule_query* d_q, int n,
                                    sge_capsule_overlap_hit* d_out, int32_t* d_found, unsigned long long* stats, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(overlap_deepest_kernel, dim3(n), dim3(kWave), 0, s, col, d_q, n, d_out, d_found, stats);
}

void launch_cast_queries(const DevCollision& col, const sge_capsule_query* d_q, int n, sge_capsule_cast_hit* d_out,
                         unsigned long long* stats, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(cast_query_kernel, dim3(n), dim3(kWave), 0, s, col, d_q, n, d_out, stats);
}
void launch_overlap_queries(const DevCollision& col, const sge_capsule_query* d_q, int n, int maxHits,
                            sge_capsule_overlap_hit* d_out, int32_t* d_counts, unsigned long long* stats, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(overlap_query_kernel, dim3(n), dim3(kWave), 0, s, col, d_q, n, maxHits, d_out, d_counts, stats);
}

// collectAgentStates (Systems.swift:1592-1611) as a device pack
__global__ void agents_export_kernel(DevCrowd crowd, sge_agent_state* out) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= crowd.count) return;
    const sge_controller_params& P = crowd.params[e];
    const sge_body_state& b = crowd.bodies[e];
    bool solid = (P.agentFlags & SGE_AGENT_PRESENT) && (P.agentFlags & SGE_AGENT_SOLID);
    float radius = (P.agentFlags & SGE_AGENT_RADIUS_OVERRIDE) ? P.agentRadiusOverride : P.radius;
    sge_agent_state a;
    for (int k = 0; k < 3; ++k) { a.position[k] = (float)b.position[k]; a.velocity[k] = (float)b.linearVelocity[k]; }
    a.radius = solid ? radius : -1.0f;
    a.halfHeight = P.halfHeight;
    out[e] = a;
}
__global__ void agents_pad_kernel(sge_agent_state* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sge_agent_state a{};
    a.radius = -1.0f; // not an agent
    out[i] = a;
}
void launch_agents_pad(sge_agent_state* d_out, int n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(agents_pad_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_out, n);
}
void launch_agents_export(const DevCrowd& crowd, sge_agent_state* d_out, hipStream_t s) {
    if (crowd.count <= 0) return;
    // one-wave workgroups: beside a pose launch on the pose stream (16k one-wave workgroups taking every place that comes free) a
    // 256-thread workgroup waited for four free places on one CU and the launch took 57 us for 10,000 agents
    hipLaunchKernelGGL(agents_export_kernel, dim3((crowd.count + kWave - 1) / kWave), dim3(kWave), 0, s, crowd, d_out);
}

} // namespace sge
