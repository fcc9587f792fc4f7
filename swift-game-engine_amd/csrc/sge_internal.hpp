// Internal declarations of libsge_amd.so (context, device layouts, launchers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <string>
#include <vector>
#include "../../include/sge_amd.h"
#include "sge_math.hpp"

namespace sge {

static_assert(sizeof(sge_body_state) == 96, "layout");
static_assert(sizeof(sge_controller_params) == 64, "layout");
static_assert(sizeof(sge_controller_state) == 128, "layout");
static_assert(sizeof(sge_move_intent) == 32, "layout");
static_assert(sizeof(sge_locomotion_state) == 96, "layout");
static_assert(sizeof(sge_action_state) == 32, "layout");
static_assert(sizeof(sge_agent_state) == 32, "layout");
static_assert(sizeof(sge_capsule_query) == 44, "layout");
static_assert(sizeof(sge_capsule_cast_hit) == 60, "layout");
static_assert(sizeof(sge_capsule_overlap_hit) == 56, "layout");
static_assert(sizeof(sge_bvh_node) == 44, "layout");
static_assert(sizeof(sge_tick_desc) == 32, "layout");

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);
#define SGE_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return ::sge::hip_fail(_e, #expr); } while (0)

// ---- device-side layouts ------------------------------------------------- //

// BVH node as the kernels read it (32 B). Interior: a = left, b = right (>= 0).
// Leaf: a = ~firstSlot (< 0), b = triangle count (1..4).
struct DevNode { float mnx, mny, mnz, mxx, mxy, mxz; int32_t a, b; };
static_assert(sizeof(DevNode) == 32, "layout");

// One collision triangle in triOrder ("slot") order (48 B): world-space vertices,
// collision layer, the reference's triangle index and its DFS visit rank (the
// position at which CollisionQuery's right-child-first traversal reaches it).
struct DevTri { float v0x, v0y, v0z, v1x, v1y, v1z, v2x, v2y, v2z; uint32_t layer; int32_t triIndex; int32_t rank; };
static_assert(sizeof(DevTri) == 48, "layout");

struct DevMaterial { float muS, muK; uint32_t flatten; };

// Skeleton constants in device memory (SoA over bones).
struct DevSkeleton {
    int boneCount;
    int pelvisIndex, leanIndex, leanParent;
    int maxDepth;          // deepest hierarchy level
    int leanChainLen;      // ancestors of leanIndex, root first, leanIndex last
    float unitScale;
    const int32_t* parent;      // [B]
    const int32_t* depth;       // [B]
    const int32_t* leanChain;   // [leanChainLen]
    const int32_t* path;        // [B][maxDepth + 1]: bone i's ancestors root first, i itself at position depth[i]
    const int32_t* slotBone;    // [B] the order in which pose_kernel's lanes take the bones: slot s (pass s / 64) is bone slotBone[s]
    int lastPassStatic;         // 1: no uploaded profile has an entry for any bone of the last pass
    int extraParentReady;       // 1: the parent of every bone in a pass after the first sits in an earlier pass
    const float* bindLocal;     // [B][12] affine by column
    const float* invBind;       // [B][16] full 4x4 used for the palette (mesh re-bind applied at upload)
    const float* restT;         // [B][3]
    const float* rawRestT;      // [B][3]
    const float* preRot;        // [B][12] rotationXYZDegrees(preRotationDegrees[i]) (c3 = 0), bone 0 NOT yet rootFix'ed
    float rootFix[12];          // rotation part of rootRotationFix
};

struct DevProfiles {
    int count;
    int stride;                 // floats per (bone,axis) = max coefficient count over everything uploaded
    int maxOrder;               // largest Fourier order uploaded (selects the kernel instantiation)
    uint32_t nonRootTranslation; // bit p: profile p carries translation for some bone other than bone 0
    int32_t order[SGE_MAX_PROFILES];
    float cycleRaw[SGE_MAX_PROFILES];
    const float* coeffs;        // [P][B][6][stride]
    const uint8_t* coeffCount;  // [P][B][6]
    const uint8_t* bonePresent; // [P][B]
};

// Wide BVH the kernels traverse: every wide node is kWideWidth consecutive DevNode entries (2 KB) — a treelet cut
// out of the reference's binary tree. Entry: bounds of one binary subtree + either a = index of the child wide
// node (b = 0) or a = ~firstSlot, b = triangle count (<= kWideWidth) of a contiguous slot range. Unused entries
// carry an inverted box. One wavefront tests a whole wide node per step, one entry per lane.
constexpr int kWideWidth = 64;

struct DevCollision {
    int nodeCount, triCount, root;  // root < 0: no triangle in either set
    const DevNode* wide;            // [wideCount][kWideWidth]; wide node 0 is the root of the first non-empty set
    int wideCount;
    int dynWide;                    // wide node of the dynamic set's root when both sets hold triangles, else -1
    const DevTri* tris;             // slot order: static set, then dynamic set (triIndex / rank offset by the static count)
    const DevMaterial* materials;   // by (offset) triIndex
    // the binary BVH of each set (leaf: a = ~firstSlot, b = count; inner: a = left, b = right) for the raycast, which
    // follows the reference's own traversal order; binRoot < 0: empty set; binSlotBase: the set's first slot in `tris`
    const DevNode* binNodes[2];
    int binRoot[2], binSlotBase[2];
    const int* slotOfRank;          // slot in `tris` of the triangle with a given (offset) visit rank
};

// What the animation stages read of a character's body and controller, copied out by the move stage's write-back (16 dwords): with it
// pose(n) needs nothing that move(n+1) writes and runs BESIDE move(n+1), on a stream of its own, instead of in front of it
// (overlap mode, sge_tick). The dwords are body words 6..15 (linearVelocity, rotation), controller words 0..2 (groundNormal) and the
// controller's flags / groundDistance pair.
struct PoseInput {
    double linearVelocity[3];
    float rotation[4];
    float groundNormal[3];
    uint32_t flags;
    float groundDistance;
    uint32_t pad;
};
static_assert(sizeof(PoseInput) == 64, "one 64-byte line per character");
static_assert(offsetof(sge_body_state, linearVelocity) == 24 && offsetof(sge_body_state, rotation) == 48 &&
              offsetof(sge_body_state, transformRotation) == 64, "PoseInput copies body words 6..15; words 16.. are not the move stage's");
static_assert(offsetof(sge_controller_state, groundNormal) == 0 &&
              offsetof(sge_controller_state, groundDistance) == offsetof(sge_controller_state, flags) + 4, "PoseInput controller words");
constexpr int kCtrlFlagsWord = (int)(offsetof(sge_controller_state, flags) / 4);
constexpr int kBodyMoveWords = 16; // the words of sge_body_state the move / separation kernels store (transformRotation is the pose stage's)

struct DevCrowd {
    int count;
    sge_body_state* bodies;
    sge_controller_params* params;
    sge_controller_state* controllers;
    sge_move_intent* intents;
    sge_locomotion_state* locomotion;
    sge_action_state* actions;
    float* palettes;   // [N][B][16]
    float* poseModel;  // [N][B][16] (debug) or null
    float* poseLocal;  // [N][B][16] (debug) or null
    PoseInput* poseIn; // [N] or null. Move stage: written with every body / controller write-back. Animation stages: read INSTEAD of
                       // bodies / controllers (null: read those).
};

struct DevMesh {
    int vertexCount;
    const float* positions;       // [V][3]
    const float* normals;         // [V][3]
    const float* tangents;        // [V][4]
    const uint16_t* boneIndices;  // [V][4]
    const float* boneWeights;     // [V][4]
};

// parameters of the agent grid, computed ON THE DEVICE every step (no host round trip inside sge_tick)
struct AgentGrid {
    float originX, originZ, invCell;
    int nx, nz;                 // nx == 0: no solid agent in the snapshot
    float maxRadius, maxSpeed;  // over the solid agents of the snapshot (bounds the XZ reach of a sweep)
    int cells;
};
constexpr int kAgentMaxCells = 1 << 20;

struct DevAgents {
    const sge_agent_state* all; // gathered snapshot
    int total, selfOffset;
    // uniform XZ grid over the snapshot
    const AgentGrid* grid;      // device
    const int32_t* cellStart;   // [cells+1]
    const int32_t* cellItems;   // [total] agent indices sorted by cell
};

// ---- host-side collision build (sge_host.cpp) ---------------------------- //
struct HostBVHNode { float mn[3], mx[3]; int left, right, start, count, parent; };
struct HostMeshSlice { int vertexBegin = 0, vertexEnd = 0, indexBegin = 0, indexEnd = 0, triBegin = 0, triEnd = 0; bool valid = false; };
struct HostCollision {
    std::vector<float> positions;   // xyz, world space
    std::vector<float> localPositions; // xyz as given (collisionMesh.streams.positions), same indexing as `positions`
    std::vector<HostMeshSlice> slices; // per entity of the last rebuild (CollisionQuery.swift:480-485)
    std::vector<uint32_t> indices;
    std::vector<float> aabbs;       // [T][6] min xyz max xyz
    std::vector<sge_surface_material> materials;
    std::vector<uint32_t> layers;
    std::vector<HostBVHNode> nodes;
    std::vector<int> triOrder, triLeaf, rank; // rank[tri]
    int root = -1;
    int maxDepth = 0;
    std::vector<DevNode> wide;      // flattened wide nodes, kWideWidth entries each
    std::vector<int> wideBinary;    // binary node behind every wide entry (-1: padding), to re-bound after a refit
    int wideLevels = 0;             // depth of the wide tree (a depth-first traversal keeps <= 63 siblings pending per level)
    void rebuild(const sge_static_mesh_entity* ents, int count);
    void buildWide();
    // TriangleMeshSet.updateTransforms + BVH.refit (CollisionQuery.swift:419-462, 528-575); returns #updated triangles
    int updateTransforms(const int32_t* entities, const float* modelMatrices, int n);
    void refit(const std::vector<int>& updatedTriangles);
    void reboundWide();
};

// ---- skinned-geometry acceleration structure (sge_host.cpp builds, sge_blas.hip refits / traverses) ---- //
// Topology shared by every clone of the mesh; see include/sge_amd.h for the meaning of the arrays.
struct HostBlas {
    int triCount = 0, vertexCount = 0, clusterCount = 0, levels = 0;
    std::vector<int> entryLink;          // [E][2]
    std::vector<int> wideFirst;          // [W + 1]
    std::vector<int> wideParentEntry;    // [W]
    std::vector<int> wideLevelStart;     // [levels + 1]: wide nodes are numbered level by level (breadth first)
    std::vector<uint32_t> slotTriangle;  // [T]
    std::vector<uint32_t> slotIndices;   // [T][3]: vertex indices of the triangle at every slot
    std::vector<int> vertexEntryStart, vertexEntries; // CSR vertex -> clusters
    // refit schedule: the vertex stream is cut into tiles of tileVerts vertices; a chunk = up to 16 vertices of ONE tile
    // that belong to ONE cluster, the work of ONE lane; 64 chunks of similar length make a round, the work of one wavefront
    int tileVerts = 0, tileCount = 0, chunkCount = 0, tileCap = 4096;
    std::vector<int> tileRoundStart;     // [tileCount + 1]
    std::vector<int> roundLen;           // [roundCount] vertices per chunk in this round (1..16; shorter chunks repeat their first vertex)
    std::vector<int> roundCluster;       // [roundCount][64] leaf entry of every lane's chunk (a short last round repeats its first chunk) | roundLen << 24
    std::vector<uint32_t> roundIds;      // [roundCount][64][8]: word j of a lane = LDS byte offsets (4 * local vertex id) of its vertices 2j | 2j+1 << 16
                                         // (a lane's eight words are contiguous: two 16-byte loads; entries past the round's length repeat the lane's first vertex)
    int entryCount() const { return (int)entryLink.size() / 2; }
    int wideCount() const { return (int)wideParentEntry.size(); }
    // false + message when the mesh cannot be handled
    bool build(const float* positions, int vertexCount, const uint32_t* indices, int indexCount, std::string& err);
};

struct DevBlas {
    int entryCount, wideCount, triCount, vertexCount, clusterCount, levels;
    const int2* entryLink;
    const int* wideFirst;
    const int* wideParentEntry;
    const int* wideLevelStart;
    const uint32_t* slotIndices;
    const uint32_t* slotTriangle;
    int tileVerts, tileCount, tileCap;
    const int* tileRoundStart;
    const int* roundLen;
    const int* roundCluster;
    const uint32_t* roundIds;
};
#ifndef SGE_BLAS_BLOCK
#define SGE_BLAS_BLOCK 512
#endif
constexpr int kBlasRefitBlock = SGE_BLAS_BLOCK;
constexpr int kBlasTileVerts = 4096; // most vertices staged in LDS at a time (48 KB): kBlasRefitBlock threads x 8
// LDS capacity of the tile: the largest of these that lets three workgroups share a CU's 160 KB beside their box tables,
// else 4096 with two (HostBlas::build picks; the kernel is instantiated per capacity)
constexpr int kBlasTileSteps[4] = {4096, 3072, 2048, 1024};
// LDS of the refit kernel: the box table, six floats per row, (entryCount + 1) rows, + one tile of positions (SoA) + tileRoundStart
inline size_t blasRefitLdsBytes(int entryCount, int tileCount, int tileCap = kBlasTileVerts) { return (size_t)(entryCount + 1) * 24 + (size_t)tileCap * 12 + (size_t)(tileCount + 1) * 4; }
// + the shape of the wide tree, staged once per workgroup (wideLevelStart[levels + 1], wideFirst[wideCount + 1], wideParentEntry[wideCount]):
// the end-of-character reduction reads it from LDS — a global load there queues behind the position loads already in flight for the
// next tile (loads return in order) and took a third of the kernel
inline size_t blasTopoBytes(int wideCount, int levels) { return (size_t)(levels + 2 * wideCount + 2) * 4 + 16; }
constexpr size_t kBlasMaxLdsBytes = 144 * 1024; // of the CU's 160 KB

// ---- kernel launchers ----------------------------------------------------- //
struct MoveLaunch {
    DevCrowd crowd; DevCollision col; DevAgents agents;
    float dt; float gx, gy, gz; uint32_t stages; int first, count;
    unsigned long long* stats; // [8]
    void* scratch;             // [count] x kMoveScratchBytes: per-character working set between the two launches
    const sge_platform_state* platforms; // PlatformCarry inputs of this step
    int platformCount;
    // light / heavy split of part 1 (sge_ccd.hip, "heavy characters"); lists == nullptr: one launch over [first, first+count)
    int* cost;                 // [crowd.count] distance evaluations of each character's last step
    uint8_t* hint;             // [crowd.count] 1 when the last step needed the four offset ground casts
    int* lists; int* listCounts; // heavy list at lists[count..], its length at listCounts[1] (classify_kernel)
    uint8_t* heavyFlags;         // [crowd.count] 1: taken by the multi-wave launch this step
    int heavyThreshold, heavyCap;
    hipStream_t heavyStream; hipEvent_t evClassified, evHeavyDone; // second stream of the move stage (the grouped launch runs there)
    int* heavyDemandHost;      // pinned int[2]: characters above the threshold, and the sum of all costs, of the step last copied back; or null
    const int* list; const int* listCount; // what a part-1 launch iterates over (set by launch_move)
    const int* order; const int* orderCount; // grouped launch: characters sorted by last step's cost (device), its length
    int* orderHist;                        // [64] histogram / cursors of the order list (zero between steps)
    int solo;                              // grouped launch: its first `solo` wavefronts take one character each (the most expensive)
    int listsReady, nextHeavyCap;          // the heavy / order lists of this step were built behind the previous one; cap for the next
    hipEvent_t evListsReady;
    // diagnostics (SGE_WAVE_PROF=1), rows of 8 x u64 in three regions of crowd.count rows each: [0] one row per wavefront of
    // move_group_kernel, [1] one row per character of move_kernel<0>, [2] one row per character of pose_kernel
    unsigned long long* waveProf;
    int listsPending;                      // an earlier launch_move left a list build in flight on heavyStream (evListsReady marks its end)
};
// Per-device launch state: a process may hold contexts on several GPUs (sge_context_create(device_index)), and function attributes
// and the CU count belong to the device that is current at the launch (every entry point calls hipSetDevice(ctx->device) first).
constexpr int kMaxDevices = 32;
inline int currentDeviceSlot() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev < 0 || dev >= kMaxDevices ? 0 : dev;
}
inline int currentDeviceCUs() {
    static int cus[kMaxDevices] = {};
    const int d = currentDeviceSlot();
    if (!cus[d]) {
        int n = 256;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d);
        cus[d] = n > 0 ? n : 256;
    }
    return cus[d];
}
constexpr int kMoveScratchBytes = 256;
constexpr int kTraversalStackCap = 256; // LDS stack of pending wide nodes per query (sge_ccd.hip)
constexpr int kStatShards = 256; // device counters: kStatShards lines of 8 x u64
// Overlap mode: crowds from this size on skin with RESIDENT workgroups (kResidentSkinQuarters / 4 per CU, drawing work
// units of kResidentSkinCharsPerUnit characters from a ticket counter); smaller crowds with workgroups that come and go. Measured with
// the pose launch beside the next move stage (profiles/r3_schedule_sweeps.txt, block 15): 5,000 characters 0.534 (come and go) against 0.554 ms
// on the cheese mesh and 0.68 against 0.74 on the synthetic terrain; 6,500 characters 0.72 against 0.62 and 0.85 against 0.78;
// 10,000 on the synthetic terrain 1.35-1.42 against 1.13-1.17 (the rule that kept collision-heavy steps with the come-and-go form
// dates from the schedule with pose in front of the next move stage and is gone).
constexpr int kResidentSkinCharacters = 6000;
constexpr int kResidentSkinQuarters = 6;
// with character-vs-character sweeps the cast launches are the longer side of the step and one resident LBS workgroup per CU leaves them
// more room: 10k characters, quarters 3 / 4 / 5 / 6 / 8: 1.13 / 1.06 / 1.10 / 1.14 / 1.18 ms per step (profiles/r4_agents_kernels.txt)
constexpr int kResidentSkinQuartersWithAgents = 4;
constexpr int kResidentSkinCharsPerUnit = 4;
// returns true when the lists of the NEXT step (same range, threshold, cap = nextHeavyCap) have been enqueued behind this stage
bool launch_move(const MoveLaunch& L, hipStream_t s);
constexpr size_t kSeparationAgentBytes = 56; // SepAgentDev (sge_ccd.hip)
// agentScratch: crowd.count x kSeparationAgentBytes; flowScratch: separationFlowBytes(crowd.count) (used above SGE_MAX_SEPARATION_AGENTS)
size_t separationFlowBytes(int count);
size_t separationFlowControlOffset(int count);
// reach: 2 (the rule) or 3, see SepFlow::reach; flagsHost: pinned int[2] the stage copies {pushed further than a cell, pass redone} to, or null
void launch_separation(const DevCrowd& crowd, const DevCollision& col, int iterations, float separationMargin, float heightMargin,
                       void* agentScratch, int* counts, void* flowScratch, hipStream_t s, int reach = 2, int* flagsHost = nullptr);
void launch_cast_queries(const DevCollision& col, const sge_capsule_query* d_q, int n, sge_capsule_cast_hit* d_out,
                         unsigned long long* stats, hipStream_t s);
void launch_overlap_queries(const DevCollision& col, const sge_capsule_query* d_q, int n, int maxHits,
                            sge_capsule_overlap_hit* d_out, int32_t* d_counts, unsigned long long* stats, hipStream_t s);
void launch_raycast_queries(const DevCollision& col, const sge_ray_query* d_q, int n, sge_raycast_hit* d_out, hipStream_t s);
void launch_overlap_deepest_queries(const DevCollision& col, const sge_capsule_query* d_q, int n,
                                    sge_capsule_overlap_hit* d_out, int32_t* d_found, unsigned long long* stats, hipStream_t s);

struct PoseLaunch {
    DevCrowd crowd; DevSkeleton sk; DevProfiles prof;
    float dt; uint32_t stages; int first, count;
    unsigned long long* waveProf; // diagnostics: region [2] of MoveLaunch::waveProf, or null
};
void launch_pose(const PoseLaunch& L, hipStream_t s);

struct SkinLaunch {
    const void* srcPos; const void* srcNrm; const void* srcTan; const void* srcIdx; const void* srcWgt;
    const float* palettes;      // [chars][paletteCount][16]
    int paletteCount, vertexCount, chars;
    long long dstBaseVertex;    // of the first character; character c writes at dstBaseVertex + c*vertexCount
    int srcLayout, dstLayout;
    void* outPos; void* outNrm; void* outTan;
};
// residentQueue + residentQuarters > 0: the resident form (quarters of a workgroup per CU, one device int as the ticket counter)
// charsPerUnit: 1, 2, 4 or 8 characters share every loaded source vertex in the resident form
// returns the form the launch took: 0 = workgroups that come and go, n >= 1 = resident workgroups with n characters per work unit
int launch_skin(const SkinLaunch& L, hipStream_t s, int maxWorkgroupsPerCU = 0, int* residentQueue = nullptr, int residentQuarters = 0, int charsPerUnit = 1);
// one record per RTSkinningJob of a batched encode (device copy)
struct SkinJobDev {
    const void* srcPos; const void* srcNrm; const void* srcTan; const void* srcIdx; const void* srcWgt;
    const float* palette;
    int paletteCount, vertexCount;
    long long dstBaseVertex;
    int srcStride, pad;
};
void launch_skin_jobs(const SkinJobDev* d_jobs, const int2* d_blockJob, int blocks, int vertsPerBlock, int dstLayout,
                      void* outPos, void* outNrm, void* outTan, hipStream_t s);
void launch_store_probe(void* outPos, void* outNrm, void* outTan, int chars, int vertexCount, int dstLayout, hipStream_t s);

void launch_agents_export(const DevCrowd& crowd, sge_agent_state* d_out, hipStream_t s);
void launch_agents_pad(sge_agent_state* d_out, int n, hipStream_t s);

// boxes of `chars` characters: character k reads vertices [firstVertex + k * vertexCount, +vertexCount) of `positions`
// (layout SGE_LAYOUT_*) and writes bounds[k][entryCount + 1][6]
int launch_blas_refit(const DevBlas& B, const void* positions, int layout, long long firstVertex, int chars, float* bounds, int* queue, hipStream_t s); // queue: one device int
// skin stage + refit stage as one launch (SGE_OPT_FUSE_BLAS_REFIT): bounds[chars][entryCount + 1][6]; queue = one device int
// (the kernel's ticket counter, zeroed on `s` by the launcher)
int launch_skin_refit(const SkinLaunch& L, const DevBlas& B, float* bounds, int* queue, hipStream_t s, int maxWorkgroupsPerCU);
struct BlasTrace {
    DevBlas blas;
    const void* positions; const void* normals; const float* tangents; // skinned streams of the crowd
    const uint32_t* indices; // [T][3] in primitive order (the item's slice of dynamicIndexBuffer)
    int layout;
    const float* bounds;     // [chars][entryCount + 1][6]
    const float* instances;  // [chars][16] model matrices
    int chars;
    const float* worldBoxes; // [chars + ceil(chars / 64)][6]: per instance, then per 64 consecutive instances; filled by
                             // launch_blas_intersect when a ray asks for every instance
    const float* uvs;        // [V][2] of the shared mesh, or null
    int worldBoxesValid;     // 0: this launch did not refresh worldBoxes — a ray with instance < 0 reports a miss instead of walking
                             // boxes that are missing or belong to an earlier frame (the host cannot inspect device-resident rays)
    // The instance level of a ray that names no character (RTAccelerationBuilder.swift:168-185: the TLAS over all items), rebuilt
    // with the world boxes: the instances sorted by the cell of an XZ grid over their centres (the agent grid's kernels,
    // sge_api.hip), `instOrder`; one box per 64 consecutive entries of that order (`groupBoxes`), one per 64 groups (`superBoxes`).
    // Null: the flat form (groups of 64 consecutive character indices, boxes behind the instances' in worldBoxes).
    const int* instOrder;    // [chars]
    const float* groupBoxes; // [ceil(chars / 64)][6]
    const float* superBoxes; // [ceil(chars / 4096)][6]
};
constexpr int kBlasTraversalStackCap = 256; // pending wide nodes of one closest-hit query (sge_blas.hip); sge_blas_build checks it
void launch_blas_intersect(BlasTrace T, const sge_blas_ray* d_rays, int n, sge_blas_hit* d_hits, bool anyInstance, hipStream_t s);
// the per-frame part of the instance level: every character's world box (+ the flat form's boxes of 64 consecutive indices)
void launch_blas_world_boxes(const BlasTrace& T, float* worldBoxes, hipStream_t s);
// centre + half diagonal of every world box as an agent record (what the XZ grid kernels bin)
void launch_blas_instance_points(const float* worldBoxes, int chars, sge_agent_state* out, hipStream_t s);
// boxes of the groups of 64 consecutive entries of `order`, and of 64 consecutive groups
void launch_blas_group_boxes(const float* worldBoxes, const int* order, int chars, float* groupBoxes, float* superBoxes, hipStream_t s);

} // namespace sge
