// The character-vs-character exchange as a C / C++ / Swift host would drive it: one process per GPU, the host owns the
// ncclComm_t, libsge_amd.so does export -> ncclAllGather -> import on its own stream (sge_agents_allgather). This program runs the
// sequence on ONE GPU with a real single-rank RCCL communicator (ncclCommInitRank with nranks = 1: the collective executes, the
// data path is the production one) and checks it against the exchange-free form (export + import of the same buffer).
//   hipcc tests/cpp/allgather_smoke.cpp -Iinclude -Lswift-game-engine_amd -lsge_amd -lrccl -Wl,-rpath,$PWD/swift-game-engine_amd -o /tmp/allgather_smoke
// With N ranks the only differences are: every rank passes its own `rank`, `world_size = N`, `slot = max_r count_r`, and the
// ncclUniqueId travels from rank 0 to the others (MPI_Bcast, a file, torch's store ...) before ncclCommInitRank.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "sge_amd.h"

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, sge_last_error()); return 1; } } while (0)

static void settle_crowd(sge_context* ctx, int n, std::vector<sge_body_state>& bodies) {
    std::vector<sge_controller_params> params(n);
    std::vector<sge_controller_state> ctrl(n);
    std::vector<sge_move_intent> intents(n);
    std::vector<sge_locomotion_state> loco(n);
    std::vector<sge_action_state> actions(n);
    bodies.assign(n, sge_body_state{});
    for (int i = 0; i < n; ++i) {
        sge_controller_params& p = params[i];
        std::memset(&p, 0, sizeof(p));
        p.radius = 1.5f; p.halfHeight = 1.0f; p.skinWidth = 0.3f; p.groundSnapSkin = 0.05f; p.snapDistance = 0.8f; p.fallProbeDistance = 200.0f;
        p.groundSnapMaxSpeed = 5.0f; p.groundSnapMaxToi = 0.1f; p.groundSnapMaxStep = 0.1f; p.groundSweepMaxStep = 0.1f;
        p.maxSlideIterations = 4; p.minGroundDot = 0.5f; p.collisionMask = 0xFFFFFFFFu;
        p.agentFlags = SGE_AGENT_PRESENT | SGE_AGENT_SOLID; p.agentMassWeight = 1.0f;
        std::memset(&ctrl[i], 0, sizeof(ctrl[i]));
        ctrl[i].groundNormal[1] = 1.0f; ctrl[i].groundTriangleIndex = -1; ctrl[i].groundDistance = 3.4e38f;
        std::memset(&intents[i], 0, sizeof(intents[i]));
        intents[i].flags = SGE_INTENT_PRESENT; intents[i].maxAcceleration = 20.0f; intents[i].maxDeceleration = 36.0f;
        // two rows walking into each other
        const int row = i & 1, col = i >> 1;
        bodies[i].position[0] = -6.0 + 12.0 * row; bodies[i].position[1] = -0.45; bodies[i].position[2] = 4.0 * col - 2.0 * n / 4.0;
        bodies[i].rotation[3] = 1; bodies[i].transformRotation[3] = 1; bodies[i].bodyType = SGE_BODY_DYNAMIC;
        intents[i].desiredVelocity[0] = row ? -4.5f : 4.5f;
        std::memset(&loco[i], 0, sizeof(loco[i]));
        std::memset(&actions[i], 0, sizeof(actions[i]));
    }
    if (sge_characters_resize(ctx, n) != SGE_OK) return;
    sge_characters_upload(ctx, 0, n, bodies.data(), params.data(), ctrl.data(), intents.data(), loco.data(), actions.data());
}

static int run(sge_context* ctx, void* comm, bool through_library, int n, int steps, std::vector<sge_body_state>& out) {
    std::vector<sge_body_state> bodies;
    settle_crowd(ctx, n, bodies);
    void* dLocal = nullptr;
    if (!through_library && hipMalloc(&dLocal, (size_t)n * sizeof(sge_agent_state)) != hipSuccess) return 1;
    const uint32_t pre = SGE_STAGE_INTENT | SGE_STAGE_GRAVITY, rest = SGE_STAGE_MOVE | SGE_STAGE_AGENTS;
    for (int s = 0; s < steps; ++s) {
        sge_tick_desc d{};
        d.dt = 1.0f / 60.0f; d.gravity[1] = -98.0f;
        d.stages = pre;
        CHECK(sge_tick(ctx, &d) == SGE_OK);
        if (through_library) {
            CHECK(sge_agents_allgather(ctx, comm, 0, 1, n) == SGE_OK);   // export -> ncclAllGather -> import, all on the context's stream
        } else {
            CHECK(sge_agents_export(ctx, dLocal) == SGE_OK);
            CHECK(sge_agents_import(ctx, dLocal, n, 0) == SGE_OK);
        }
        d.stages = rest;
        CHECK(sge_tick(ctx, &d) == SGE_OK);
    }
    out.resize(n);
    CHECK(sge_characters_download(ctx, 0, n, out.data(), nullptr, nullptr, nullptr, nullptr, nullptr) == SGE_OK);
    if (dLocal) (void)hipFree(dLocal);
    return 0;
}

int main() {
    sge_context* ctx = sge_context_create(0);
    CHECK(ctx != nullptr);
    const float quad[12] = {-40, 0, 40, 40, 0, 40, 40, 0, -40, -40, 0, -40};
    const uint32_t quadIdx[6] = {0, 1, 2, 0, 2, 3};
    sge_static_mesh_entity ground{};
    ground.positions = quad; ground.vertexCount = 4; ground.indices = quadIdx; ground.indexCount = 6;
    const float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, -3, 0, 1};
    std::memcpy(ground.modelMatrix, m, sizeof(m));
    ground.material = sge_surface_material{0.9f, 0.8f, 0};
    ground.collisionLayer = 1;
    CHECK(sge_collision_rebuild_static(ctx, &ground, 1) == SGE_OK);

    // a real communicator of one rank; the stream the collective runs on is the context's own
    ncclUniqueId id;
    CHECK(ncclGetUniqueId(&id) == ncclSuccess);
    ncclComm_t comm = nullptr;
    CHECK(ncclCommInitRank(&comm, 1, id, 0) == ncclSuccess);
    void* stream = nullptr;
    CHECK(sge_context_get_stream(ctx, &stream) == SGE_OK && stream != nullptr);

    const int n = 24, steps = 180;
    std::vector<sge_body_state> viaNccl, direct, viaNull;
    CHECK(run(ctx, comm, true, n, steps, viaNccl) == 0);
    // With one rank the library has nothing to gather and calls no collective; the collective itself is executed here on the very
    // layout the library uses (in place: this rank's records at recvbuff + rank * slot), on the context's stream:
    {
        std::vector<sge_agent_state> host(n);
        void* dBuf = nullptr;
        CHECK(hipMalloc(&dBuf, (size_t)n * sizeof(sge_agent_state)) == hipSuccess);
        CHECK(sge_agents_export(ctx, dBuf) == SGE_OK);
        CHECK(ncclAllGather(dBuf, dBuf, (size_t)n * sizeof(sge_agent_state), ncclChar, comm, (hipStream_t)stream) == ncclSuccess);   // in place, rank 0 of 1
        CHECK(hipStreamSynchronize((hipStream_t)stream) == hipSuccess);
        CHECK(hipMemcpy(host.data(), dBuf, host.size() * sizeof(sge_agent_state), hipMemcpyDeviceToHost) == hipSuccess);
        for (int i = 0; i < n; ++i) CHECK(host[i].radius == 1.5f && std::fabs(host[i].position[0] - (float)viaNccl[i].position[0]) < 1e-6f);
        (void)hipFree(dBuf);
    }
    CHECK(run(ctx, nullptr, false, n, steps, direct) == 0);
    CHECK(run(ctx, nullptr, true, n, steps, viaNull) == 0);
    int met = 0;
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) {
            CHECK(viaNccl[i].position[k] == direct[i].position[k] && viaNccl[i].position[k] == viaNull[i].position[k]);
            CHECK(viaNccl[i].linearVelocity[k] == direct[i].linearVelocity[k]);
        }
        // the rows met in the middle and stopped each other: nobody walked through
        if (std::fabs(direct[i].position[0]) < 4.0) met += 1;
    }
    CHECK(met == n);
    CHECK(sge_agents_allgather(ctx, nullptr, 0, 2, n) != SGE_OK);   // two ranks need a communicator
    CHECK(sge_agents_allgather(ctx, comm, 0, 1, n - 1) != SGE_OK);  // slot below the character count
    ncclCommDestroy(comm);
    sge_context_destroy(ctx);
    std::printf("allgather smoke ok: %d agents, %d steps\n", n, steps);
    return 0;
}
