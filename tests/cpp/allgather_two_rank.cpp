// sge_agents_allgather with a real multi-rank ncclComm_t, the way a Swift / C++ host of the reference drives config 5:
// one process per GPU, the host owns the communicator, libsge_amd.so does export -> ncclAllGather -> import on its own stream.
//
//   allgather_two_rank RANK WORLD ID_FILE OUT_FILE
//
// Every rank steps its contiguous block of a crowd of 64 characters (two rows that walk into each other; the rows are split so that
// with two ranks every character's opponent lives on the OTHER rank) for 180 fixed steps and writes its sge_body_state block to
// OUT_FILE. WORLD = 1 runs the whole crowd in one process with no communicator: the concatenation of the ranks' files must equal
// that file byte for byte (tests/test_multi_gpu.py). The ncclUniqueId travels from rank 0 to the others through ID_FILE.
//   hipcc tests/cpp/allgather_two_rank.cpp -Iinclude -Lswift-game-engine_amd -lsge_amd -lrccl -Wl,-rpath,$PWD/swift-game-engine_amd -o /tmp/allgather_two_rank
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "sge_amd.h"

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, sge_last_error()); return 1; } } while (0)

static const int kTotal = 64, kSteps = 180;

int main(int argc, char** argv) {
    if (argc != 5) { std::fprintf(stderr, "usage: %s RANK WORLD ID_FILE OUT_FILE\n", argv[0]); return 2; }
    const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
    const std::string idFile = argv[3], outFile = argv[4];
    CHECK(world >= 1 && rank >= 0 && rank < world && kTotal % world == 0);
    int devices = 0;
    CHECK(hipGetDeviceCount(&devices) == hipSuccess && devices >= world);
    CHECK(hipSetDevice(rank) == hipSuccess);

    ncclComm_t comm = nullptr;
    if (world > 1) {
        ncclUniqueId id;
        if (rank == 0) {
            CHECK(ncclGetUniqueId(&id) == ncclSuccess);
            const std::string tmp = idFile + ".tmp";
            FILE* f = std::fopen(tmp.c_str(), "wb");
            CHECK(f && std::fwrite(&id, sizeof(id), 1, f) == 1);
            std::fclose(f);
            CHECK(std::rename(tmp.c_str(), idFile.c_str()) == 0);
        } else {
            FILE* f = nullptr;
            for (int tries = 0; tries < 600 && !(f = std::fopen(idFile.c_str(), "rb")); ++tries) usleep(100000);
            CHECK(f && std::fread(&id, sizeof(id), 1, f) == 1);
            std::fclose(f);
        }
        CHECK(ncclCommInitRank(&comm, world, id, rank) == ncclSuccess);
    }

    sge_context* ctx = sge_context_create(rank);
    CHECK(ctx != nullptr);
    const float quad[12] = {-60, 0, 60, 60, 0, 60, 60, 0, -60, -60, 0, -60};
    const uint32_t quadIdx[6] = {0, 1, 2, 0, 2, 3};
    sge_static_mesh_entity ground{};
    ground.positions = quad; ground.vertexCount = 4; ground.indices = quadIdx; ground.indexCount = 6;
    const float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, -3, 0, 1};
    std::memcpy(ground.modelMatrix, m, sizeof(m));
    ground.material = sge_surface_material{0.9f, 0.8f, 0};
    ground.collisionLayer = 1;
    CHECK(sge_collision_rebuild_static(ctx, &ground, 1) == SGE_OK);

    // the whole crowd is drawn on every rank; a rank uploads its block [first, first + count)
    const int count = kTotal / world, first = rank * count;
    std::vector<sge_body_state> bodies(kTotal);
    std::vector<sge_controller_params> params(kTotal);
    std::vector<sge_controller_state> ctrl(kTotal);
    std::vector<sge_move_intent> intents(kTotal);
    std::vector<sge_locomotion_state> loco(kTotal);
    std::vector<sge_action_state> actions(kTotal);
    for (int i = 0; i < kTotal; ++i) {
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
        const int row = i >= kTotal / 2, col = i % (kTotal / 2);   // row 0 = the first half of the crowd: with two ranks, rank 0
        std::memset(&bodies[i], 0, sizeof(bodies[i]));
        bodies[i].position[0] = -6.0 + 12.0 * row; bodies[i].position[1] = -0.45; bodies[i].position[2] = 4.0 * col - 2.0 * kTotal / 2.0;
        bodies[i].rotation[3] = 1; bodies[i].transformRotation[3] = 1; bodies[i].bodyType = SGE_BODY_DYNAMIC;
        intents[i].desiredVelocity[0] = row ? -4.5f : 4.5f;
        std::memset(&loco[i], 0, sizeof(loco[i]));
        std::memset(&actions[i], 0, sizeof(actions[i]));
    }
    CHECK(sge_characters_resize(ctx, count) == SGE_OK);
    CHECK(sge_characters_upload(ctx, 0, count, bodies.data() + first, params.data() + first, ctrl.data() + first, intents.data() + first,
                                loco.data() + first, actions.data() + first) == SGE_OK);

    const uint32_t pre = SGE_STAGE_INTENT | SGE_STAGE_GRAVITY, rest = SGE_STAGE_MOVE | SGE_STAGE_AGENTS;
    for (int s = 0; s < kSteps; ++s) {
        sge_tick_desc d{};
        d.dt = 1.0f / 60.0f; d.gravity[1] = -98.0f;
        d.stages = pre;
        CHECK(sge_tick(ctx, &d) == SGE_OK);
        CHECK(sge_agents_allgather(ctx, comm, rank, world, count) == SGE_OK);  // stream-ordered: no host synchronisation in this loop
        d.stages = rest;
        CHECK(sge_tick(ctx, &d) == SGE_OK);
    }
    std::vector<sge_body_state> out(count);
    CHECK(sge_characters_download(ctx, 0, count, out.data(), nullptr, nullptr, nullptr, nullptr, nullptr) == SGE_OK);
    int met = 0;
    for (int i = 0; i < count; ++i) if (std::fabs(out[i].position[0]) < 4.0) met += 1;  // the rows stopped each other in the middle
    CHECK(met == count);
    FILE* f = std::fopen(outFile.c_str(), "wb");
    CHECK(f && std::fwrite(out.data(), sizeof(sge_body_state), out.size(), f) == out.size());
    std::fclose(f);
    if (comm) ncclCommDestroy(comm);
    sge_context_destroy(ctx);
    std::printf("allgather rank %d/%d ok: %d of %d agents, %d steps\n", rank, world, count, kTotal, kSteps);
    return 0;
}
