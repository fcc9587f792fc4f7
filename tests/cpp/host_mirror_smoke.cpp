// Exercises the C++ host mirror (swift-game-engine_amd/host/sge_host.hpp) against the C ABI on a real GPU:
// CollisionQuery (cast / ground cast / overlap / overlapAll / raycast / updateDynamicTransforms) and a character dropped on the
// demo's ground quad through KinematicMoveStopSystem (config 1 of SURVEY.md §8d: settles at y = -3 + 2.5 + 0.05).
//   g++ -std=c++17 tests/cpp/host_mirror_smoke.cpp -Iinclude -Lswift-game-engine_amd -lsge_amd -Wl,-rpath,$PWD/swift-game-engine_amd -o /tmp/host_mirror_smoke
#include <cmath>
#include <cstdio>
#include <cstring>
#include "../../swift-game-engine_amd/host/sge_host.hpp"

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

static sge_static_mesh_entity entity(const float* pos, int nv, const uint32_t* idx, int ni, float tx, float ty, float tz, uint32_t layer) {
    sge_static_mesh_entity e{};
    e.positions = pos; e.vertexCount = nv; e.indices = idx; e.indexCount = ni;
    const float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, tx, ty, tz, 1};
    std::memcpy(e.modelMatrix, m, sizeof(m));
    e.material = sge_surface_material{0.9f, 0.8f, 0};
    e.collisionLayer = layer;
    return e;
}

int main() {
    auto world = sge::World::make(0);
    CHECK(world != nullptr);
    // ProceduralMeshes.plane(size: 80) at y = -3 (DemoScene.swift:103-130)
    const float quad[12] = {-40, 0, -40, 40, 0, -40, 40, 0, 40, -40, 0, 40};
    const uint32_t quadIdx[6] = {0, 1, 2, 0, 2, 3};
    const float box[24] = {-2, -0.5f, -2, 2, -0.5f, -2, 2, 0.5f, -2, -2, 0.5f, -2, -2, -0.5f, 2, 2, -0.5f, 2, 2, 0.5f, 2, -2, 0.5f, 2};
    const uint32_t boxIdx[36] = {0, 2, 1, 0, 3, 2, 4, 5, 6, 4, 6, 7, 0, 1, 5, 0, 5, 4, 3, 6, 2, 3, 7, 6, 0, 4, 7, 0, 7, 3, 1, 2, 6, 1, 6, 5};
    std::vector<sge_static_mesh_entity> statics{entity(quad, 4, quadIdx, 6, 0, -3, 0, 1)};
    std::vector<sge_static_mesh_entity> dynamics{entity(box, 8, boxIdx, 36, 10, 0, 0, 2)};
    sge::CollisionQuery query(*world, statics, dynamics);

    auto down = query.capsuleCastGround({0, 5, 0}, {0, -20, 0}, 1.5f, 1.0f, 0.5f);
    CHECK(down.has_value());
    CHECK(std::fabs(down->toi - (5 + 3 - 2.5f)) < 2e-3f && down->triangleIndex < 2 && down->normal.y > 0.99f);
    CHECK(!query.capsuleCast({0, 5, 0}, {0, -1, 0}, 1.5f, 1.0f).has_value());               // too short: nil
    auto onBox = query.capsuleCast({10, 6, 0}, {0, -20, 0}, 1.5f, 1.0f);
    CHECK(onBox.has_value() && onBox->triangleIndex >= 2 && std::fabs(onBox->toi - (6 - 0.5f - 2.5f)) < 2e-3f);
    CHECK(query.capsuleCast({10, 6, 0}, {0, -20, 0}, 1.5f, 1.0f, /*mask*/ 1)->triangleIndex < 2);   // the box's layer masked out
    CHECK(query.capsuleOverlapAll({0, -3 + 2.2f, 0}, 1.5f, 1.0f).size() == 2);
    CHECK(query.capsuleOverlapAll({0, -3 + 2.2f, 0}, 1.5f, 1.0f, 1).size() == 1);
    CHECK(query.capsuleOverlap({0, 30, 0}, 1.5f, 1.0f) == std::nullopt ? true : false);
    auto deepest = query.capsuleOverlap({0, -3 + 2.2f, 0}, 1.5f, 1.0f);
    CHECK(deepest.has_value() && std::fabs(deepest->depth - 0.3f) < 1e-4f);
    auto ray = query.raycast({10, 5, 0}, {0, -1, 0}, 100.0f);
    CHECK(ray.has_value() && std::fabs(ray->distance - 4.5f) < 1e-5f && ray->normal.y > 0.99f && ray->triangleIndex >= 2);
    // the platform moves 3 units in +x: the same ray now falls through to the ground
    query.updateDynamicTransforms({0}, {std::array<float, 16>{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 13, 0, 0, 1}});
    ray = query.raycast({10, 5, 0}, {0, -1, 0}, 100.0f);
    CHECK(ray.has_value() && std::fabs(ray->distance - 8.0f) < 1e-5f && ray->triangleIndex < 2);
    std::vector<sge_capsule_query> batch(3, sge_capsule_query{{0, 5, 0}, {0, -20, 0}, 1.5f, 1.0f, 0.5f, 0xFFFFFFFFu, SGE_CAST_GROUND});
    std::vector<sge_capsule_cast_hit> hits;
    query.capsuleCastBatch(batch, hits);
    CHECK(hits.size() == 3 && hits[0].hit && hits[2].toi == hits[0].toi);

    // one character: CharacterFactory defaults (CharacterFactory.swift:77-91), dropped from y = 7.5
    world->resize(1);
    sge_body_state body{};
    body.position[1] = 7.5; body.rotation[3] = 1; body.transformRotation[3] = 1; body.bodyType = SGE_BODY_DYNAMIC;
    sge_controller_params p{};
    p.radius = 1.5f; p.halfHeight = 1.0f; p.skinWidth = 0.3f; p.groundSnapSkin = 0.05f; p.snapDistance = 0.8f; p.fallProbeDistance = 200.0f;
    p.groundSnapMaxSpeed = 8.0f; p.groundSnapMaxToi = 0.2f; p.groundSnapMaxStep = 0.1f; p.groundSweepMaxStep = 0.1f;
    p.maxSlideIterations = 4; p.minGroundDot = 0.5f; p.collisionMask = 0xFFFFFFFFu;
    sge_controller_state c{};
    c.groundNormal[1] = 1; c.groundTriangleIndex = -1;
    sge_move_intent intent{};
    sge_locomotion_state loco{};
    sge_action_state action{};
    world->upload(0, 1, &body, &p, &c, &intent, &loco, &action);
    sge::KinematicMoveStopSystem move;   // gravity (0, -98, 0), Systems.swift:1407
    for (int s = 0; s < 240; ++s) move.fixedUpdate(*world, 1.0f / 60.0f);
    world->synchronize();
    world->download(0, 1, &body, nullptr, &c, nullptr, nullptr, nullptr);
    CHECK(std::fabs(body.position[1] - (-3 + 2.5 + 0.05)) < 0.02);
    CHECK((c.flags & SGE_CTRL_GROUNDED) && (c.flags & SGE_CTRL_GROUNDED_NEAR) && c.groundTriangleIndex >= 0);
    CHECK(std::fabs(body.linearVelocity[1]) < 1e-6);
    // stages that need assets which were never uploaded report an error through the mirror's exception
    bool threw = false;
    try { sge::PoseStackSystem().fixedUpdate(*world, 1.0f / 60.0f); } catch (const sge::Error& e) { threw = std::strstr(e.what(), "skeleton") != nullptr; }
    CHECK(threw);
    std::printf("host mirror smoke ok: settled at y = %.4f\n", body.position[1]);
    return 0;
}
