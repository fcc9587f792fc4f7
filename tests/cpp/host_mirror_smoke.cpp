// Exercises the C++ host mirror (swift-game-engine_amd/host/sge_host.hpp) against the C ABI on a real GPU:
// CollisionQuery (cast / ground cast / overlap / overlapAll / raycast / updateDynamicTransforms) and a character dropped on the
// demo's ground quad through KinematicMoveStopSystem (config 1 of SURVEY.md §8d: settles at y = -3 + 2.5 + 0.05).
//   g++ -std=c++17 tests/cpp/host_mirror_smoke.cpp -Iinclude -Lswift-game-engine_amd -lsge_amd -Wl,-rpath,$PWD/swift-game-engine_amd -o /tmp/host_mirror_smoke
#include <cmath>
#include <cstdio>
#include <cstring>
#include <unordered_set>
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
    // ---- the World <-> GPU bridge: a host that keeps its component stores (World.swift:64-75) ----------------------------------
    // 256 entities with non-contiguous ids walk over the ground quad; a "steering system" rewrites the MoveIntent store every
    // step; GPUCharacterStepSystem pushes the intents through the pinned staging, ticks, pulls the step back into the stores.
    // The stores after every step must equal a synchronous download, in both modes (World at step n / World one step behind).
    {
        const uint32_t stages = SGE_STAGE_INTENT | SGE_STAGE_GRAVITY | SGE_STAGE_MOVE | SGE_STAGE_LOCOMOTION | SGE_STAGE_ACTION | SGE_STAGE_WRITEBACK;
        for (int lagged = 0; lagged < 2; ++lagged) {
            sge::EntityWorld ew;
            std::unordered_map<sge::Entity, double> startX;
            const int n = 256;
            for (int i = 0; i < n; ++i) {
                const sge::Entity e = 1000u + 7u * (uint32_t)((i * 37) % n);   // ids out of order: the crowd sorts them
                sge::PhysicsBodyComponent b;
                b.position[0] = -30.0 + 4.0 * (i % 16); b.position[1] = -0.45; b.position[2] = -30.0 + 4.0 * (i / 16);
                ew.bodies[e] = b;
                startX[e] = b.position[0];
                ew.transforms[e] = sge::TransformComponent{};
                ew.controllers[e] = sge::CharacterControllerComponent{};
                ew.intents[e] = sge::MoveIntentComponent{};
                ew.movements[e] = sge::MovementComponent{};
            }
            sge::GPUCrowd crowd(*world);
            crowd.rebuild(ew);
            CHECK((int)crowd.entities().size() == n && crowd.entities()[0] < crowd.entities()[1] && crowd.index(crowd.entities()[5]).value() == 5);
            sge::GPUCharacterStepSystem step(*world, crowd, ew, stages, lagged != 0);
            std::vector<sge_body_state> prevBodies(n), nowBodies(n);
            std::vector<sge_controller_state> nowCtrl(n);
            for (int s = 0; s < 90; ++s) {
                for (auto& kv : ew.intents) {   // the steering system: everybody circles, phase by entity id
                    const float a = 0.05f * (float)s + 0.01f * (float)kv.first;
                    kv.second.desiredVelocity[0] = 4.5f * std::cos(a); kv.second.desiredVelocity[2] = 4.5f * std::sin(a);
                }
                std::unordered_set<sge::Entity> dirty;
                if (s == 40) {                 // a teleport written into the World by some other system
                    const sge::Entity e = crowd.entities()[17];
                    ew.bodies[e].position[1] += 3.0;
                    dirty.insert(e);
                }
                step.fixedUpdate(1.0f / 60.0f, dirty);
                prevBodies = nowBodies;
                world->download(0, n, nowBodies.data(), nullptr, nowCtrl.data(), nullptr, nullptr, nullptr);   // synchronous reference
                const std::vector<sge_body_state>& expect = lagged ? prevBodies : nowBodies;
                if (lagged && s == 0) continue;
                for (int i = 0; i < n; ++i) {
                    const sge::PhysicsBodyComponent& b = ew.bodies.at(crowd.entities()[(size_t)i]);
                    // (the step after a teleport the lagged World still holds the teleported value the host wrote: skip that entity)
                    if (lagged && s == 40 && i == 17) continue;
                    CHECK(std::memcmp(b.position, expect[i].position, sizeof(b.position)) == 0);
                    CHECK(std::memcmp(b.linearVelocity, expect[i].linearVelocity, sizeof(b.linearVelocity)) == 0);
                }
                if (!lagged)
                    for (int i = 0; i < n; ++i) {
                        const sge::CharacterControllerComponent& c = ew.controllers.at(crowd.entities()[(size_t)i]);
                        CHECK(c.grounded == ((nowCtrl[i].flags & SGE_CTRL_GROUNDED) != 0) && c.groundTriangleIndex == nowCtrl[i].groundTriangleIndex);
                        CHECK((int)c.contactManifoldTriangles.size() == nowCtrl[i].manifoldCount);
                    }
            }
            step.finish();
            // everybody moved, the teleported one came down again
            double moved = 0;
            for (const auto& kv : ew.bodies) moved += std::fabs(kv.second.position[0] - startX.at(kv.first));
            CHECK(moved > 10.0);
            CHECK(std::fabs(ew.bodies.at(crowd.entities()[17]).position[1] - (-0.45)) < 0.05);
            CHECK(std::fabs(ew.transforms.at(crowd.entities()[3]).translation[0] - (float)ew.bodies.at(crowd.entities()[3]).position[0]) < 1e-6f);
        }
    }
    std::printf("host mirror smoke ok: settled at y = %.4f\n", body.position[1]);
    return 0;
}
