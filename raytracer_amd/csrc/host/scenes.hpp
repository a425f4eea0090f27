// Scene library interface: the builders of src/main.rs:244-765 and the camera /
// background / sample-count constants of src/main.rs:55-179.
#pragma once
#include "../../../include/rt/host.h"
#include "scene.hpp"

namespace rth {

struct SceneOptions {
    bool bvh_shutter_times = false;  // random_scene: BVH over (0,1) instead of the reference's (0,0.1) (src/main.rs:315)
    bool bvh_widest_axis = false;    // BVHNode split axis: widest spread of box minima instead of random
    uint32_t boxes_per_side = 20;    // src/main.rs:591
    uint64_t n_triangles = 0;        // synthetic triangle count of RT_SCENE_KALEIDOSCOPE_MESH
    const uint8_t* image_rgb = nullptr;
    uint32_t image_w = 0, image_h = 0;
};

bool scene_setup(int scene, RtSceneSetup& out);
HitPtr build_scene(int scene, const SceneOptions& options);

}  // namespace rth
