// TEST INFRASTRUCTURE ONLY — the reference's scene library (src/main.rs:244-765)
// and per-scene constants (src/main.rs:55-179) restated for the CPU oracle.
// random_double / random_int draw from the seeded scene stream (philox.hpp).
#pragma once
#include "oracle.hpp"

namespace orc {

struct SceneOptions {
    bool bvh_shutter_times = false;  // random_scene BVH over (0,1) instead of the reference's (0,0.1)
    bool bvh_widest_axis = false;    // BVHNode split axis: widest spread of box minima instead of random
    uint32_t boxes_per_side = 20;    // src/main.rs:591
    uint64_t n_triangles = 0;        // synthetic triangle soup of scene 12
    const uint8_t* image_rgb = nullptr;
    uint32_t image_w = 0, image_h = 0;
};

struct SceneSetup {
    Point lookfrom{13.0, 2.0, 3.0}, lookat{0.0, 0.0, 0.0};  // src/main.rs:55-63
    Vec3 vup{0.0, 1.0, 0.0};
    double vfov = 40.0, aspect_ratio = 3.0 / 2.0, aperture = 0.0, focus_dist = 10.0, time0 = 0.0, time1 = 1.0;
    Color background{0.0, 0.0, 0.0};
    uint32_t samples_per_pixel = 64, width = 800;
};

inline bool scene_setup(int scene, SceneSetup& s) {  // src/main.rs:66-162
    s = SceneSetup();
    switch (scene) {
        case 1: s.background = Color(0.7, 0.8, 1.0); s.vfov = 20.0; s.aperture = 0.1; return true;
        case 2: s.background = Color(0.7, 0.8, 1.0); s.vfov = 20.0; return true;
        case 3: s.samples_per_pixel = 400; s.lookfrom = Point(26.0, 3.0, 6.0); s.lookat = Point(0.0, 2.0, 0.0); s.vfov = 20.0; return true;
        case 4: s.lookfrom = Point(13.0, 5.0, 10.0); s.focus_dist = 15.0; s.aperture = 0.4; s.vfov = 40.0; s.samples_per_pixel = 300; return true;
        case 5: s.vfov = 20.0; s.background = Color(0.5, 0.8, 0.8); return true;
        case 6: s.vfov = 20.0; s.background = Color(0.5, 0.8, 0.8); return true;
        case 7: case 8:
            s.aspect_ratio = 1.0; s.samples_per_pixel = 200; s.lookfrom = Point(278.0, 278.0, -800.0);
            s.lookat = Point(278.0, 278.0, 0.0); s.vfov = 40.0; return true;
        case 9:
            s.aspect_ratio = 1.0; s.samples_per_pixel = 1000; s.lookfrom = Point(478.0, 278.0, -600.0);
            s.lookat = Point(278.0, 278.0, 0.0); s.vfov = 40.0; return true;
        case 10:
            s.aspect_ratio = 1.0; s.samples_per_pixel = 256; s.background = Color(0.52, 0.80, 0.92);
            s.lookfrom = Point(3.0, 2.0, -7.0); s.vfov = 40.0; return true;
        case 11: case 12:
            s.aspect_ratio = 1.0; s.samples_per_pixel = 256; s.background = Color(0.52, 0.80, 0.92);
            s.lookfrom = Point(1.5, -2.4, -0.866); s.vup = Vec3(0.0, 0.0, -1.0); s.lookat = Point(1.5, 0.0, -0.866);
            s.vfov = 70.0; return true;
    }
    return false;
}

template <class T, class... A>
std::shared_ptr<T> mk(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }

inline MatPtr checker_ground() {  // src/main.rs:246-250
    return mk<Lambertian>(TexPtr(mk<CheckerTexture>(Color(0.2, 0.3, 0.1), Color(0.9, 0.9, 0.9))));
}

inline HitPtr random_scene(const SceneOptions& o) {  // src/main.rs:244-316
    HittableList world;
    world.add(mk<Sphere>(Point(0.0, -1000.0, 0.0), 1000.0, checker_ground()));
    for (int a = -11; a < 11; ++a)
        for (int b = -11; b < 11; ++b) {
            double choose_mat = random_double(0.0, 1.0);
            double cx = (double)a + 0.9 * random_double(0.0, 1.0);
            double cz = (double)b + 0.9 * random_double(0.0, 1.0);
            Point center(cx, 0.2, cz);
            if ((center - Point(4.0, 0.2, 0.0)).length() > 0.9) {
                if (choose_mat < 0.8) {
                    Color c1 = Color::random(0.0, 1.0);
                    Color c2 = Color::random(0.0, 1.0);
                    MatPtr m = mk<Lambertian>(Vec3::elemul(c1, c2));
                    Point center2 = center + Vec3(0.0, random_double(0.0, 0.5), 0.0);
                    world.add(mk<MovingSphere>(center, center2, 0.0, 1.0, 0.2, m));
                } else if (choose_mat < 0.95) {
                    Color albedo = Color::random(0.5, 1.0);
                    double fuzz = random_double(0.0, 0.5);
                    world.add(mk<Sphere>(center, 0.2, MatPtr(mk<Metal>(albedo, fuzz))));
                } else {
                    world.add(mk<Sphere>(center, 0.2, MatPtr(mk<Dielectric>(1.5))));
                }
            }
        }
    world.add(mk<Sphere>(Point(0.0, 1.0, 0.0), 1.0, MatPtr(mk<Dielectric>(1.5))));
    world.add(mk<Sphere>(Point(-4.0, 1.0, 0.0), 1.0, MatPtr(mk<Lambertian>(Color(0.4, 0.2, 0.1)))));
    world.add(mk<Sphere>(Point(4.0, 1.0, 0.0), 1.0, MatPtr(mk<Metal>(Color(0.7, 0.6, 0.5), 0.0))));
    return BVHNode_new(world, 0.0, o.bvh_shutter_times ? 1.0 : 0.1);
}

inline HitPtr simple_light() {  // src/main.rs:318-344
    HittableList world;
    MatPtr ground = checker_ground();
    world.add(mk<Sphere>(Point(0.0, -1000.0, 0.0), 1000.0, ground));
    world.add(mk<Sphere>(Point(0.0, 2.0, 0.0), 2.0, ground));
    MatPtr difflight = mk<DiffuseLight>(TexPtr(mk<SolidColor>(Color(4.0, 4.0, 4.0))));
    world.add(XYRect(3.0, 5.0, 1.0, 3.0, -2.0, difflight));
    return BVHNode_new(world, 0.0, 1.0);
}

inline HitPtr light_demo() {  // src/main.rs:346-422
    HittableList world;
    world.add(mk<Sphere>(Point(0.0, -1000.0, 0.0), 1000.0, checker_ground()));
    for (int a = -15; a < 15; ++a)
        for (int b = -15; b < 15; ++b) {
            double choose_mat = random_double(0.0, 1.0);
            double r = random_double(0.1, 0.4);
            double cx = (double)a / 1.5 + 0.25 * random_double(0.0, 1.0);
            double cz = (double)b / 1.5 + 0.25 * random_double(0.0, 1.0);
            Point center(cx, r, cz);
            if ((center - Point(0.0, 0.0, 0.0)).length() > 2.0 && (center - Point(2.0, 0.0, 0.0)).length() > 1.3 &&
                (center - Point(-2.0, 0.0, 0.0)).length() > 1.3) {
                if (choose_mat < 0.2) {
                    Color c1 = Color::random(0.0, 1.0);
                    Color c2 = Color::random(0.0, 1.0);
                    world.add(mk<Sphere>(center, r, MatPtr(mk<Lambertian>(Vec3::elemul(c1, c2)))));
                } else if (choose_mat < 0.7) {
                    MatPtr l = mk<DiffuseLight>(TexPtr(mk<SolidColor>(Color::random(0.0, 1.0))));
                    world.add(mk<Sphere>(center, r, l));
                } else if (choose_mat < 0.85) {
                    Color albedo = Color::random(0.5, 1.0);
                    double fuzz = random_double(0.0, 0.5);
                    world.add(mk<Sphere>(center, r, MatPtr(mk<Metal>(albedo, fuzz))));
                } else {
                    world.add(mk<Sphere>(center, r, MatPtr(mk<Dielectric>(1.5))));
                }
            }
        }
    MatPtr difflight = mk<DiffuseLight>(TexPtr(mk<CheckerTexture>(Color(1.0, 0.5, 0.44), Color(1.0, 0.64, 0.0))));
    world.add(mk<Sphere>(Point(0.0, 0.9, 0.0), 0.9, difflight));
    MatPtr material1 = mk<Dielectric>(1.5);
    MatPtr material3 = mk<Metal>(Color(0.7, 0.6, 0.5), 0.0);
    world.add(mk<Sphere>(Point(-2.0, 0.65, 0.0), 0.65, material3));
    world.add(mk<Sphere>(Point(2.0, 0.65, 0.0), 0.65, material1));
    return BVHNode_new(world, 0.0, 0.1);
}

inline HitPtr two_perlin_spheres() {  // src/main.rs:424-442
    HittableList world;
    TexPtr pertext = mk<NoiseTexture>(4.0);
    world.add(mk<Sphere>(Point(0.0, -1000.0, 0.0), 1000.0, MatPtr(mk<Lambertian>(pertext))));
    world.add(mk<Sphere>(Point(0.0, 2.0, 0.0), 2.0, MatPtr(mk<Lambertian>(pertext))));
    return BVHNode_new(world, 0.0, 0.1);
}

inline TexPtr earth_texture(const SceneOptions& o) {
    static const uint8_t grey[12] = {128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128};
    if (o.image_rgb) return mk<ImageTexture>(o.image_w, o.image_h, o.image_rgb);
    return mk<ImageTexture>(2u, 2u, grey);
}

inline HitPtr earth(const SceneOptions& o) {  // src/main.rs:444-457
    MatPtr earth_surface = mk<Lambertian>(earth_texture(o));
    HittableList world;
    world.add(mk<Sphere>(Point(0.0, 0.0, 0.0), 2.0, earth_surface));
    return BVHNode_new(world, 0.0, 1.0);
}

inline void cornell_walls(HittableList& world, double lx0, double lx1, double lz0, double lz1, double emit, MatPtr& white) {
    MatPtr red = mk<Lambertian>(Color(0.65, 0.05, 0.05));  // src/main.rs:461-466 / 521-526
    white = mk<Lambertian>(Color(0.73, 0.73, 0.73));
    MatPtr green = mk<Lambertian>(Color(0.12, 0.45, 0.15));
    MatPtr light = mk<DiffuseLight>(TexPtr(mk<SolidColor>(Color(emit, emit, emit))));
    world.add(YZRect(0.0, 555.0, 0.0, 555.0, 555.0, green));  // :468-496 / :528-556
    world.add(YZRect(0.0, 555.0, 0.0, 555.0, 0.0, red));
    world.add(XZRect(lx0, lx1, lz0, lz1, 554.0, light));
    world.add(XZRect(0.0, 555.0, 0.0, 555.0, 0.0, white));
    world.add(XZRect(0.0, 555.0, 0.0, 555.0, 555.0, white));
    world.add(XYRect(0.0, 555.0, 0.0, 555.0, 555.0, white));
}

inline HitPtr cornell_box() {  // src/main.rs:459-517
    HittableList world;
    MatPtr white;
    cornell_walls(world, 213.0, 343.0, 227.0, 332.0, 15.0, white);
    HitPtr box1 = mk<Box6>(Point(0.0, 0.0, 0.0), Point(165.0, 330.0, 165.0), white);
    box1 = mk<RotateY>(box1, 15.0);
    box1 = mk<Translate>(box1, Vec3(265.0, 0.0, 295.0));
    world.add(box1);
    HitPtr box2 = mk<Box6>(Point(0.0, 0.0, 0.0), Point(165.0, 165.0, 165.0), white);
    box2 = mk<RotateY>(box2, -18.0);
    box2 = mk<Translate>(box2, Vec3(130.0, 0.0, 65.0));
    world.add(box2);
    return BVHNode_new(world, 0.0, 1.0);
}

inline HitPtr cornell_smoke() {  // src/main.rs:519-585
    HittableList world;
    MatPtr white;
    cornell_walls(world, 113.0, 443.0, 127.0, 432.0, 7.0, white);
    HitPtr box1 = mk<Box6>(Point(0.0, 0.0, 0.0), Point(165.0, 330.0, 165.0), white);
    box1 = mk<RotateY>(box1, 15.0);
    box1 = mk<Translate>(box1, Vec3(265.0, 0.0, 295.0));
    world.add(mk<ConstantMedium>(box1, 0.01, Color(0.0, 0.0, 0.0)));
    HitPtr box2 = mk<Box6>(Point(0.0, 0.0, 0.0), Point(165.0, 165.0, 165.0), white);
    box2 = mk<RotateY>(box2, -18.0);
    box2 = mk<Translate>(box2, Vec3(130.0, 0.0, 65.0));
    world.add(mk<ConstantMedium>(box2, 0.01, Color(1.0, 1.0, 1.0)));
    return BVHNode_new(world, 0.0, 1.0);
}

inline HitPtr final_scene(const SceneOptions& o) {  // src/main.rs:587-699
    HittableList boxes1;
    MatPtr ground = mk<Lambertian>(Color(0.48, 0.83, 0.53));
    int boxes_per_side = (int)o.boxes_per_side;
    for (int i = 0; i < boxes_per_side; ++i)
        for (int j = 0; j < boxes_per_side; ++j) {
            double w = 100.0;
            double x0 = -1000.0 + (double)i * w;
            double z0 = -1000.0 + (double)j * w;
            double y0 = 0.0;
            double x1 = x0 + w;
            double y1 = random_double(1.0, 101.0);
            double z1 = z0 + w;
            boxes1.add(mk<Box6>(Point(x0, y0, z0), Point(x1, y1, z1), ground));
        }
    HittableList world;
    world.add(BVHNode_new(boxes1, 0.0, 1.0));
    MatPtr light = mk<DiffuseLight>(Color(7.0, 7.0, 7.0));
    world.add(XZRect(123.0, 423.0, 147.0, 412.0, 554.0, light));
    Point center1(400.0, 400.0, 200.0);
    Point center2 = center1 + Vec3(30.0, 0.0, 0.0);
    world.add(mk<MovingSphere>(center1, center2, 0.0, 1.0, 50.0, MatPtr(mk<Lambertian>(Color(0.7, 0.3, 0.1)))));
    world.add(mk<Sphere>(Point(260.0, 150.0, 45.0), 50.0, MatPtr(mk<Dielectric>(1.5))));
    world.add(mk<Sphere>(Point(0.0, 150.0, 145.0), 50.0, MatPtr(mk<Metal>(Color(0.8, 0.8, 0.9), 10.0))));
    HitPtr boundary = mk<Sphere>(Point(360.0, 150.0, 145.0), 70.0, MatPtr(mk<Dielectric>(1.5)));
    world.add(boundary);
    world.add(mk<ConstantMedium>(boundary, 0.02, Color(0.2, 0.4, 0.9)));
    HitPtr boundary1 = mk<Sphere>(Point(0.0, 0.0, 0.0), 5000.0, MatPtr(mk<Dielectric>(1.5)));
    world.add(mk<ConstantMedium>(boundary1, 0.0001, Color(1.0, 1.0, 1.0)));
    MatPtr emat = mk<Lambertian>(earth_texture(o));
    world.add(mk<Sphere>(Point(400.0, 200.0, 400.0), 100.0, emat));
    TexPtr pertext = mk<NoiseTexture>(0.1);
    world.add(mk<Sphere>(Point(220.0, 280.0, 300.0), 80.0, MatPtr(mk<Lambertian>(pertext))));
    HittableList boxes2;
    MatPtr white = mk<Lambertian>(Color(0.73, 0.73, 0.73));
    int ns = 100;
    for (int j = 0; j < ns; ++j) boxes2.add(mk<Sphere>(Point::random(0.0, 165.0), 10.0, white));
    HitPtr inner = BVHNode_new(boxes2, 0.0, 1.0);
    world.add(mk<Translate>(HitPtr(mk<RotateY>(inner, 15.0)), Vec3(-100.0, 270.0, 395.0)));
    return BVHNode_new(world, 0.0, 1.0);
}

inline HitPtr try_triangle() {  // src/main.rs:701-738
    HittableList world;
    Point point1(2.0, 0.0, 0.0), point2(-2.0, 0.0, 0.0), point3(1.0, 2.0, 0.0), point4(-2.0, 4.0, -1.0);
    MatPtr material = mk<Metal>(Color(0.7, 0.6, 0.5), 0.3);
    world.add(mk<Triangle>(point1, point2, point3, material));
    world.add(mk<Triangle>(point4, point2, point3, material));
    Color c1 = Color::random(0.0, 1.0);
    Color c2 = Color::random(0.0, 1.0);
    world.add(mk<Sphere>(Point(0.0, 0.5, -3.0), 0.5, MatPtr(mk<Lambertian>(Vec3::elemul(c1, c2)))));
    world.add(mk<Sphere>(Point(0.0, -1000.0, 0.0), 1000.0, checker_ground()));
    return BVHNode_new(world, 0.0, 1.0);
}

inline void kaleidoscope_objects(HittableList& world) {  // src/main.rs:741-763
    MatPtr metal_material = mk<Metal>(Color(0.8, 0.8, 0.8), 0.01);
    HitPtr edge = mk<Box6>(Point(0.0, 0.0, 0.0), Point(3.0, 20.0, 0.1), metal_material);
    world.add(edge);
    world.add(mk<Translate>(HitPtr(mk<RotateY>(edge, 60.0)), Vec3(0.0, 0.0, 0.0)));
    world.add(mk<Translate>(HitPtr(mk<RotateY>(edge, 120.0)), Vec3(2.98, 0.0, 0.0)));
    world.add(mk<Triangle>(Point(2.5, 22.0, -0.1), Point(1.5, 20.0, -0.86), Point(1.9, 20.0, -1.5),
                           MatPtr(mk<Lambertian>(Color(1.0, 0.45, 0.45)))));
}

inline HitPtr kaleidoscope() {  // src/main.rs:740-765
    HittableList world;
    kaleidoscope_objects(world);
    return BVHNode_new(world, 0.0, 1.0);
}

// Scene 12 (no reference counterpart): the kaleidoscope plus a soup of jittered
// Lambertian triangles in four layers behind the far end of the mirror tube.
//   L = 4 layers at y = 20.5 + 0.5*l; G x G cells over x in [-8.5, 11.5],
//   z in [-10.866, 9.134], G = ceil(sqrt(n / (2L))); per cell four corners
//   (a,b) in (0,0),(1,0),(0,1),(1,1), each jittered by 0.2 cell in x, y and z
//   (draw order x, y, z), two triangles (p10,p01,p00) and (p01,p10,p11),
//   each with a palette material picked by random_int(0, 63).
inline HitPtr kaleidoscope_mesh(const SceneOptions& o) {
    HittableList world;
    kaleidoscope_objects(world);
    std::vector<MatPtr> palette;
    for (int i = 0; i < 64; ++i) palette.push_back(mk<Lambertian>(Color::random(0.1, 0.95)));
    const int L = 4;
    uint64_t G = (uint64_t)std::ceil(std::sqrt((double)o.n_triangles / (2.0 * L)));
    if (G < 1) G = 1;
    const double X0 = -8.5, Z0 = -10.866, W = 20.0;
    const double h = W / (double)G;
    for (int l = 0; l < L; ++l) {
        double yl = 20.5 + 0.5 * (double)l;
        for (uint64_t i = 0; i < G; ++i)
            for (uint64_t j = 0; j < G; ++j) {
                Point c[2][2];
                for (int b = 0; b < 2; ++b)
                    for (int a = 0; a < 2; ++a) {
                        double px = X0 + ((double)i + (double)a) * h + 0.2 * h * random_double(-1.0, 1.0);
                        double py = yl + 0.2 * h * random_double(-1.0, 1.0);
                        double pz = Z0 + ((double)j + (double)b) * h + 0.2 * h * random_double(-1.0, 1.0);
                        c[a][b] = Point(px, py, pz);
                    }
                MatPtr m1 = palette[(size_t)random_int(0, 63)];
                world.add(mk<Triangle>(c[1][0], c[0][1], c[0][0], m1));
                MatPtr m2 = palette[(size_t)random_int(0, 63)];
                world.add(mk<Triangle>(c[0][1], c[1][0], c[1][1], m2));
            }
    }
    return BVHNode_new(world, 0.0, 1.0);
}

inline HitPtr build_scene(int scene, const SceneOptions& o) {
    switch (scene) {
        case 1: case 2: return random_scene(o);
        case 3: return simple_light();
        case 4: return light_demo();
        case 5: return two_perlin_spheres();
        case 6: return earth(o);
        case 7: return cornell_box();
        case 8: return cornell_smoke();
        case 9: return final_scene(o);
        case 10: return try_triangle();
        case 11: return kaleidoscope();
        case 12: return kaleidoscope_mesh(o);
    }
    return nullptr;
}

}  // namespace orc
