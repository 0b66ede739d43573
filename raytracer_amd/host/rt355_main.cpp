// rt355_main.cpp -- the reference's main() (image_synthesis/main.cpp:522-642) on the GPU path:
// same camera, the four objects it pushes, the directional light it constructs (plus, with
// `2` as third argument, the point light at light_src2's position), Shader colours of
// main.cpp:630, alias 0; writes a BMP and a raw RGB8 dump.  With a fifth argument, the path of the OBJ file
// whose vertices are the gather directions (main.cpp:639 reads "geodesic.obj"), the frame is rendered the
// way HEAD of the reference does: CalcSubPixel2 with Shader::shading3 (main.cpp:205,331).
// With `all` as fifth argument and a BMP file as sixth, the list holds everything main() constructs (its six
// planes, three spheres -- sphere3 the mirror with the 3D-function texture -- and the textured Environment,
// main.cpp:547-597) plus what a user of the other classes would add: a glass sphere, a solid texture with a
// projector of its own, an Ellipsoid and a Cylinder.
//   rt355_demo [width height [n_lights [out_prefix [geodesic.obj | all texture.bmp]]]]
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "rt355_host.hpp"

using namespace rt355;

int main(int argc, char **argv) {
    int width = argc > 2 ? std::atoi(argv[1]) : 800, height = argc > 2 ? std::atoi(argv[2]) : 600;
    int n_lights = argc > 3 ? std::atoi(argv[3]) : 1;
    std::string prefix = argc > 4 ? argv[4] : "rt355_demo";
    int alias = 0;
    Scene sce;
    sce.p_eye = Point3D(0, -20, 0);
    sce.v_view = Vector3D(0, 1, 0);
    sce.v_up = Vector3D(0, 0, 1);
    sce.dist = 10;
    sce.s_x = sce.dist * 4.0 * width / height;
    sce.s_y = sce.dist * 4;
    std::vector<unsigned char> pixmap((size_t)width * height * 3);
    std::vector<AnyObject *> objs;
    objs.push_back(new Plane(Point3D(0, 13, 0), Vector3D(0, -1, 0), Vector3D(-1, 0, 0), Color(200, 200, 200), 0));
    objs.push_back(new Plane(Point3D(0, 0, -10), Vector3D(0, 0, 1), Vector3D(1, 0, 0), Color(230, 230, 230), 0));
    objs.push_back(new Sphere(Point3D(-10.5, 0, 0), 10, Color(100, 200, 100), Vector3D(1, 0, 0), Vector3D(0, 1, 0), Vector3D(0, 0, 1), 0));
    objs.push_back(new Sphere(Point3D(10.5, 0, 0), 10, Color(200, 100, 100), Vector3D(1, 0, 0), Vector3D(0, 1, 0), Vector3D(0, 0, 1), 0));
    const bool all = argc > 6 && std::string(argv[5]) == "all";
    if (all) {
        objs.clear();
        objs.push_back(new Plane(Point3D(0, 13, 0), Vector3D(0, -1, 0), Vector3D(-1, 0, 0), Color(200, 200, 200), 0));
        objs.push_back(new Plane(Point3D(0, -100, 0), Vector3D(0, 1, 0), Vector3D(-1, 0, 0), Color(214, 147, 44), 0));
        objs.push_back(new Plane(Point3D(0, 0, -10), Vector3D(0, 0, 1), Vector3D(1, 0, 0), Color(230, 230, 230), 0));
        objs.push_back(new Plane(Point3D(0, 0, 100), Vector3D(0, 0, -1), Vector3D(1, 0, 0), Color(255, 0, 0), 0));
        objs.push_back(new Plane(Point3D(100, 0, 0), Vector3D(-1, 0, 0), Vector3D(0, 1, 0), Color(0, 255, 0), 0));
        objs.push_back(new Plane(Point3D(-100, 0, 0), Vector3D(1, 0, 0), Vector3D(0, 1, 0), Color(0, 255, 0), 0));
        Sphere *sphere1 = new Sphere(Point3D(-10.5, 0, 0), 10, Color(100, 200, 100), Vector3D(1, 0, 0), Vector3D(0, 1, 0), Vector3D(0, 0, 1), 0);
        // a solid texture: the picture projected along (0.1, -1, 0.2) onto a 24 x 18 rectangle about (-8, 0, 2)
        for (int k = 0; k < 2; k++)
            sphere1->AddTexture(std::make_shared<Texture>(argv[6], 24.0, 18.0, Normalize(Vector3D(1, 0.2, 0)), Normalize(Vector3D(-0.1, 0.1, 1)),
                                                          Normalize(Vector3D(0.1, -1, 0.2)), Point3D(-8, 0, 2)));
        sphere1->texture_type = 2;
        objs.push_back(sphere1);
        Sphere *sphere2 = new Sphere(Point3D(10.5, 0, 0), 10, Color(200, 100, 100), Vector3D(1, 0, 0), Vector3D(0, 1, 0), Vector3D(0, 0, 1), 0.3);
        sphere2->fresnel = 0.3; sphere2->IOR = 1.5; // glass (main.cpp:116-154)
        objs.push_back(sphere2);
        Sphere *sphere3 = new Sphere(Point3D(-20, -20, 20), 10, Color(73, 179, 248), Vector3D(1, 0, 0), Vector3D(0, 1, 0), Vector3D(0, 0, 1), 1);
        sphere3->AddTexture(std::make_shared<Texture>(argv[6], 0.5, 0.5)); sphere3->AddTexture(std::make_shared<Texture>(argv[6], 0.5, 0.5));
        sphere3->texture_type = 3;
        objs.push_back(sphere3);
        Environment *environment = new Environment(Point3D(0, -50, 20), 1000, Color(255, 255, 0), Vector3D(-1, 1, 0), Vector3D(-1, -1, 0), Vector3D(0, 0, 1), 0);
        environment->AddTexture(std::make_shared<Texture>(argv[6], 1, 1)); environment->AddTexture(std::make_shared<Texture>(argv[6], 1, 1));
        environment->texture_type = 1;
        objs.push_back(environment);
        Ellipsoid *egg = new Ellipsoid();
        egg->SetPCenter(Point3D(0, -6, -5)); egg->SetN(Vector3D(1, 0, 0), Vector3D(0, 1, 0), Vector3D(0, 0, 1));
        egg->SetS(4, 2.5f, 3); egg->SetColor(Color(220, 200, 80)); egg->SetKs(0.2);
        objs.push_back(egg);
        Cylinder *post = new Cylinder();
        post->SetPCenter(Point3D(30, 5, 0)); post->SetN(Normalize(Vector3D(1, 0.1, 0)), Normalize(Vector3D(-0.1, 1, 0.2)), Normalize(Vector3D(0, -0.2, 1)));
        post->SetS(2.2f, 2.2f, 1); post->SetColor(Color(90, 200, 90));
        objs.push_back(post);
    }
    sce.objs = objs;
    std::vector<light_src *> light;
    light.push_back(new light_src(0, Point3D(-100, -30, 100), Vector3D(-1, 1, -1), Vector3D(1, 1, 1)));
    if (n_lights >= 2) light.push_back(new light_src(1, Point3D(30, -30, 50), Vector3D(0, 0, -1), Vector3D(1, 1, 1)));
    Shader shd(light, Color(0x000000u), Color(0xddddddu), Color(0x000000u));
    try {
        DeviceScene dev(sce.objs);
        const bool gather = argc > 5 && !all;
        if (gather) dev.set_directions(Geodesic(argv[5]));
        auto t0 = std::chrono::steady_clock::now();
        rt355_stats st = setPixels(dev, sce, shd, alias, width, height, pixmap.data(), 0, 0,
                                   gather ? RT355_SHADING_BLEED : RT355_SHADING_LIGHTS);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        SaveImage((prefix + ".bmp").c_str(), width, height, pixmap.data());
        FILE *f = std::fopen((prefix + ".rgb").c_str(), "wb");
        if (f) { std::fwrite(pixmap.data(), 1, pixmap.size(), f); std::fclose(f); }
        std::printf("%dx%d: %llu primary + %llu shadow + %llu gather rays, %.3f ms (device %.3f ms) -> %s.bmp\n", width, height,
                    (unsigned long long)st.primary_rays, (unsigned long long)st.shadow_rays,
                    (unsigned long long)st.gather_rays, ms, st.ms_device, prefix.c_str());
    } catch (const Error &e) {
        std::fprintf(stderr, "rt355: %s\n", e.what());
        return 1;
    }
    for (auto *o : objs) delete o;
    for (auto *l : light) delete l;
    return 0;
}
