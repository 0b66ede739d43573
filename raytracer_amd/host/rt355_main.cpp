// rt355_main.cpp -- the reference's main() (image_synthesis/main.cpp:522-642) on the GPU path:
// same camera, the four objects it pushes, the directional light it constructs (plus, with
// `2` as third argument, the point light at light_src2's position), Shader colours of
// main.cpp:630, alias 0; writes a BMP and a raw RGB8 dump.  With a fifth argument, the path of the OBJ file
// whose vertices are the gather directions (main.cpp:557 reads "geodesic.obj"), the frame is rendered the
// way HEAD of the reference does: CalcSubPixel2 with Shader::shading3 (main.cpp:205,331).
//   rt355_demo [width height [n_lights [out_prefix [geodesic.obj]]]]
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
    sce.objs = objs;
    std::vector<light_src *> light;
    light.push_back(new light_src(0, Point3D(-100, -30, 100), Vector3D(-1, 1, -1), Vector3D(1, 1, 1)));
    if (n_lights >= 2) light.push_back(new light_src(1, Point3D(30, -30, 50), Vector3D(0, 0, -1), Vector3D(1, 1, 1)));
    Shader shd(light, Color(0x000000u), Color(0xddddddu), Color(0x000000u));
    try {
        DeviceScene dev(sce.objs);
        const bool gather = argc > 5;
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
