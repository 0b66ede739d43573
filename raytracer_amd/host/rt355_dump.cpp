// rt355_dump.cpp -- what the host-side loaders of rt355_host.hpp read from a file, as raw numbers, so that
// they can be compared with the reference's own loaders (anyobject.h:546-644, texture.h:42-79,
// main.cpp:408-450).  No GPU work; builds with any C++ compiler.
//   rt355_dump obj  <file.obj> <ox> <oy> <oz> <scale> <out.f64>   triangles, 15 doubles each (p0 p1 p2 uv0 uv1 uv2)
//   rt355_dump bmp  <file.bmp> <out.u8>                            int32 width, int32 height, then the texels
//   rt355_dump geo  <file.obj> <out.f64>                           directions, 3 doubles each
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rt355_host.hpp"

using namespace rt355;

int main(int argc, char **argv) {
    try {
        if (argc == 8 && !std::strcmp(argv[1], "obj")) {
            Mesh m(argv[2], Point3D(std::atof(argv[3]), std::atof(argv[4]), std::atof(argv[5])), std::atof(argv[6]));
            FILE *f = std::fopen(argv[7], "wb");
            if (!f) return 2;
            for (const Triangle &t : m.triangles) {
                const double row[15] = {t.p0.x, t.p0.y, t.p0.z, t.p1.x, t.p1.y, t.p1.z, t.p2.x, t.p2.y, t.p2.z,
                                        t.UV_p0.x, t.UV_p0.y, t.UV_p1.x, t.UV_p1.y, t.UV_p2.x, t.UV_p2.y};
                std::fwrite(row, sizeof(double), 15, f);
            }
            std::fclose(f);
            return 0;
        }
        if (argc == 4 && !std::strcmp(argv[1], "bmp")) {
            Texture t(argv[2]);
            FILE *f = std::fopen(argv[3], "wb");
            if (!f) return 2;
            const int32_t wh[2] = {t.width, t.height};
            std::fwrite(wh, sizeof(int32_t), 2, f);
            std::fwrite(t.rgb.data(), 1, t.rgb.size(), f);
            std::fclose(f);
            return 0;
        }
        if (argc == 4 && !std::strcmp(argv[1], "geo")) {
            std::vector<Vector3D> g = Geodesic(argv[2]);
            FILE *f = std::fopen(argv[3], "wb");
            if (!f) return 2;
            for (const Vector3D &v : g) { const double row[3] = {v.x, v.y, v.z}; std::fwrite(row, sizeof(double), 3, f); }
            std::fclose(f);
            return 0;
        }
    } catch (const Error &e) {
        std::fprintf(stderr, "rt355_dump: %s\n", e.what());
        return 1;
    }
    std::fprintf(stderr, "usage: rt355_dump obj|bmp|geo ... (see the source)\n");
    return 64;
}
