// Host-side entry points of the C ABI (BVH builder, PNG / OBJ readers, camera, cube-map slicing) exercised under
// AddressSanitizer + UBSan on the CPU build (tests/test_host_sanitizers.py).  GPU sanitizers are not available on the pool.
#include "rt_mi355.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cstring>
#include <string>
int main(int argc, char **argv) {
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    std::mt19937 r(1);
    std::uniform_real_distribution<float> U(-1, 1);
    for (int n : {0, 1, 7, 8, 9, 100, 5000}) {
        std::vector<float> t9((size_t)n * 9 + 9);
        for (auto &v : t9) v = U(r);
        std::vector<float> nodes((size_t)n * 24 + 12), tris((size_t)n * 12 + 12);
        int nn = rt_build_bvh(t9.data(), n, nodes.data(), tris.data());
        std::printf("n=%d nodes=%d\n", n, nn);
    }
    // degenerate: all identical triangles
    { int n = 300; std::vector<float> t9((size_t)n * 9, 0.5f), nodes((size_t)n * 24 + 12), tris((size_t)n * 12 + 12); std::printf("degenerate nodes=%d\n", rt_build_bvh(t9.data(), n, nodes.data(), tris.data())); }
    // png round trip
    { int W = 37, H = 21; std::vector<uint8_t> px((size_t)W * H * 4); for (auto &v : px) v = (uint8_t)r();
      rt_save_png((tmp + "/a.png").c_str(), px.data(), W, H, 4, 1);
      uint8_t *q = nullptr; int w, h, c; int rc = rt_load_png((tmp + "/a.png").c_str(), &q, &w, &h, &c); std::printf("png rc=%d %dx%dx%d\n", rc, w, h, c); rt_free(q);
      rc = rt_load_png((tmp + "/a.obj").c_str(), &q, &w, &h, &c); std::printf("not-a-png rc=%d\n", rc); }
    // obj
    { FILE *f = std::fopen((tmp + "/a.obj").c_str(), "w"); std::fprintf(f, "# c\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvn 0 0 1\nf 1 2 3 4\nf -1/1/1 -2//1 -3\nf 1 2\nf 9 9 9\n"); std::fclose(f);
      float *p = nullptr; uint32_t *ix = nullptr; int nv, ni; int rc = rt_load_obj((tmp + "/a.obj").c_str(), &p, &nv, &ix, &ni); std::printf("obj rc=%d nv=%d ni=%d\n", rc, nv, ni); if (rc == 0) { rt_free(p); rt_free(ix); } }
    // camera / uniforms
    { RtRenderParams p; rt_default_render_params(&p); RtCamera c; rt_default_camera(&c); float V[16], P[16], VP[16]; rt_camera_view(&c, V); rt_camera_proj(&c, P); rt_mat4_mul(P, V, VP);
      RtUniforms u; rt_make_uniforms(&p, &c, V, VP, VP, 640, 480, 3, 0, 1, 0, 10, 10, 1, &u); float j[2]; for (int i = 0; i < 40; ++i) rt_generate_jitter(i, j); std::printf("uniform fov %f jitter %f %f\n", u.tanHalfFov, j[0], j[1]); }
    // cubemap cross
    { int n = 5; std::vector<uint8_t> img((size_t)4 * n * 3 * n * 3), faces((size_t)6 * n * n * 3); for (auto &v : img) v = (uint8_t)r(); std::printf("cross faces=%d\n", rt_cubemap_from_cross(img.data(), 4 * n, 3 * n, 3, faces.data()));
      std::printf("bad cross=%d\n", rt_cubemap_from_cross(img.data(), 4 * n + 1, 3 * n, 3, faces.data())); }
    return 0;
}
