// rt_cli.cpp -- headless C++17 host for librt_mi355.so (the reference has no CLI; its ImGui shell is
// replaced by flags whose defaults are RenderParams.h's).  Walks the reference's start-up and frame loop through the
// C ABI only: load .obj -> gather_model_triangles -> build_bvh -> upload (application.cpp:260-275), load the cube-map
// cross (:281-304), then N x { mainLoop steps + renderRay } and the present pass, and writes the back buffer as PNG.
//
//   rt_cli --obj models/bunny.obj --env cubemaps/Sky_16.png --size 1920x1080 --spp 4 --frames 32 --bvh --out frame
// Several --obj files are merged into one triangle soup before build_bvh (a multi-object scene, BASELINE config 5);
// --dump-targets also writes the four render targets of the last frame as little-endian PFM (float, bottom row first).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_mi355.h"

static void die(RtContext *c, const char *what, int rc) {
    std::fprintf(stderr, "rt_cli: %s failed (%d): %s\n", what, rc, rt_last_error(c));
    std::exit(1);
}

int main(int argc, char **argv) {
    std::vector<std::string> objs;
    std::string env, out = "frame";
    bool dumpTargets = false;
    int W = 1920, H = 1080, frames = 1, device = 0, useBVH = 0, showMotion = 0;
    RtRenderParams params;
    rt_default_render_params(&params);
    RtCamera cam;
    rt_default_camera(&cam);
    bool aspectSet = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "rt_cli: %s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--obj") { objs.push_back(next()); useBVH = 1; }
        else if (a == "--dump-targets") dumpTargets = true;
        else if (a == "--env") env = next();
        else if (a == "--out") out = next();
        else if (a == "--size") { if (std::sscanf(next(), "%dx%d", &W, &H) != 2) { std::fprintf(stderr, "rt_cli: --size WxH\n"); return 2; } }
        else if (a == "--spp") params.sppPerFrame = std::atoi(next());
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--bvh") useBVH = 1;
        else if (a == "--analytic") useBVH = 0;
        else if (a == "--motion") showMotion = 1;
        else if (a == "--no-gi") params.enableGI = 0;
        else if (a == "--no-ao") params.enableAO = 0;
        else if (a == "--no-taa") params.enableTAA = 0;
        else if (a == "--no-svgf") params.enableSVGF = 0;
        else if (a == "--no-env") params.enableEnvMap = 0;
        else if (a == "--exposure") params.exposure = (float)std::atof(next());
        else if (a == "--cam") { if (std::sscanf(next(), "%f,%f,%f,%f,%f", &cam.pos[0], &cam.pos[1], &cam.pos[2], &cam.yaw, &cam.pitch) != 5) { std::fprintf(stderr, "rt_cli: --cam x,y,z,yaw,pitch\n"); return 2; } }
        else if (a == "--fov") cam.fov = (float)std::atof(next());
        else if (a == "--aspect") { cam.aspect = (float)std::atof(next()); aspectSet = true; }
        else { std::fprintf(stderr, "usage: rt_cli [--obj f.obj] [--env cross.png] [--size WxH] [--spp n] [--frames n] [--bvh|--analytic] [--motion]\n"
                                    "              [--cam x,y,z,yaw,pitch] [--fov deg] [--aspect a] [--exposure e] [--no-gi --no-ao --no-taa --no-svgf --no-env] [--out prefix]\n"
                                    "              (--obj may be repeated; --dump-targets writes prefix_{color,motion,gpos,gnrm}.pfm)\n"); return a == "--help" ? 0 : 2; }
    }
    if (!aspectSet) cam.aspect = (float)W / (float)H;

    RtDeviceConfig cfg{};
    cfg.device = device; cfg.rank = 0; cfg.worldSize = 1; cfg.pipeline = RT_PIPELINE_AUTO;
    RtContext *ctx = nullptr;
    int rc = rt_create(&cfg, &ctx);
    if (rc != RT_OK) die(nullptr, "rt_create", rc);

    if (!objs.empty()) {
        float M[16];
        rt_default_bvh_transform(M);                                  // include/app/state.h:26-31
        std::vector<float> tris9;
        for (const std::string &obj : objs) {
            float *pos = nullptr; uint32_t *idx = nullptr; int nv = 0, ni = 0;
            if ((rc = rt_load_obj(obj.c_str(), &pos, &nv, &idx, &ni)) != RT_OK) die(ctx, "rt_load_obj", rc);
            const size_t at = tris9.size();
            tris9.resize(at + (size_t)(ni / 3) * 9);
            const int nt = rt_gather_triangles(pos, idx, ni, M, tris9.data() + at);
            tris9.resize(at + (size_t)nt * 9);
            std::printf("[OBJ] %s: %d vertices, %d triangles\n", obj.c_str(), nv, nt);
            rt_free(pos); rt_free(idx);
        }
        const int nt = (int)(tris9.size() / 9);
        std::vector<float> nodes12((size_t)nt * 24 + 12), tris12((size_t)nt * 12 + 12);
        const int nn = rt_build_bvh(tris9.data(), nt, nodes12.data(), tris12.data());
        if ((rc = rt_upload_bvh(ctx, nodes12.data(), nn, tris12.data(), nt)) != RT_OK) die(ctx, "rt_upload_bvh", rc);
        std::printf("[BVH] %d triangles, %d nodes\n", nt, nn);
    }
    if (!env.empty()) {
        uint8_t *px = nullptr; int w = 0, h = 0, ch = 0;
        if ((rc = rt_load_png(env.c_str(), &px, &w, &h, &ch)) != RT_OK) die(ctx, "rt_load_png", rc);
        std::vector<uint8_t> faces((size_t)6 * (h / 3) * (h / 3) * ch + 16);
        const int n = rt_cubemap_from_cross(px, w, h, ch, faces.data());
        rt_free(px);
        if (n == 0) { std::fprintf(stderr, "[ENV] %s is not a 4x3 cross, keeping the dummy cube map\n", env.c_str()); }   // application.cpp:294-304
        else if ((rc = rt_upload_env(ctx, faces.data(), n, ch)) != RT_OK) die(ctx, "rt_upload_env", rc);
        else std::printf("[ENV] %s: 6 x %dx%d\n", env.c_str(), n, n);
    }
    if ((rc = rt_resize(ctx, W, H)) != RT_OK) die(ctx, "rt_resize", rc);

    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; ++f)
        if ((rc = rt_render_ray(ctx, &params, &cam, useBVH, showMotion, nullptr, nullptr)) != RT_OK) die(ctx, "rt_render_ray", rc);
    rt_synchronize(ctx);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("[ACCUM] %d frame(s) %dx%d spp=%d in %.2f ms (%.2f ms/frame)\n", frames, W, H, params.sppPerFrame, ms, ms / (frames > 0 ? frames : 1));

    RtPresentParams pp;
    rt_make_present_params(&params, showMotion, W, H, &pp);
    std::vector<uint8_t> rgba((size_t)W * H * 4);
    if ((rc = rt_present(ctx, &pp, rgba.data())) != RT_OK) die(ctx, "rt_present", rc);
    const std::string png = out + ".png";
    if ((rc = rt_save_png(png.c_str(), rgba.data(), W, H, 4, /*flipY=*/1)) != RT_OK) die(ctx, "rt_save_png", rc);
    std::printf("[PRESENT] wrote %s\n", png.c_str());
    if (dumpTargets) {
        static const char *names[4] = {"color", "motion", "gpos", "gnrm"};
        static const int chans[4] = {4, 2, 4, 4};
        for (int t = 0; t < 4; ++t) {
            std::vector<float> img((size_t)W * H * chans[t]);
            if ((rc = rt_read_target(ctx, t, img.data(), RT_FORMAT_F32)) != RT_OK) die(ctx, "rt_read_target", rc);
            // PFM holds 1 or 3 channels: RGB of COLOR0 / GPOS / GNRM, and motion as (x, y, 0)
            std::vector<float> rgb((size_t)W * H * 3, 0.0f);
            for (size_t i = 0; i < (size_t)W * H; ++i)
                for (int c = 0; c < 3 && c < chans[t]; ++c) rgb[i * 3 + c] = img[i * chans[t] + c];
            const std::string f = out + "_" + names[t] + ".pfm";
            FILE *fp = std::fopen(f.c_str(), "wb");
            if (!fp) { std::fprintf(stderr, "rt_cli: cannot write %s\n", f.c_str()); return 1; }
            std::fprintf(fp, "PF\n%d %d\n-1.0\n", W, H);            // negative scale = little endian; rows bottom to top = our row order
            std::fwrite(rgb.data(), sizeof(float), rgb.size(), fp);
            std::fclose(fp);
            std::printf("[DUMP] wrote %s\n", f.c_str());
        }
    }
    rt_destroy(ctx);
    return 0;
}
