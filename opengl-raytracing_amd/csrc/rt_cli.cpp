// rt_cli.cpp -- headless C++17 host for librt_mi355.so (the reference has no CLI; its ImGui shell is
// replaced by flags whose defaults are RenderParams.h's).  Walks the reference's start-up and frame loop through the
// C ABI only: load .obj -> gather_model_triangles -> build_bvh -> upload (application.cpp:260-275), load the cube-map
// cross (:281-304), then N x { mainLoop steps + renderRay } and the present pass, and writes the back buffer as PNG.
//
//   rt_cli --obj models/bunny.obj --env cubemaps/Sky_16.png --size 1920x1080 --spp 4 --frames 32 --bvh --out frame
// Several --obj files are merged into one triangle soup before build_bvh (a multi-object scene, BASELINE config 5);
// --dump-targets also writes the four render targets of the last frame as little-endian PFM (float, bottom row first).
//
// Tile-parallel: --ranks N forks N processes BEFORE anything touches a GPU (rank r drives --devices[r], default device r); each
// renders the 16x16 tiles with tile % N == r, rank 0 creates the RCCL id (rt_comm_unique_id) and hands it over through a file,
// all call rt_comm_init, and every --gather-every k-th frame (and the last one) ends with rt_gather_frame -- for a static camera
// the accumulation history is tile-local, so frames in between need no exchange at all.  Rank 0 presents from
// the gathered targets and writes the same files a single-GPU run writes.
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cerrno>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt_mi355.h"

// ---- scene files --------------------------------------------------------------------------------------------------------
// --scene file.json: one flat JSON object.  Keys are RenderParams' field names (include/render/RenderParams.h: the reference
// has no scene file, its defaults are the schema) plus: "obj" (string or array of strings), "env", "size" [W, H], "frames",
// "bvh" (bool), "motion" (bool), "out", "camera" {"pos": [x, y, z], "yaw", "pitch", "fov"}.  Later command-line flags win.
namespace scenefile {
struct Value { enum Kind { Num, Str, Bool, Arr, Obj } kind = Num; double num = 0; std::string str; std::vector<Value> arr; std::vector<std::pair<std::string, Value>> obj; };
struct Parser {
    const std::string &s; size_t i = 0; std::string err;
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i; }
    bool fail(const char *m) { if (err.empty()) err = std::string(m) + " at byte " + std::to_string(i); return false; }
    bool str(std::string &out) {
        if (i >= s.size() || s[i] != '"') return fail("expected string");
        for (++i; i < s.size() && s[i] != '"'; ++i) { if (s[i] == '\\' && i + 1 < s.size()) ++i; out.push_back(s[i]); }
        if (i >= s.size()) return fail("unterminated string");
        ++i; return true;
    }
    bool value(Value &v) {
        ws();
        if (i >= s.size()) return fail("unexpected end");
        if (s[i] == '"') { v.kind = Value::Str; return str(v.str); }
        if (s[i] == '[') {
            v.kind = Value::Arr; ++i; ws();
            if (i < s.size() && s[i] == ']') { ++i; return true; }
            for (;;) { Value e; if (!value(e)) return false; v.arr.push_back(e); ws(); if (i < s.size() && s[i] == ',') { ++i; continue; } if (i < s.size() && s[i] == ']') { ++i; return true; } return fail("expected , or ]"); }
        }
        if (s[i] == '{') {
            v.kind = Value::Obj; ++i; ws();
            if (i < s.size() && s[i] == '}') { ++i; return true; }
            for (;;) { ws(); std::string k; if (!str(k)) return false; ws(); if (i >= s.size() || s[i] != ':') return fail("expected :"); ++i; Value e; if (!value(e)) return false; v.obj.emplace_back(k, e);
                       ws(); if (i < s.size() && s[i] == ',') { ++i; continue; } if (i < s.size() && s[i] == '}') { ++i; return true; } return fail("expected , or }"); }
        }
        if (!s.compare(i, 4, "true")) { v.kind = Value::Bool; v.num = 1; i += 4; return true; }
        if (!s.compare(i, 5, "false")) { v.kind = Value::Bool; v.num = 0; i += 5; return true; }
        char *end = nullptr; v.kind = Value::Num; v.num = std::strtod(s.c_str() + i, &end);
        if (end == s.c_str() + i) return fail("expected a value");
        i = (size_t)(end - s.c_str()); return true;
    }
};
struct Field { const char *name; int kind; size_t off; int n; };   // kind 0 = int32 (numbers and booleans), 1 = float[n]
#define RT_PI(f) {#f, 0, offsetof(RtRenderParams, f), 1}
#define RT_PF(f) {#f, 1, offsetof(RtRenderParams, f), 1}
#define RT_P3(f) {#f, 1, offsetof(RtRenderParams, f), 3}
static const Field kFields[] = {
    RT_PI(sppPerFrame), RT_PF(exposure), RT_P3(matAlbedoColor), RT_PF(matAlbedoSpecStrength), RT_PF(matAlbedoGloss), RT_PI(matGlassEnabled),
    RT_P3(matGlassColor), RT_PF(matGlassIOR), RT_PF(matGlassDistortion), RT_PI(matMirrorEnabled), RT_P3(matMirrorColor), RT_PF(matMirrorGloss),
    RT_PI(enableJitter), RT_PF(jitterStillScale), RT_PF(jitterMovingScale), RT_PI(enableGI), RT_PF(giScaleAnalytic), RT_PF(giScaleBVH),
    RT_PI(enableEnvMap), RT_PF(envMapIntensity), RT_PI(sunEnabled), RT_P3(sunColor), RT_PF(sunIntensity), RT_PF(sunYaw), RT_PF(sunPitch),
    RT_PI(skyEnabled), RT_P3(skyColor), RT_PF(skyIntensity), RT_PF(skyYaw), RT_PF(skyPitch), RT_PI(pointLightEnabled), RT_P3(pointLightColor),
    RT_PF(pointLightIntensity), RT_P3(pointLightPos), RT_PI(pointLightOrbitEnabled), RT_PF(pointLightOrbitRadius), RT_PF(pointLightOrbitSpeed),
    RT_PF(pointLightYaw), RT_PF(pointLightPitch), RT_PI(enableAO), RT_PI(aoSamples), RT_PF(aoRadius), RT_PF(aoBias), RT_PF(aoMin), RT_PI(enableTAA),
    RT_PF(taaStillThresh), RT_PF(taaHardMovingThresh), RT_PF(taaHistoryMinWeight), RT_PF(taaHistoryAvgWeight), RT_PF(taaHistoryMaxWeight),
    RT_PF(taaHistoryBoxSize), RT_PI(enableSVGF), RT_PF(svgfVarMax), RT_PF(svgfKVar), RT_PF(svgfKColor), RT_PF(svgfKVarMotion),
    RT_PF(svgfKColorMotion), RT_PF(svgfStrength), RT_PF(motionScale)};
static bool set_param(RtRenderParams &p, const std::string &key, const Value &v) {
    for (const Field &f : kFields) {
        if (key != f.name) continue;
        char *base = reinterpret_cast<char *>(&p) + f.off;
        if (f.kind == 0) { if (v.kind != Value::Num && v.kind != Value::Bool) return false; *reinterpret_cast<int32_t *>(base) = (int32_t)v.num; return true; }
        if (f.n == 1) { if (v.kind != Value::Num) return false; *reinterpret_cast<float *>(base) = (float)v.num; return true; }
        if (v.kind != Value::Arr || (int)v.arr.size() != f.n) return false;
        for (int k = 0; k < f.n; ++k) { if (v.arr[(size_t)k].kind != Value::Num) return false; reinterpret_cast<float *>(base)[k] = (float)v.arr[(size_t)k].num; }
        return true;
    }
    return false;
}
}  // namespace scenefile

static float half_to_float(uint16_t h) {   // exact widening of an IEEE binary16
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = s;
        else { int sh = 0; uint32_t mm = m; while (!(mm & 1024u)) { mm <<= 1; ++sh; } bits = s | (uint32_t)(113 - sh) << 23 | (mm & 1023u) << 13; }
    } else if (e == 31) bits = s | 0x7f800000u | m << 13;
    else bits = s | (e + 112u) << 23 | m << 13;
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

static void die(RtContext *c, const char *what, int rc) {
    std::fprintf(stderr, "rt_cli: %s failed (%d): %s\n", what, rc, rt_last_error(c));
    std::exit(1);
}

int main(int argc, char **argv) {
    std::vector<std::string> objs;
    std::string env, out = "frame";
    bool dumpTargets = false;
    float dt = 1.0f / 60.0f;   // seconds per frame for the point-light orbit
    int W = 1920, H = 1080, frames = 1, device = 0, useBVH = 0, showMotion = 0;
    int giBounces = 1;
    int envFilter = 0;   // RtExtension.envFilter: 0 = bilinear weights in exact fp32, 1 = texel coordinates rounded to 1/256 texel
    int ranks = 0, gatherEvery = 1;          // ranks 0 = plain single-process run without a communicator
    bool dryRun = false;                     // --dry-run: fork + id hand-over only, nothing touches a GPU (rehearsal of the launcher on any host)
    std::vector<int> devices;
    RtRenderParams params;
    rt_default_render_params(&params);
    RtCamera cam;
    rt_default_camera(&cam);
    bool aspectSet = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "rt_cli: %s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--scene") {
            std::ifstream in(next());
            if (!in) { std::fprintf(stderr, "rt_cli: cannot read scene file %s\n", argv[i]); return 2; }
            std::stringstream ss; ss << in.rdbuf();
            const std::string text = ss.str();
            scenefile::Parser ps{text};
            scenefile::Value root;
            if (!ps.value(root) || root.kind != scenefile::Value::Obj) { std::fprintf(stderr, "rt_cli: %s: %s\n", argv[i], ps.err.empty() ? "not a JSON object" : ps.err.c_str()); return 2; }
            for (const auto &kv : root.obj) {
                const std::string &k = kv.first; const scenefile::Value &v = kv.second;
                bool ok = true;
                if (k == "obj") { if (useBVH == 0) useBVH = 1; if (v.kind == scenefile::Value::Str) objs.push_back(v.str); else if (v.kind == scenefile::Value::Arr) for (const auto &e : v.arr) { ok = ok && e.kind == scenefile::Value::Str; objs.push_back(e.str); } else ok = false; }
                else if (k == "env") { ok = v.kind == scenefile::Value::Str; env = v.str; }
                else if (k == "out") { ok = v.kind == scenefile::Value::Str; out = v.str; }
                else if (k == "size") { ok = v.kind == scenefile::Value::Arr && v.arr.size() == 2; if (ok) { W = (int)v.arr[0].num; H = (int)v.arr[1].num; } }
                else if (k == "frames") frames = (int)v.num;
                else if (k == "dt") dt = (float)v.num;
                else if (k == "bvh") useBVH = v.num != 0;
                else if (k == "hybrid") { if (v.num != 0) useBVH = RT_SCENE_HYBRID; }
                else if (k == "giBounces") giBounces = (int)v.num;
                else if (k == "envFilter") envFilter = (int)v.num;
                else if (k == "motion") showMotion = v.num != 0;
                else if (k == "camera") {
                    ok = v.kind == scenefile::Value::Obj;
                    for (const auto &ck : v.obj) {
                        if (ck.first == "pos" && ck.second.kind == scenefile::Value::Arr && ck.second.arr.size() == 3) for (int q = 0; q < 3; ++q) cam.pos[q] = (float)ck.second.arr[(size_t)q].num;
                        else if (ck.first == "yaw") cam.yaw = (float)ck.second.num;
                        else if (ck.first == "pitch") cam.pitch = (float)ck.second.num;
                        else if (ck.first == "fov") cam.fov = (float)ck.second.num;
                        else if (ck.first == "aspect") { cam.aspect = (float)ck.second.num; aspectSet = true; }
                        else ok = false;
                    }
                } else ok = scenefile::set_param(params, k, v);
                if (!ok) { std::fprintf(stderr, "rt_cli: %s: bad or unknown key \"%s\"\n", argv[i], k.c_str()); return 2; }
            }
        }
        else if (a == "--obj") { objs.push_back(next()); if (useBVH == 0) useBVH = 1; }
        else if (a == "--dump-targets") dumpTargets = true;
        else if (a == "--dt") dt = (float)std::atof(next());
        else if (a == "--env") env = next();
        else if (a == "--out") out = next();
        else if (a == "--size") { if (std::sscanf(next(), "%dx%d", &W, &H) != 2) { std::fprintf(stderr, "rt_cli: --size WxH\n"); return 2; } }
        else if (a == "--spp") params.sppPerFrame = std::atoi(next());
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--ranks") ranks = std::atoi(next());
        else if (a == "--gather-every") gatherEvery = std::max(1, std::atoi(next()));
        else if (a == "--dry-run") dryRun = true;
        else if (a == "--devices") { std::stringstream ss(next()); std::string t; while (std::getline(ss, t, ',')) devices.push_back(std::atoi(t.c_str())); }
        else if (a == "--bvh") useBVH = 1;
        else if (a == "--analytic") useBVH = 0;
        else if (a == "--hybrid") useBVH = RT_SCENE_HYBRID;          // EXTENSION: analytic scene + the mesh (include/rt_mi355.h)
        else if (a == "--gi-bounces") giBounces = std::atoi(next());  // EXTENSION: bounces of the analytic / hybrid GI path
        else if (a == "--env-filter") envFilter = std::atoi(next());  // cube-map filter model (RtExtension.envFilter)
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
                                    "              [--hybrid [--gi-bounces n]]   EXTENSION: the analytic scene with the .obj mesh added to it, n diffuse GI bounces\n"
                                    "              [--env-filter 0|1]   cube-map filter model: exact fp32 weights (default) / texel coordinates rounded to 1/256 texel\n"
                                    "              [--ranks N [--devices d0,d1,..] [--gather-every k] [--dry-run]]   tile-parallel over N GPUs, one process each, RCCL gather to rank 0\n"
                                    "              (--obj may be repeated; --dump-targets writes prefix_{color,motion,gpos,gnrm}.pfm; --scene file.json sets any of\n"
                                    "               the above and every RenderParams field by name)\n"); return a == "--help" ? 0 : 2; }
    }
    if (!aspectSet) cam.aspect = (float)W / (float)H;

    // ---- tile-parallel: one fresh process per GPU, forked before any GPU call (nothing above this line touches HIP)
    int rank = 0, world = 1;
    // the id file carries the launcher's pid: concurrent runs with the same --out never see each other's id
    const std::string idFile = out + ".rccl_id." + std::to_string((long)getpid());
    if (ranks > 0) {
        world = ranks;
        if (!devices.empty() && (int)devices.size() != ranks) { std::fprintf(stderr, "rt_cli: --devices needs %d entries\n", ranks); return 2; }
        std::remove(idFile.c_str());
        std::fflush(stdout); std::fflush(stderr);          // nothing buffered may be inherited by the children
        std::vector<pid_t> kids;
        bool child = false, forkFailed = false;
        for (int r = 0; r < ranks; ++r) {
            pid_t pid = fork();
            if (pid < 0) { std::perror("rt_cli: fork"); forkFailed = true; break; }
            if (pid == 0) { rank = r; child = true; break; }
            kids.push_back(pid);
        }
        if (!child) {
            // Wait for whichever rank ends first: a rank that dies before or inside rt_comm_init (bad --devices entry, rt_create or
            // .obj failure on one rank) would leave the others in ncclCommInitRank / the grouped send-recv forever, so the first
            // failure -- or a failed fork -- ends the remaining ranks.
            int bad = forkFailed ? 1 : 0;
            size_t left = kids.size();
            auto stop_rest = [&]() { for (pid_t k : kids) if (k > 0) kill(k, SIGTERM); };
            if (forkFailed) stop_rest();
            while (left > 0) {
                int st = 0;
                const pid_t done = waitpid(-1, &st, 0);
                if (done < 0) { if (errno == EINTR) continue; break; }
                for (pid_t &k : kids) if (k == done) { k = -1; --left; }
                if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { if (!bad) stop_rest(); ++bad; }
            }
            std::remove(idFile.c_str());
            if (bad) std::fprintf(stderr, "rt_cli: %d of %d ranks failed or were stopped\n", bad, ranks);
            return bad ? 1 : 0;
        }
        device = devices.empty() ? rank : devices[(size_t)rank];
    }
    const bool root = rank == 0;

    RtDeviceConfig cfg{};
    cfg.device = device; cfg.rank = rank; cfg.worldSize = world; cfg.pipeline = RT_PIPELINE_AUTO;
    RtContext *ctx = nullptr;
    int rc = RT_OK;
    if (dryRun && ranks <= 0) { std::printf("[DRY] single process, device %d, no GPU call made\n", device); return 0; }
    if (!dryRun) {
        rc = rt_create(&cfg, &ctx);
        if (rc != RT_OK) die(nullptr, "rt_create", rc);
    }
    if (ranks > 0) {
        unsigned char id[RT_COMM_ID_BYTES];
        if (root) {
            if (dryRun) { for (size_t q = 0; q < sizeof id; ++q) id[q] = (unsigned char)(q * 7u + 3u); }   // stand-in for ncclGetUniqueId
            else if ((rc = rt_comm_unique_id(id, sizeof id)) != RT_OK) die(nullptr, "rt_comm_unique_id", rc);
            const std::string tmp = idFile + ".tmp";
            FILE *fp = std::fopen(tmp.c_str(), "wb");
            if (!fp || std::fwrite(id, 1, sizeof id, fp) != sizeof id) { std::fprintf(stderr, "rt_cli: cannot write %s\n", tmp.c_str()); return 1; }
            std::fclose(fp);
            std::rename(tmp.c_str(), idFile.c_str());          // atomic: readers see all 128 bytes or no file
        } else {
            bool got = false;
            for (int tries = 0; tries < 1200 && !got; ++tries) {   // up to 2 minutes
                FILE *fp = std::fopen(idFile.c_str(), "rb");
                if (fp) { got = std::fread(id, 1, sizeof id, fp) == sizeof id; std::fclose(fp); }
                if (!got) usleep(100 * 1000);
            }
            if (!got) { std::fprintf(stderr, "rt_cli: rank %d never saw %s\n", rank, idFile.c_str()); return 1; }
        }
        if (dryRun) {
            bool ok = true;
            for (size_t q = 0; q < sizeof id; ++q) ok = ok && id[q] == (unsigned char)(q * 7u + 3u);
            std::printf("[DRY] rank %d of %d device %d id %s\n", rank, world, device, ok ? "ok" : "CORRUPT");
            std::fflush(stdout);
            if (const char *e = std::getenv("RT_CLI_DRY_FAIL_RANK")) if (std::atoi(e) == rank) return 3;   // test hook: one rank dies early
            if (const char *e = std::getenv("RT_CLI_DRY_HANG_RANK")) if (std::atoi(e) == rank) for (;;) pause();   // test hook: a rank stuck in a collective
            return ok ? 0 : 1;
        }
        if ((rc = rt_comm_init(ctx, id, sizeof id)) != RT_OK) die(ctx, "rt_comm_init", rc);
        if (root) std::printf("[RCCL] %d ranks, communicator up\n", world);
    }
#define RT_SAY(...) do { if (root) std::printf(__VA_ARGS__); } while (0)

    if (!objs.empty()) {
        float M[16];
        rt_default_bvh_transform(M);                                  // include/app/state.h:26-31
        std::vector<float> tris9;
        for (const std::string &obj : objs) {
            float *pos = nullptr; uint32_t *idx = nullptr; int nv = 0, ni = 0;
            if ((rc = rt_load_obj(obj.c_str(), &pos, &nv, &idx, &ni)) != RT_OK) die(ctx, "rt_load_obj", rc);
            const size_t at = tris9.size();
            tris9.resize(at + (size_t)(ni / 3) * 9);
            const int nt = rt_gather_triangles_checked(pos, nv, idx, ni, M, tris9.data() + at);
            if (nt < 0) die(ctx, "rt_gather_triangles_checked", nt);
            tris9.resize(at + (size_t)nt * 9);
            RT_SAY("[OBJ] %s: %d vertices, %d triangles\n", obj.c_str(), nv, nt);
            rt_free(pos); rt_free(idx);
        }
        const int nt = (int)(tris9.size() / 9);
        std::vector<float> nodes12((size_t)nt * 24 + 12), tris12((size_t)nt * 12 + 12);
        const int nn = rt_build_bvh(tris9.data(), nt, nodes12.data(), tris12.data());
        if ((rc = rt_upload_bvh(ctx, nodes12.data(), nn, tris12.data(), nt)) != RT_OK) die(ctx, "rt_upload_bvh", rc);
        RT_SAY("[BVH] %d triangles, %d nodes\n", nt, nn);
    }
    if (!env.empty()) {
        uint8_t *px = nullptr; int w = 0, h = 0, ch = 0;
        if ((rc = rt_load_png(env.c_str(), &px, &w, &h, &ch)) != RT_OK) die(ctx, "rt_load_png", rc);
        std::vector<uint8_t> faces((size_t)6 * (h / 3) * (h / 3) * ch + 16);
        const int n = rt_cubemap_from_cross(px, w, h, ch, faces.data());
        rt_free(px);
        if (n == 0) { std::fprintf(stderr, "[ENV] %s is not a 4x3 cross, keeping the dummy cube map\n", env.c_str()); }   // application.cpp:294-304
        else if ((rc = rt_upload_env(ctx, faces.data(), n, ch)) != RT_OK) die(ctx, "rt_upload_env", rc);
        else RT_SAY("[ENV] %s: 6 x %dx%d\n", env.c_str(), n, n);
    }
    if ((rc = rt_resize(ctx, W, H)) != RT_OK) die(ctx, "rt_resize", rc);
    if (giBounces != 1 || envFilter != 0) { RtExtension ext{}; ext.giBounces = giBounces; ext.envFilter = envFilter; if ((rc = rt_set_extension(ctx, &ext)) != RT_OK) die(ctx, "rt_set_extension", rc); }

    const auto t0 = std::chrono::steady_clock::now();
    const bool lightMoving = params.pointLightOrbitEnabled != 0 && std::fabs(params.pointLightOrbitSpeed) > 1e-5f && params.pointLightOrbitRadius > 0.0f;
    for (int f = 0; f < frames;) {
        // point-light orbit animation, application.cpp:341-348 (deg/s * s), with a fixed time step instead of glfwGetTime()
        if (params.pointLightOrbitEnabled) {
            params.pointLightYaw += params.pointLightOrbitSpeed * dt;
            if (params.pointLightYaw > 360.0f) params.pointLightYaw -= 360.0f;
            if (params.pointLightYaw < -360.0f) params.pointLightYaw += 360.0f;
        }
        // a still scene accumulates: hand the library runs of frames (up to the next gather), it renders them in batches.  The yaw
        // above advances once per loop turn, so frames are only grouped when that changes nothing: orbit off, or a speed of exactly 0.
        int n = 1;
        if (!lightMoving && !(params.pointLightOrbitEnabled != 0 && params.pointLightOrbitSpeed != 0.0f)) {
            n = frames - f;
            if (ranks > 0) n = std::min(n, gatherEvery - f % gatherEvery);
        }
        if ((rc = rt_render_ray_frames(ctx, &params, &cam, useBVH, showMotion, n)) != RT_OK) die(ctx, "rt_render_ray_frames", rc);
        f += n;
        // tile-parallel: COLOR0 to rank 0 every gatherEvery-th frame (what an interactive viewer would show) -- frames in between
        // need no exchange, the history a static camera reads is tile-local
        if (ranks > 0 && (f % gatherEvery == 0 || f == frames) && (rc = rt_gather_frame(ctx, RT_TARGET_COLOR)) != RT_OK) die(ctx, "rt_gather_frame", rc);
        // an orbiting light is dynamic geometry for the accumulation: the history is invalid after every frame (application.cpp:538-553)
        if (lightMoving && f < frames && (rc = rt_reset_accum(ctx)) != RT_OK) die(ctx, "rt_reset_accum", rc);
    }
    rt_synchronize(ctx);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    RT_SAY("[ACCUM] %d frame(s) %dx%d spp=%d in %.2f ms (%.2f ms/frame)%s\n", frames, W, H, params.sppPerFrame, ms, ms / (frames > 0 ? frames : 1),
           ranks > 0 ? ", tile-parallel incl. the gathers" : "");

    RtPresentParams pp;
    rt_make_present_params(&params, showMotion, W, H, &pp);
    std::vector<uint8_t> rgba((size_t)W * H * 4);
    if (ranks > 0) {
        // the 7x7 present filter crosses tiles: all four targets of the last frame go to rank 0, which filters from the gathered blocks
        for (int t = 1; t < 4; ++t) if ((rc = rt_gather_frame(ctx, t)) != RT_OK) die(ctx, "rt_gather_frame", rc);
        if (!root) { rt_synchronize(ctx); rt_destroy(ctx); return 0; }
        if ((rc = rt_present_last_gathered(ctx, &pp, rgba.data())) != RT_OK) die(ctx, "rt_present_last_gathered", rc);
    } else if ((rc = rt_present(ctx, &pp, rgba.data())) != RT_OK) die(ctx, "rt_present", rc);
    const std::string png = out + ".png";
    if ((rc = rt_save_png(png.c_str(), rgba.data(), W, H, 4, /*flipY=*/1)) != RT_OK) die(ctx, "rt_save_png", rc);
    std::printf("[PRESENT] wrote %s\n", png.c_str());
    if (dumpTargets) {
        static const char *names[4] = {"color", "motion", "gpos", "gnrm"};
        static const int chans[4] = {4, 2, 4, 4};
        for (int t = 0; t < 4; ++t) {
            std::vector<float> img((size_t)W * H * chans[t]);
            if (ranks > 0) {
                std::vector<uint16_t> halfs(img.size());
                if ((rc = rt_read_gathered(ctx, t, halfs.data())) != RT_OK) die(ctx, "rt_read_gathered", rc);
                for (size_t i = 0; i < img.size(); ++i) img[i] = half_to_float(halfs[i]);
            } else if ((rc = rt_read_target(ctx, t, img.data(), RT_FORMAT_F32)) != RT_OK) die(ctx, "rt_read_target", rc);
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
