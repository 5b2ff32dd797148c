// main.cpp — headless counterpart of the reference's window loop (reference src/main.cpp:6-138):
// same objects built in the same order, `frames` iterations of trace (+ filter), then PNGs.
//   rt_headless scene.obj [--size WxH] [--frames N] [--bounces B] [--filter ITERS] [--exposure E]
//               [--auto-camera] [--light] [--out PREFIX] [--res DIR] [--kernel trace3|trace|trace1|trace2|trace3ltc]
//               [--rect-light cx,cy,cz,xx,xy,xz,yx,yy,yz]... [--light-intensity I] [--roughness R]   (trace3ltc)
//               [--keys "FRAME:KEY,FRAME:KEY,..."] [--reset-at FRAME] [--freeze-time] [--dump-raw PREFIX] [--gpus N]
//               [--watertight] [--fast-shadows] [--weighted-lights] [--blue-noise]   (opt-in variants, include/rt_launcher.h)
// --kernel selects one of the reference's other kernels (its analytic box scenes; the obj file is still loaded,
// as the reference's main does, but not traced); trace1 / trace2 read their tables from --res (default "res").
// --kernel trace3ltc is not a program of the reference: trace2.cs's LTC shading at the primary hit of the obj scene, lit by
// the --rect-light rectangles (corner, edge x, edge y; RT_KERNEL_TRACE3_LTC in include/rt_launcher.h).
// --keys feeds the reference's input handler (glfw_setter.hpp) before the given frame of the loop: W A S D Z X move the
// camera and restart the accumulation, COMMA / PERIOD change maxBounces (restart), SPACE toggles the filter, MINUS /
// EQUAL change its iterations, 1 / 2 the exposure, SCROLL+n / SCROLL-n the field of view, MOUSE<x>x<y> moves the cursor
// (mouse-look: the first position is the reference point, later ones turn the camera); --reset-at K stands for any
// input that only resets frameNumber at frame K.  --gpus N renders on N GPUs of this node, one host thread and one
// launcher context per GPU: the frame is split into screen-space tiles (tiling.hpp), every GPU traces its tile plus the
// filter halo and filters it locally, rt_gather_tiles assembles the frame on GPU 0 over RCCL.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "glfw_setter.hpp"
#include "png_write.hpp"
#include "render.hpp"
#include "rt_loader.hpp"
#include "scene.hpp"
#include "tiling.hpp"

namespace {

struct Options {
    std::string obj;
    int width = 512, height = 512, frames = 4, bounces = 5, filterIters = 0, gpus = 0;
    float exposure = 1.0f, time0 = 1.25f, timeStep = 0.0625f;
    bool autoCamera = false, light = false, freezeTime = false;
    std::string out = "frame", res = "res", kernel = "trace3", dumpRaw;
    int launchFlags = 0, maxFollows = 1000, resetAt = -1;
    std::vector<std::pair<int, std::string>> keys; // (frame, key name)
    std::vector<rt_rect_light> rectLights;          // --rect-light (kernel trace3ltc)
    float lightIntensity = 6.0f, roughness = 0.5f;  // LIGHT_INTENSITY of trace2.cs:20; the `roughness` uniform
};

struct FrameState { // what the loop of reference main.cpp:67-134 sees at the top of one iteration
    rtm::vec3 eye, lookAt;
    float fov, exposure;
    int frameNumber, maxBounces, iterations;
    bool filterOn;
};

int keyCode(const std::string &k) {
    static const std::pair<const char *, int> names[] = {
        {"W", GLFW_KEY_W}, {"A", GLFW_KEY_A}, {"S", GLFW_KEY_S}, {"D", GLFW_KEY_D}, {"Z", GLFW_KEY_Z}, {"X", GLFW_KEY_X},
        {"COMMA", GLFW_KEY_COMMA}, {"PERIOD", GLFW_KEY_PERIOD}, {"SPACE", GLFW_KEY_SPACE}, {"MINUS", GLFW_KEY_MINUS},
        {"EQUAL", GLFW_KEY_EQUAL}, {"1", GLFW_KEY_1}, {"2", GLFW_KEY_2}};
    for (const auto &n : names)
        if (k == n.first) return n.second;
    return -1;
}

// Dry run of the input script: the per-frame state every GPU replays (the handler's state is process-wide, like the
// reference's static members, so it is evaluated once, here).
bool scriptStates(const Options &o, Camera camera, std::vector<FrameState> &states) {
    GlfwSetter::reset();
    GlfwSetter::camera = &camera;
    int maxBounces = o.bounces, iterations = std::max(o.filterIters, 1);
    GlfwSetter::maxBounces = &maxBounces;
    GlfwSetter::iterations = &iterations;
    GlfwSetter::filter = o.filterIters > 0;
    GlfwSetter::exposure = o.exposure;
    for (int f = 0; f < o.frames; ++f) {
        for (const auto &k : o.keys) {
            if (k.first != f) continue;
            if (k.second.rfind("SCROLL", 0) == 0) {
                GlfwSetter::scroll_callback(0.0, std::atof(k.second.c_str() + 6));
            } else if (k.second.rfind("MOUSE", 0) == 0) { // MOUSE<x>x<y>: a cursor position (the first one only sets the reference point)
                double mx = 0, my = 0;
                if (std::sscanf(k.second.c_str() + 5, "%lfx%lf", &mx, &my) != 2) {
                    std::cerr << "bad mouse event " << k.second << " (MOUSE<x>x<y>)" << std::endl;
                    return false;
                }
                GlfwSetter::mouse_callback(mx, my);
            } else {
                const int code = keyCode(k.second);
                if (code < 0) {
                    std::cerr << "unknown key " << k.second << std::endl;
                    return false;
                }
                GlfwSetter::key_callback(code, 0, GLFW_PRESS, 0);
            }
        }
        if (f == o.resetAt) GlfwSetter::frameNumber = 0;
        states.push_back(FrameState{camera.eye, camera.lookAt, camera.fov, GlfwSetter::exposure, GlfwSetter::frameNumber,
                                    maxBounces, iterations, GlfwSetter::filter});
        GlfwSetter::frameNumber++; // main.cpp:129
    }
    GlfwSetter::reset();
    return true;
}

bool dumpFile(const std::string &path, const void *p, size_t n) {
    std::ofstream os(path, std::ios::binary);
    os.write(static_cast<const char *>(p), (std::streamsize)n);
    return (bool)os;
}

struct Shared { // what the GPU threads read (built once) and where they report
    explicit Shared(const Camera &c) : camera0(c) {}
    const Options *o;
    ModelLoader *loader;
    QTable *qt;
    BVHContainer *bvh;
    Camera camera0;
    std::vector<FrameState> states;
    unsigned char commId[RT_COMM_ID_BYTES];
    std::vector<int> status;
    std::vector<std::string> errors;
    // One halo for the whole run: the largest any scripted frame filters with.  The tile grid depends on the halo
    // (tiling.hpp: tileGrid) and SPACE / EQUAL / MINUS do not restart the accumulation (reference glfw_setter.cpp:
    // 207-218), so a halo that grew with the filter settings would move the tiles — and blend freshly traced halo
    // pixels onto colour that was never accumulated there — in the middle of a running average.
    int halo = 0;
    // a GPU thread that fails tells the others (checked once per frame) and main (which then exits non-zero even if
    // a peer is stuck inside ncclCommInitRank / a collective waiting for the failed rank)
    std::atomic<bool> abort{false};
    std::atomic<int> finished{0};
    std::mutex m;
    std::condition_variable cv;
};

struct RankAborted : std::runtime_error {
    RankAborted() : std::runtime_error("stopped: another GPU thread failed") {}
};

// The body of the reference's main() from the FrameBuffer on (main.cpp:32-134), on one GPU; with nranks > 1 for one
// tile of the frame.
void renderOn(Shared &sh, int rank, int nranks) {
    const Options &o = *sh.o;
    try {
        if (nranks > 0) rtMakeContextCurrent(rank); // one context per GPU, current on this thread
        const RtApi &api = rtApi();
        rt_ctx *ctx = rtCurrentContext();
        const int W = o.width, H = o.height;
        const bool tiled = nranks > 0;
        const int world = tiled ? nranks : 1;
        if (tiled) rtCheck(api.comm_init(ctx, world, rank, sh.commId, RT_COMM_ID_BYTES), "rt_comm_init");

        FrameBuffer framebuffer(W, H);
        Filter filter(W, H, framebuffer.texture, framebuffer.normalTex, framebuffer.posTex);
        Camera camera = sh.camera0;
        const std::string main_cs = "src/shaders/" + o.kernel + ".cs";
        Tracer tracer = o.kernel == "trace3"
                            ? Tracer(&camera, &framebuffer, sh.bvh, sh.qt)
                            : Tracer(&camera, &framebuffer, sh.bvh, sh.qt,
                                     std::vector<const char *>{main_cs.c_str(), "src/shaders/random.glsl"});
        tracer.resourceDir = o.res;
        if (o.kernel == "trace1" || (o.launchFlags & RT_FLAG_BLUE_NOISE)) tracer.genBlueNoiseBuffers();
        if (o.kernel == "trace2" || o.kernel == "trace3ltc") tracer.genLtcData();
        if (o.kernel == "trace3ltc") tracer.setRectLights(o.rectLights, o.lightIntensity);
        tracer.roughness = o.roughness;
        tracer.launchFlags = o.launchFlags;
        tracer.maxFollows = o.maxFollows;
        rtCheck(api.profile(ctx, 1), "rt_profile");

        bool filtered = false;
        float exposure = o.exposure;
        for (int f = 0; f < o.frames; ++f) { // the loop of main.cpp:67-134
            const FrameState &s = sh.states[f];
            camera.eye = s.eye;
            camera.lookAt = s.lookAt;
            camera.fov = s.fov;
            tracer.maxBounces = s.maxBounces;
            filter.iterations = s.iterations;
            exposure = s.exposure;
            filtered = s.filterOn;
            if (sh.abort.load()) throw RankAborted();
            const int halo = sh.halo;
            std::vector<rt_rect> rects((size_t)world);
            for (int r = 0; r < world; ++r) rects[(size_t)r] = rttile::tileRect(W, H, world, r, halo);
            if (tiled) {
                tracer.tile = rttile::growRect(rects[(size_t)rank], halo, W, H);
                filter.tile = rects[(size_t)rank];
            }
            tracer.timeOverride = o.time0 + o.timeStep * (float)f;
            tracer.freezeTime = o.freezeTime && f > 0;
            tracer.trace(s.frameNumber);
            if (filtered) {
                filter.filter();
                filter.setTextureRead();
            } else {
                framebuffer.setTextureRead();
            }
            // which image is gathered comes from this frame's scripted state, the same on every rank — not from
            // FrameBuffer::readTexture, a process-wide static that the GPU threads write at their own pace
            if (tiled)
                rtCheck(api.gather_tiles(ctx, filtered ? (int)RT_IMAGE_FILTERED : (int)RT_IMAGE_COLOR, rects.data(), world, 0),
                        "rt_gather_tiles");
        }
        rtCheck(api.sync(ctx), "rt_sync");

        if (rank == 0) {
            const unsigned shown = tiled ? (unsigned)RT_IMAGE_ASSEMBLED : (filtered ? (unsigned)RT_IMAGE_FILTERED : (unsigned)RT_IMAGE_COLOR);
            if (!tiled) framebuffer.saveToImage(o.out + "_linear.png");
            framebuffer.saveTonemapped(o.out + ".png", shown, exposure);
            if (!o.dumpRaw.empty()) {
                const size_t n = (size_t)W * H;
                std::vector<unsigned char> buf(n * 16);
                auto dump = [&](int image, size_t pb, const char *suffix) {
                    rtCheck(api.readback(ctx, image, buf.data(), n * pb), "rt_readback");
                    if (!dumpFile(o.dumpRaw + suffix, buf.data(), n * pb)) std::cerr << "ERROR::FILE::COULD NOT WRITE " << o.dumpRaw << suffix << std::endl;
                };
                if (tiled) {
                    dump(RT_IMAGE_ASSEMBLED, filtered ? 8 : 16, filtered ? "_assembled.f16" : "_assembled.f32");
                } else {
                    dump(RT_IMAGE_COLOR, 16, "_color.f32");
                    dump(RT_IMAGE_NORMAL, 4, "_normal.i8");
                    dump(RT_IMAGE_POSITION, 8, "_pos.f16");
                    if (filtered) dump(RT_IMAGE_FILTERED, 8, "_filtered.f16");
                }
            }
        }
        rt_timings t;
        rtCheck(api.timings_read(ctx, &t, 0), "rt_timings_read");
        const rt_counters c = tracer.counters();
        std::ostringstream line;
        line.setf(std::ios::fixed);
        line.precision(3);
        line << (tiled ? "gpu " + std::to_string(rank) + ": " : std::string()) << W << " x " << H << ", " << o.frames
             << " frames, " << o.bounces << " bounces: trace " << t.ms[RT_TIME_TRACE] << " ms, filter " << t.ms[RT_TIME_FILTER]
             << " ms (" << t.launches[RT_TIME_FILTER] << " passes); " << c.rays << " rays, "
             << (t.ms[RT_TIME_TRACE] > 0 ? c.rays / (t.ms[RT_TIME_TRACE] * 1e3) : 0.0) << " Mrays/s\n";
        std::cout << line.str() << std::flush;
        rtReleaseCurrentContext();
        sh.status[(size_t)rank] = 0;
    } catch (const RankAborted &e) {
        sh.errors[(size_t)rank] = e.what();
        sh.status[(size_t)rank] = 2;
    } catch (const std::exception &e) {
        sh.errors[(size_t)rank] = e.what();
        sh.status[(size_t)rank] = 1;
        sh.abort.store(true);
    }
    {
        std::lock_guard<std::mutex> lock(sh.m);
        sh.finished.fetch_add(1);
    }
    sh.cv.notify_all();
}

} // namespace

int main(int argc, char **argv) {
    if (argc < 2) {
        std::cout << "Usage:\t" << argv[0] << "\tobj_file_path [options]" << std::endl;
        return -1;
    }
    Options o;
    o.obj = argv[1];
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--size") std::sscanf(next(), "%dx%d", &o.width, &o.height);
        else if (a == "--frames") o.frames = std::atoi(next());
        else if (a == "--bounces") o.bounces = std::atoi(next());
        else if (a == "--filter") o.filterIters = std::atoi(next());
        else if (a == "--exposure") o.exposure = (float)std::atof(next());
        else if (a == "--auto-camera") o.autoCamera = true;
        else if (a == "--light") o.light = true;
        else if (a == "--out") o.out = next();
        else if (a == "--res") o.res = next();
        else if (a == "--kernel") o.kernel = next();
        else if (a == "--rect-light") {
            rt_rect_light l;
            if (std::sscanf(next(), "%f,%f,%f,%f,%f,%f,%f,%f,%f", &l.c[0], &l.c[1], &l.c[2], &l.x[0], &l.x[1], &l.x[2], &l.y[0],
                            &l.y[1], &l.y[2]) != 9) {
                std::cerr << "--rect-light wants nine comma-separated numbers (corner, edge x, edge y)" << std::endl;
                return -1;
            }
            o.rectLights.push_back(l);
        } else if (a == "--light-intensity") o.lightIntensity = (float)std::atof(next());
        else if (a == "--roughness") o.roughness = (float)std::atof(next());
        else if (a == "--watertight") o.launchFlags |= RT_FLAG_WATERTIGHT;      // opt-in variants, not the reference's algorithm
        else if (a == "--fast-shadows") o.launchFlags |= RT_FLAG_FAST_SHADOWS;
        else if (a == "--weighted-lights") o.launchFlags |= RT_FLAG_WEIGHTED_LIGHTS;
        else if (a == "--blue-noise") o.launchFlags |= RT_FLAG_BLUE_NOISE; // trace3 on the blue-noise sampler (tables from --res)
        else if (a == "--max-follows") o.maxFollows = std::atoi(next());
        else if (a == "--reset-at") o.resetAt = std::atoi(next());
        else if (a == "--freeze-time") o.freezeTime = true;
        else if (a == "--dump-raw") o.dumpRaw = next();
        else if (a == "--gpus") o.gpus = std::atoi(next());
        else if (a == "--time0") o.time0 = (float)std::atof(next());
        else if (a == "--keys") {
            std::stringstream ss(next());
            std::string item;
            while (std::getline(ss, item, ',')) {
                const size_t c = item.find(':');
                if (c == std::string::npos) { std::cerr << "--keys wants FRAME:KEY items" << std::endl; return -1; }
                o.keys.emplace_back(std::atoi(item.substr(0, c).c_str()), item.substr(c + 1));
            }
        } else { std::cerr << "unknown option " << a << std::endl; return -1; }
    }
    if (o.kernel != "trace3" && o.kernel != "trace" && o.kernel != "trace1" && o.kernel != "trace2" && o.kernel != "trace3ltc") {
        std::cerr << "unknown kernel " << o.kernel << std::endl;
        return -1;
    }
    if (o.gpus < 0 || o.gpus > 64 || o.frames < 1 || o.width < 1 || o.height < 1) {
        std::cerr << "bad --gpus / --frames / --size" << std::endl;
        return -1;
    }
    try {
        ModelLoader loader(o.obj.c_str());
        if (loader.triangles.empty()) {
            std::cerr << "ERROR::FILE::NO TRIANGLES LOADED" << std::endl;
            return -1;
        }
        rtm::vec3 lo, hi;
        loader.bounds(lo, hi);
        const rtm::vec3 d = hi - lo;
        const float diag = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
        if (o.light && loader.emitters.empty())
            loader.addQuadLight(rtm::vec3((lo.x + hi.x) * 0.5f, hi.y + 0.1f * diag, (lo.z + hi.z) * 0.5f),
                                0.25f * d.x, 0.25f * d.z, rtm::vec3(0.553949f, 0.914877f, 1.0f));
        QTable qt(loader.numObjs);
        qt.makeVoronoi(loader.triangles_per_obj);
        BVHContainer bvh(loader.triangles, loader.materialsVector, loader.emitters, qt.normal_per_voronoi,
                         qt.voronoi_centres);

        rtm::vec3 eye(0.0f, 2.2f, 3.5f), lookAt(0.0f, 3.0f, 0.0f);
        if (o.autoCamera) {
            lookAt = (lo + hi) * 0.5f;
            eye = lookAt + rtm::vec3(0.3f, 0.4f, 1.0f) * (1.5f * diag);
        }
        Shared sh(Camera(eye, lookAt, rtm::vec3(0.0f, 1.0f, 0.0f), 60.0f, (float)o.width, (float)o.height, 1.0f, 2.0f));
        sh.o = &o;
        sh.loader = &loader;
        sh.qt = &qt;
        sh.bvh = &bvh;
        if (!scriptStates(o, sh.camera0, sh.states)) return -1;
        for (const FrameState &s : sh.states)
            if (s.filterOn) sh.halo = std::max(sh.halo, rttile::filterHalo(s.iterations));
        const int threads = std::max(o.gpus, 1);
        sh.status.assign((size_t)threads, 1);
        sh.errors.assign((size_t)threads, std::string());
        auto report = [&]() {
            int rc = 0;
            for (int r = 0; r < threads; ++r)
                if (sh.status[(size_t)r]) {
                    std::cerr << "gpu " << r << ": " << (sh.errors[(size_t)r].empty() ? "did not finish" : sh.errors[(size_t)r]) << std::endl;
                    rc = 1;
                }
            return rc;
        };
        if (o.gpus > 0) {
            rtCheck(rtApi().comm_unique_id(sh.commId, RT_COMM_ID_BYTES), "rt_comm_unique_id");
            std::vector<std::thread> pool;
            for (int r = 0; r < o.gpus; ++r) pool.emplace_back(renderOn, std::ref(sh), r, o.gpus);
            {
                // all done, or one failed: then the others get ten seconds to notice (they check once per frame); a thread
                // stuck in a collective that waits for the failed rank never will, so the process exits without it
                std::unique_lock<std::mutex> lock(sh.m);
                sh.cv.wait(lock, [&] { return sh.finished.load() == o.gpus || sh.abort.load(); });
                if (sh.finished.load() != o.gpus)
                    sh.cv.wait_for(lock, std::chrono::seconds(10), [&] { return sh.finished.load() == o.gpus; });
            }
            if (sh.finished.load() != o.gpus) {
                report();
                std::cerr << "rt_headless: a GPU thread failed and " << (o.gpus - sh.finished.load())
                          << " other(s) did not return; exiting" << std::endl;
                std::cerr.flush();
                std::_Exit(1);
            }
            for (std::thread &t : pool) t.join();
        } else {
            renderOn(sh, 0, 0);
        }
        return report();
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
}
