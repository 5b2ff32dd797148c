// main.cpp — headless counterpart of the reference's window loop (reference src/main.cpp:6-138):
// same objects built in the same order, `frames` iterations of trace (+ filter), then PNGs.
//   rt_headless scene.obj [--size WxH] [--frames N] [--bounces B] [--filter ITERS] [--exposure E]
//               [--auto-camera] [--light] [--out PREFIX] [--res DIR] [--kernel trace3|trace|trace1|trace2]
// --kernel selects one of the reference's other kernels (its analytic box scenes; the obj file is still loaded,
// as the reference's main does, but not traced); trace1 / trace2 read their tables from --res (default "res").
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "render.hpp"
#include "rt_loader.hpp"
#include "scene.hpp"

int main(int argc, char **argv) {
    if (argc < 2) {
        std::cout << "Usage:\t" << argv[0] << "\tobj_file_path [options]" << std::endl;
        return -1;
    }
    int WIDTH = 512, HEIGHT = 512, frames = 4, bounces = 5, filterIters = 0;
    float exposure = 1.0f;
    bool autoCamera = false, light = false;
    std::string out = "frame", res = "res", kernel = "trace3";
    int launchFlags = 0, maxFollows = 1000;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--size") std::sscanf(next(), "%dx%d", &WIDTH, &HEIGHT);
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--bounces") bounces = std::atoi(next());
        else if (a == "--filter") filterIters = std::atoi(next());
        else if (a == "--exposure") exposure = (float)std::atof(next());
        else if (a == "--auto-camera") autoCamera = true;
        else if (a == "--light") light = true;
        else if (a == "--out") out = next();
        else if (a == "--res") res = next();
        else if (a == "--kernel") kernel = next();
        else if (a == "--watertight") launchFlags |= RT_FLAG_WATERTIGHT;      // opt-in variants, not the reference's algorithm
        else if (a == "--fast-shadows") launchFlags |= RT_FLAG_FAST_SHADOWS;
        else if (a == "--max-follows") maxFollows = std::atoi(next());
        else { std::cerr << "unknown option " << a << std::endl; return -1; }
    }
    try {
        FrameBuffer framebuffer(WIDTH, HEIGHT);
        Filter filter(WIDTH, HEIGHT, framebuffer.texture, framebuffer.normalTex, framebuffer.posTex);

        ModelLoader loader(argv[1]);
        if (loader.triangles.empty()) {
            std::cerr << "ERROR::FILE::NO TRIANGLES LOADED" << std::endl;
            return -1;
        }
        rtm::vec3 lo, hi;
        loader.bounds(lo, hi);
        const rtm::vec3 d = hi - lo;
        const float diag = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
        if (light && loader.emitters.empty())
            loader.addQuadLight(rtm::vec3((lo.x + hi.x) * 0.5f, hi.y + 0.1f * diag, (lo.z + hi.z) * 0.5f),
                                0.25f * d.x, 0.25f * d.z, rtm::vec3(0.553949f, 0.914877f, 1.0f));
        QTable qt(loader.numObjs);
        qt.makeVoronoi(loader.triangles_per_obj);
        BVHContainer bvh(loader.triangles, loader.materialsVector, loader.emitters, qt.normal_per_voronoi,
                         qt.voronoi_centres);

        rtm::vec3 eye(0.0f, 2.2f, 3.5f), lookAt(0.0f, 3.0f, 0.0f);
        if (autoCamera) {
            lookAt = (lo + hi) * 0.5f;
            eye = lookAt + rtm::vec3(0.3f, 0.4f, 1.0f) * (1.5f * diag);
        }
        Camera camera(eye, lookAt, rtm::vec3(0.0f, 1.0f, 0.0f), 60.0f, (float)WIDTH, (float)HEIGHT, 1.0f, 2.0f);

        const std::string main_cs = "src/shaders/" + kernel + ".cs";
        if (kernel != "trace3" && kernel != "trace" && kernel != "trace1" && kernel != "trace2") {
            std::cerr << "unknown kernel " << kernel << std::endl;
            return -1;
        }
        Tracer tracer = kernel == "trace3"
                            ? Tracer(&camera, &framebuffer, &bvh, &qt)
                            : Tracer(&camera, &framebuffer, &bvh, &qt,
                                     std::vector<const char *>{main_cs.c_str(), "src/shaders/random.glsl"});
        tracer.resourceDir = res;
        if (kernel == "trace1") tracer.genBlueNoiseBuffers();
        if (kernel == "trace2") tracer.genLtcData();
        tracer.maxBounces = bounces;
        tracer.launchFlags = launchFlags;
        tracer.maxFollows = maxFollows;
        filter.iterations = filterIters;
        rtCheck(rtApi().profile(rtCurrentContext(), 1), "rt_profile");

        for (int frameNumber = 0; frameNumber < frames; ++frameNumber) {
            tracer.timeOverride = 1.25f + 0.0625f * (float)frameNumber;
            tracer.trace(frameNumber);
        }
        if (filterIters > 0) {
            filter.filter();
            filter.setTextureRead();
        } else {
            framebuffer.setTextureRead();
        }
        framebuffer.saveToImage(out + "_linear.png");
        framebuffer.saveTonemapped(out + ".png", FrameBuffer::readTexture, exposure);

        rt_timings t;
        rtCheck(rtApi().timings_read(rtCurrentContext(), &t, 0), "rt_timings_read");
        const rt_counters c = tracer.counters();
        std::printf("%d x %d, %d frames, %d bounces: trace %.3f ms, filter %.3f ms (%d passes); %llu rays, %.1f Mrays/s\n",
                    WIDTH, HEIGHT, frames, bounces, t.ms[RT_TIME_TRACE], t.ms[RT_TIME_FILTER], (int)t.launches[RT_TIME_FILTER],
                    (unsigned long long)c.rays, t.ms[RT_TIME_TRACE] > 0 ? c.rays / (t.ms[RT_TIME_TRACE] * 1e3) : 0.0);
        rtReleaseCurrentContext();
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}
