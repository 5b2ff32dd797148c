// rt_loader.cpp — see rt_loader.hpp.
#include "rt_loader.hpp"

#include <dlfcn.h>

#include <cstdlib>
#include <iostream>
#include <mutex>
#include <stdexcept>
#include <string>

namespace {

RtApi g_api;
bool g_loaded = false;
rt_ctx *g_ctx = nullptr;            // process-wide default context
thread_local rt_ctx *t_ctx = nullptr; // this thread's own context (rtMakeContextCurrent)
std::mutex g_mutex;

std::string hostLibraryDir() {
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&hostLibraryDir), &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        const size_t s = p.find_last_of('/');
        if (s != std::string::npos) return p.substr(0, s);
    }
    return ".";
}

template <class F> void bindSym(void *lib, const char *name, F &fn) {
    void *p = dlsym(lib, name);
    if (!p) throw std::runtime_error(std::string("librt_hip.so does not export ") + name);
    fn = reinterpret_cast<F>(p);
}

} // namespace

const RtApi &rtApi() {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_loaded) return g_api;
    const char *env = std::getenv("RT_HIP_LIB");
    const std::string path = env ? std::string(env) : hostLibraryDir() + "/librt_hip.so";
    void *lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        const std::string msg = "ERROR::RT::GPU LAUNCHER NOT LOADED: " + path + ": " + dlerror() +
                                " (build it with `make hip`; there is no CPU fallback)";
        std::cerr << msg << std::endl;
        throw std::runtime_error(msg);
    }
    bindSym(lib, "rt_create", g_api.create);
    bindSym(lib, "rt_destroy", g_api.destroy);
    bindSym(lib, "rt_last_error", g_api.last_error);
    bindSym(lib, "rt_set_stream", g_api.set_stream);
    bindSym(lib, "rt_upload", g_api.upload);
    bindSym(lib, "rt_set_table", g_api.set_table);
    bindSym(lib, "rt_set_rect_lights", g_api.set_rect_lights);
    bindSym(lib, "rt_resize", g_api.resize);
    bindSym(lib, "rt_clear", g_api.clear);
    bindSym(lib, "rt_trace", g_api.trace);
    bindSym(lib, "rt_trace_batch", g_api.trace_batch);
    bindSym(lib, "rt_filter", g_api.filter);
    bindSym(lib, "rt_tonemap", g_api.tonemap);
    bindSym(lib, "rt_readback", g_api.readback);
    bindSym(lib, "rt_counters_read", g_api.counters_read);
    bindSym(lib, "rt_counters_reset", g_api.counters_reset);
    bindSym(lib, "rt_profile", g_api.profile);
    bindSym(lib, "rt_timings_read", g_api.timings_read);
    bindSym(lib, "rt_sync", g_api.sync);
    bindSym(lib, "rt_work_group_size", g_api.work_group_size);
    bindSym(lib, "rt_comm_unique_id", g_api.comm_unique_id);
    bindSym(lib, "rt_comm_init", g_api.comm_init);
    bindSym(lib, "rt_comm_destroy", g_api.comm_destroy);
    bindSym(lib, "rt_gather_tiles", g_api.gather_tiles);
    bindSym(lib, "rt_walk_stop", g_api.walk_stop);
    bindSym(lib, "rt_set_frames_in_flight", g_api.set_frames_in_flight);
    g_loaded = true;
    return g_api;
}

static rt_ctx *createOn(const RtApi &api, int dev) {
    rt_ctx *c = nullptr;
    const int rc = api.create(dev, &c);
    if (rc != 0 || !c) {
        const std::string msg = "ERROR::RT::NO USABLE GPU (rt_create(" + std::to_string(dev) + ") = " +
                                std::to_string(rc) + "); there is no CPU fallback";
        std::cerr << msg << std::endl;
        throw std::runtime_error(msg);
    }
    return c;
}

rt_ctx *rtMakeContextCurrent(int device) {
    const RtApi &api = rtApi();
    if (t_ctx) api.destroy(t_ctx);
    t_ctx = nullptr;
    t_ctx = createOn(api, device);
    return t_ctx;
}

rt_ctx *rtCurrentContext() {
    const RtApi &api = rtApi();
    if (t_ctx) return t_ctx;
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_ctx) {
        const char *d = std::getenv("RT_DEVICE");
        if (!d) d = std::getenv("LOCAL_RANK");
        const int dev = d ? std::atoi(d) : 0;
        g_ctx = createOn(api, dev);
    }
    return g_ctx;
}

void rtReleaseCurrentContext() {
    if (t_ctx) {
        if (g_loaded) g_api.destroy(t_ctx);
        t_ctx = nullptr;
        return;
    }
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_ctx && g_loaded) g_api.destroy(g_ctx);
    g_ctx = nullptr;
}

void rtCheck(int rc, const char *what) {
    if (rc == 0) return;
    std::string msg = std::string("ERROR::RT::") + what + " failed (" + std::to_string(rc) + ")";
    rt_ctx *cur = t_ctx ? t_ctx : g_ctx;
    if (cur && g_loaded) msg += std::string(": ") + g_api.last_error(cur);
    std::cerr << msg << std::endl;
    throw std::runtime_error(msg);
}
