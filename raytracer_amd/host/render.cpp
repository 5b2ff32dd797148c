// render.cpp — see render.hpp.
#include "render.hpp"

#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>

#include "png_write.hpp"
#include "rt_loader.hpp"

using rtm::vec3;
using rtm::vec4;

// ---------------------------------------------------------------------------------------
Camera::Camera(vec3 eye, vec3 lookAt, vec3 up, float fov, float x, float y, float near, float far)
    : eye(eye), lookAt(lookAt), up(up), fov(fov), x(x), y(y), near(near), far(far) {}

rtm::mat4 Camera::getViewMatrix() { return rtm::lookAt(eye, lookAt, up); }

rtm::mat4 Camera::getPerspectiveMatrix() { return rtm::perspective(rtm::radians(fov), x / y, near, far); }

// ---------------------------------------------------------------------------------------
ComputeShader::ComputeShader(std::vector<const char *> shaderPathVector) {
    // The reference compiles the listed GLSL files (ComputeShader.cpp:12-63); here the first path
    // names the kernel family and nothing is read or compiled.
    ID = RT_KERNEL_TRACE3;
    if (!shaderPathVector.empty() && shaderPathVector[0]) {
        const std::string p(shaderPathVector[0]);
        if (p.find("trace3ltc") != std::string::npos) ID = RT_KERNEL_TRACE3_LTC; // (no file of the reference: rt_launcher.h)
        else if (p.find("trace3") != std::string::npos) ID = RT_KERNEL_TRACE3;
        else if (p.find("trace1") != std::string::npos) ID = RT_KERNEL_TRACE1;
        else if (p.find("trace2") != std::string::npos) ID = RT_KERNEL_TRACE2;
        else if (p.find("trace") != std::string::npos) ID = RT_KERNEL_TRACE;
        else std::cerr << "ERROR::SHADER::FILE_NOT_SUCCESFULLY_READ" << std::endl; // unknown program: keep going
    }
    workGroupSize.resize(3);
    rtApi().work_group_size((int)ID, &workGroupSize[0]);
}

void ComputeShader::use() {}
void ComputeShader::setBool(const std::string &name, bool value) const { ints[name] = (int)value; }
void ComputeShader::setInt(const std::string &name, int value) const { ints[name] = value; }
void ComputeShader::setFloat(const std::string &name, float value) const { floats[name] = value; }
void ComputeShader::setVec3(const std::string &name, vec3 value) const { vec3s[name] = value; }
void ComputeShader::setMat4(const std::string &, rtm::mat4) const {}

rt_trace_params ComputeShader::params() const {
    rt_trace_params p;
    std::memset(&p, 0, sizeof p);
    p.size = sizeof p;
    p.kernel = (int)ID;
    auto v3 = [&](const char *n, float *dst) {
        auto it = vec3s.find(n);
        if (it != vec3s.end()) { dst[0] = it->second.x; dst[1] = it->second.y; dst[2] = it->second.z; }
    };
    auto f1 = [&](const char *n, float &dst) { auto it = floats.find(n); if (it != floats.end()) dst = it->second; };
    auto i1 = [&](const char *n, int &dst) { auto it = ints.find(n); if (it != ints.end()) dst = it->second; };
    v3("eye", p.eye); v3("ray00", p.ray00); v3("ray01", p.ray01); v3("ray10", p.ray10); v3("ray11", p.ray11);
    f1("time", p.time); f1("blendFactor", p.blendFactor);
    i1("importanceSampled", p.importanceSampled); i1("numPoints", p.numPoints); i1("numEmitters", p.numEmitters);
    i1("maxPhi", p.maxPhi); i1("maxTheta", p.maxTheta); i1("maxBounces", p.maxBounces);
    i1("useBlueNoise", p.useBlueNoise); i1("multipleImportanceSampled", p.multipleImportanceSampled);
    i1("frameIndex", p.frameIndex);
    f1("phongExponent", p.phongExponent); f1("specularFactor", p.specularFactor); f1("roughness", p.roughness);
    p.maxFollows = 1000;
    // what rand_seeds[-1..-3] read on the reference's oracle platform (see rt_launcher.h)
    std::memcpy(&p.oobSeeds[0], &p.maxBounces, 4);
    p.oobSeeds[1] = p.blendFactor;
    p.oobSeeds[2] = p.time;
    return p;
}

// ---------------------------------------------------------------------------------------
unsigned FrameBuffer::readTexture = RT_IMAGE_COLOR;

FrameBuffer::FrameBuffer(int width, int height) {
    this->width = width;
    this->height = height;
    texture = RT_IMAGE_COLOR;
    normalTex = RT_IMAGE_NORMAL;
    posTex = RT_IMAGE_POSITION;
    createTextureAttachment(width, height);
}

void FrameBuffer::createTextureAttachment(int width, int height) {
    rtCheck(rtApi().resize(rtCurrentContext(), width, height), "rt_resize");
}

void FrameBuffer::resizeTextureAttachment(int width, int height) {
    this->width = width;
    this->height = height;
    createTextureAttachment(width, height);
}

void FrameBuffer::setTextureRead() { readTexture = texture; }

rtm::ivec2 FrameBuffer::getTextureDims() { return rtm::ivec2((int)width, (int)height); }

void FrameBuffer::saveToImage(std::string path) {
    const size_t n = (size_t)width * height;
    std::vector<float> rgba(n * 4);
    rtCheck(rtApi().readback(rtCurrentContext(), RT_IMAGE_COLOR, rgba.data(), n * 16), "rt_readback");
    std::vector<uint8_t> out(n * 4);
    for (size_t i = 0; i < n * 4; ++i) { // glGetTexImage(..., GL_UNSIGNED_BYTE): clamp to [0,1], scale, round
        float v = rgba[i];
        v = !(v > 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v);
        out[i] = (uint8_t)std::lrintf(v * 255.0f);
    }
    if (rt_write_png_rgba(path.c_str(), (int)width, (int)height, out.data(), true) == 0)
        std::cout << "Image saved to: " << path << std::endl;
    else std::cerr << "ERROR::FILE::COULD NOT WRITE " << path << std::endl;
}

void FrameBuffer::saveTonemapped(std::string path, unsigned image, float exposure) {
    const size_t n = (size_t)width * height;
    rtCheck(rtApi().tonemap(rtCurrentContext(), (int)image, exposure, nullptr), "rt_tonemap");
    std::vector<uint8_t> out(n * 4);
    rtCheck(rtApi().readback(rtCurrentContext(), RT_IMAGE_LDR, out.data(), n * 4), "rt_readback");
    if (rt_write_png_rgba(path.c_str(), (int)width, (int)height, out.data(), true) == 0)
        std::cout << "Image saved to: " << path << std::endl;
    else std::cerr << "ERROR::FILE::COULD NOT WRITE " << path << std::endl;
}

// ---------------------------------------------------------------------------------------
Tracer::Tracer(Camera *camera, FrameBuffer *fbo, BVHContainer *bvh, QTable *qt)
    : Tracer(camera, fbo, bvh, qt,
             std::vector<const char *>{"src/shaders/trace3.cs", "src/shaders/geometry.glsl", "src/shaders/random.glsl",
                                       "src/shaders/qtablehelper.glsl"}) {}

Tracer::Tracer(Camera *camera, FrameBuffer *fbo, BVHContainer *bvh, QTable *qt, std::vector<const char *> shaderPaths)
    : compute(shaderPaths),
      camera(camera), fbo(fbo), bvh(bvh), qt(qt), importanceSampled(true), useBlueNoise(true),
      accumulateSamples(true), maxAccumulateSamples(64), maxBounces(5),
      timeOverride(std::numeric_limits<float>::quiet_NaN()) {
    start = std::chrono::system_clock::now();
    // The reference's BVHContainer / QTable constructors upload with glBufferData (BVH.cpp:378-397,
    // QTable.cpp:138-146); the byte streams are handed to the launcher here instead.
    const RtApi &api = rtApi();
    rt_ctx *ctx = rtCurrentContext();
    auto up = [&](unsigned slot, const void *p, size_t n) { rtCheck(api.upload(ctx, (int)slot, p, n), "rt_upload"); };
    if (!bvh) return; // analytic kernels
    up(bvh->nodesSsbo, bvh->nodesBuffer.data(), bvh->nodesBuffer.size() * sizeof(GPUNode));
    up(bvh->trianglesSsbo, bvh->trianglesBuffer.data(), bvh->trianglesBuffer.size() * sizeof(GPUTriangle));
    up(bvh->materialsSsbo, bvh->mtlBuffer.data(), bvh->mtlBuffer.size() * sizeof(GPUMaterial));
    up(bvh->emittersSsbo, bvh->emittersBuffer.data(), bvh->emittersBuffer.size() * sizeof(GPUTriangle));
    up(bvh->normalsVoronoiSsbo, bvh->normalsVoronoiBuffer.data(), bvh->normalsVoronoiBuffer.size() * 4);
    up(bvh->voronoiCentresSsbo, bvh->voronoiCentresBuffer.data(), bvh->voronoiCentresBuffer.size() * 4);
    if (qt) {
        up(qt->QTableSsbo, qt->QTableBuffer.data(), qt->QTableBuffer.size() * 4);
        up(qt->QVisitsSsbo, qt->QVisitsBuffer.data(), qt->QVisitsBuffer.size() * 4);
    }
}

static bool readWhole(const std::string &path, std::vector<char> &out) {
    std::ifstream is(path, std::ios::binary);
    if (!is) {
        std::cerr << "ERROR::FILE::COULD NOT OPEN " << path << std::endl;
        return false;
    }
    out.assign(std::istreambuf_iterator<char>(is), std::istreambuf_iterator<char>());
    return true;
}

void Tracer::genLtcData() { // Tracer.cpp:23-48
    std::vector<char> b;
    if (readWhole(resourceDir + "/ltc1.data", b))
        rtCheck(rtApi().set_table(rtCurrentContext(), RT_TABLE_LTC1, b.data(), b.size()), "rt_set_table(ltc1)");
    if (readWhole(resourceDir + "/ltc2.data", b))
        rtCheck(rtApi().set_table(rtCurrentContext(), RT_TABLE_LTC2, b.data(), b.size()), "rt_set_table(ltc2)");
}

// Not in the reference (its trace2.cs holds the rectangles as shader constants, :47-64): the lights of RT_KERNEL_TRACE3_LTC
void Tracer::setRectLights(const std::vector<rt_rect_light> &lights, float intensity) {
    rtCheck(rtApi().set_rect_lights(rtCurrentContext(), lights.data(), (int)lights.size(), intensity), "rt_set_rect_lights");
}

void Tracer::genBlueNoiseBuffers() { // Tracer.cpp:50-90
    std::vector<char> b;
    if (readWhole(resourceDir + "/sobol_256_256_4spp.data", b))
        rtCheck(rtApi().set_table(rtCurrentContext(), RT_TABLE_SOBOL, b.data(), b.size()), "rt_set_table(sobol)");
    if (readWhole(resourceDir + "/scramble_128_128_8_4spp.data", b))
        rtCheck(rtApi().set_table(rtCurrentContext(), RT_TABLE_SCRAMBLE, b.data(), b.size()), "rt_set_table(scramble)");
    if (readWhole(resourceDir + "/ranking_128_128_8_4spp.data", b))
        rtCheck(rtApi().set_table(rtCurrentContext(), RT_TABLE_RANKING, b.data(), b.size()), "rt_set_table(ranking)");
}

void Tracer::genBlueNoiseTexes() { genBlueNoiseBuffers(); } // the PNG texture of Tracer.cpp:92-105 is never sampled

void Tracer::updateRays() { // Tracer.cpp:110-130
    rtm::mat4 m = camera->getPerspectiveMatrix() * camera->getViewMatrix();
    m = rtm::transpose(rtm::inverse(m));
    auto corner = [&](float cx, float cy) {
        vec4 r = vec4(cx, cy, 0, 1) * m;
        r = r / r.w;
        r = r - vec4(camera->eye, 0.0f);
        return r;
    };
    ray00 = corner(-1, -1);
    ray01 = corner(-1, +1);
    ray10 = corner(+1, -1);
    ray11 = corner(+1, +1);
}

void Tracer::trace(int frameNumber) { // Tracer.cpp:132-198
    if (!accumulateSamples && frameNumber > maxAccumulateSamples) return; // Tracer.cpp:134-135 (commented out there)
    auto end = std::chrono::system_clock::now();
    std::chrono::duration<double> elapsed_seconds = end - start;
    updateRays();
    compute.use();
    compute.setVec3("ray00", vec3(ray00.x, ray00.y, ray00.z));
    compute.setVec3("ray01", vec3(ray01.x, ray01.y, ray01.z));
    compute.setVec3("ray10", vec3(ray10.x, ray10.y, ray10.z));
    compute.setVec3("ray11", vec3(ray11.x, ray11.y, ray11.z));
    compute.setVec3("eye", camera->eye);
    if (!freezeTime) lastTime = std::isnan(timeOverride) ? (float)elapsed_seconds.count() : timeOverride;
    compute.setFloat("time", lastTime);
    compute.setFloat("blendFactor", (float)frameNumber / (frameNumber + 1.0f));
    compute.setInt("importanceSampled", int(importanceSampled));
    compute.setInt("numPoints", MAX_POINTS);
    compute.setInt("numEmitters", bvh ? (int)bvh->emitters.size() : 0);
    compute.setInt("maxPhi", MAX_PHI);
    compute.setInt("maxTheta", MAX_THETA);
    compute.setInt("maxBounces", maxBounces);
    compute.setInt("useBlueNoise", int(useBlueNoise));
    compute.setInt("frameIndex", frameNumber);
    compute.setInt("multipleImportanceSampled", int(importanceSampled));
    compute.setFloat("phongExponent", phongExponent);
    compute.setFloat("specularFactor", specularFactor);
    compute.setFloat("roughness", roughness);
    rt_trace_params p = compute.params();
    p.flags = launchFlags;
    p.maxFollows = maxFollows;
    // image / buffer bindings of Tracer.cpp:165-180 are implicit in the context; the dispatch:
    rtCheck(rtApi().trace(rtCurrentContext(), &p, tile.x1 > tile.x0 ? &tile : nullptr), "rt_trace");
}

rt_counters Tracer::counters() {
    rt_counters c;
    rtCheck(rtApi().counters_read(rtCurrentContext(), &c), "rt_counters_read");
    return c;
}

// ---------------------------------------------------------------------------------------
Filter::Filter(unsigned width, unsigned height, unsigned inputTex, unsigned normTex, unsigned posTex)
    : iterations(2), c_phi0(3.3f), n_phi0(1E-2f), p_phi0(5.5f) { // filter.cpp:3
    this->width = (int)width;
    this->height = (int)height;
    this->inputTex = inputTex;
    this->normTex = normTex;
    this->posTex = posTex;
    ftex0 = ftex1 = RT_IMAGE_FILTERED;
    FBO = 0;
    readFrom = (int)inputTex;
}

void Filter::filter() { // filter.cpp:68-104
    rtCheck(rtApi().filter(rtCurrentContext(), iterations, c_phi0, n_phi0, p_phi0, tile.x1 > tile.x0 ? &tile : nullptr),
            "rt_filter");
    readFrom = iterations > 0 ? RT_IMAGE_FILTERED : (int)inputTex;
}

void Filter::setTextureRead() { FrameBuffer::readTexture = (unsigned)readFrom; }
