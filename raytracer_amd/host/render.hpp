// render.hpp — the render-side classes of the reference with their GL calls replaced by the rt_*
// launcher (include/rt_launcher.h).  Same class names, constructors, public members and call order
// as the reference, so its main loop keeps working headless:
//   Camera         reference include/Camera.hpp:12-20,       src/Camera.cpp:5-17
//   ComputeShader  reference include/ComputeShader.hpp:21-38, src/ComputeShader.cpp:3-109
//   FrameBuffer    reference include/Framebuffer.hpp:19-39,   src/FrameBuffer.cpp:3-122
//   Tracer         reference include/Tracer.hpp:23-53,        src/Tracer.cpp:14-198
//   Filter         reference include/filter.hpp:15-44,        src/filter.cpp:3-110
// What used to be a GL object name (GLuint) is an rt_image / rt_slot number here.
#ifndef RT_AMD_RENDER_HPP
#define RT_AMD_RENDER_HPP

#include <chrono>
#include <map>
#include <string>
#include <vector>

#include "../../include/rt_launcher.h"
#include "rtmath.hpp"
#include "scene.hpp"

class Camera {
public:
    rtm::vec3 eye, lookAt, up;
    float fov, x, y, near, far;

    Camera(rtm::vec3 eye, rtm::vec3 lookAt, rtm::vec3 up, float fov, float x, float y, float near, float far);
    rtm::mat4 getViewMatrix();
    rtm::mat4 getPerspectiveMatrix();
};

// A "program" is one of the launcher's kernels, selected by the shader paths it is built from;
// uniforms set through set*() land in a name → value table that Tracer turns into rt_trace_params.
class ComputeShader {
public:
    unsigned ID;                    // enum rt_kernel
    std::vector<int> workGroupSize; // pixels covered by one thread block, as the launcher reports it

    ComputeShader(std::vector<const char *> shaderPathVector);
    void use();
    void setBool(const std::string &name, bool value) const;
    void setInt(const std::string &name, int value) const;
    void setFloat(const std::string &name, float value) const;
    void setVec3(const std::string &name, rtm::vec3 value) const;
    void setMat4(const std::string &name, rtm::mat4 value) const;

    // launcher side: the uniform table as launch parameters
    rt_trace_params params() const;

private:
    mutable std::map<std::string, int> ints;
    mutable std::map<std::string, float> floats;
    mutable std::map<std::string, rtm::vec3> vec3s;
};

class FrameBuffer {
    unsigned width, height;

public:
    unsigned texture;           // RT_IMAGE_COLOR
    unsigned normalTex, posTex; // RT_IMAGE_NORMAL, RT_IMAGE_POSITION

    FrameBuffer(int width, int height);
    void createTextureAttachment(int width, int height);
    void resizeTextureAttachment(int width, int height);
    void renderToFramebuffer() {}
    void renderToScreen() {}
    void saveToImage(std::string path); // 8-bit RGBA of the linear colour image, flipped (FrameBuffer.cpp:89-107)
    void deleteFramebuffer() {}
    void setTextureRead();
    rtm::ivec2 getTextureDims();

    // Not in the reference (its main.cpp tonemaps through quad.fs into a second FrameBuffer and the
    // screen, :94-100): tonemap `image` (RT_IMAGE_COLOR or RT_IMAGE_FILTERED) and write a PNG.
    void saveTonemapped(std::string path, unsigned image, float exposure);
    static unsigned readTexture; // what the last setTextureRead() bound to "unit 0"
};

class Tracer {
    std::chrono::time_point<std::chrono::system_clock> start;
    ComputeShader compute;
    Camera *camera;
    FrameBuffer *fbo;
    rtm::vec4 ray00, ray01, ray10, ray11;
    void updateRays();
    BVHContainer *bvh;
    QTable *qt;

public:
    Tracer(Camera *camera, FrameBuffer *fbo, BVHContainer *bvh, QTable *qt);
    // Not in the reference, where the shader list is a literal in the constructor (Tracer.cpp:14): the same
    // constructor with the list spelt out, e.g. {"src/shaders/trace1.cs", "src/shaders/random.glsl"}.  The
    // analytic kernels need no scene: bvh / qt may be null for them.
    Tracer(Camera *camera, FrameBuffer *fbo, BVHContainer *bvh, QTable *qt, std::vector<const char *> shaderPaths);
    void genBlueNoiseBuffers();
    void genBlueNoiseTexes();
    void genLtcData();
    void trace(int frameNumber);

    bool importanceSampled;
    bool useBlueNoise;
    // The three tunables reference src/main.cpp:57-59 hands to its input handler.  In the reference they are inert
    // (the guard of Tracer.cpp:134-135 and the key bindings of glfw_setter.cpp:187-206 are commented out, freezeTime is
    // never read); here they do what those commented-out lines describe, which at their default values is nothing:
    //   !accumulateSamples: trace() returns without dispatching once frameNumber > maxAccumulateSamples (Tracer.cpp:134);
    //   freezeTime: the `time` uniform keeps the value it had when the flag was set, so every frame draws the same
    //   random numbers (with blendFactor n/(n+1) the accumulated image then stays the first sample).
    bool freezeTime = false;
    bool accumulateSamples;
    int maxAccumulateSamples;
    int maxBounces;

    // Not in the reference: the `time` uniform is wall-clock seconds there (Tracer.cpp:136-137,145);
    // a headless, reproducible run pins it per frame.  NaN = use the wall clock.
    float timeOverride;
    // uniforms of trace.cs / trace1.cs / trace2.cs whose setters are commented out in Tracer.cpp:147-152
    float phongExponent = 128.0f, specularFactor = 0.2f, roughness = 0.5f;
    // Not in the reference: launch options of the MI355X launcher (include/rt_launcher.h).  launchFlags: RT_FLAG_*
    // (e.g. RT_FLAG_WATERTIGHT, RT_FLAG_FAST_SHADOWS: opt-in variants that are not the reference's algorithm);
    // maxFollows: MAX_FOLLOWS of trace3.cs:214, to be lifted for scenes of millions of triangles.
    int launchFlags = 0, maxFollows = 1000;
    // Not in the reference (single GPU): the part of the image this context traces; x1 <= x0 means the whole image.
    rt_rect tile = {0, 0, 0, 0};
    std::string resourceDir = "res"; // where the sample tables / LUTs are read from
    rt_counters counters();
    // Not in the reference: the rectangle lights of RT_KERNEL_TRACE3_LTC (a Tracer built from "trace3ltc.cs")
    void setRectLights(const std::vector<rt_rect_light> &lights, float intensity = 6.0f);

private:
    float lastTime = 0.0f; // the `time` of the last dispatch (what freezeTime freezes)
};

class Filter {
    int width, height;

public:
    unsigned inputTex, normTex, posTex, ftex0, ftex1;
    unsigned FBO;
    int iterations;
    int readFrom;
    float c_phi0, n_phi0, p_phi0;
    // Not in the reference (single GPU): the part of the image that must be final after filter(); x1 <= x0 = all of it.
    rt_rect tile = {0, 0, 0, 0};

    Filter(unsigned width, unsigned height, unsigned inputTex, unsigned normTex, unsigned posTex);
    void createTextureAttachments(int width, int height) { (void)width; (void)height; }
    void filter();
    void setTextureRead();
};

#endif
