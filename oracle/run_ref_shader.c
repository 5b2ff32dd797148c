/* run_ref_shader.c — TEST INFRASTRUCTURE (oracle side), never linked into the product.
 *
 * Runs the REFERENCE's own shaders on Mesa llvmpipe and dumps their outputs, so that golden
 * vectors and the CPU baseline come from the reference itself.
 * The shader files are read at run time from --ref-dir (default /root/reference); nothing of
 * them is stored in this repository.  Host-side behaviour restated here:
 *   - program build: one shader object per file, linked together  (reference src/ComputeShader.cpp:38-63)
 *   - uniforms / image + SSBO bindings / dispatch grid             (reference src/Tracer.cpp:139-183)
 *   - à-trous pass loop, per-pass uniforms, ping-pong              (reference src/filter.cpp:27-44,68-104)
 *
 * The reference outputs neither the primary-hit triangle index nor the node-visit count, so
 * with instrument=1 three integer images are added by text substitution at load time
 * (arithmetic untouched): unit 3 = SI.tridx of bounce 0, unit 4 = `iterations` of the primary
 * intersectGeometry, unit 5 = per-pixel (traversals, node visits, triangle tests, bits of rand_seeds[-1 - x%3]) — the
 * last one records what the reference's out-of-range rand_seeds reads (trace3.cs:284-300) return here.
 *
 * Usage: run_ref_shader job.txt      (key=value lines; floats may be C hex floats)
 *   ref_dir=/root/reference  width= height=  spp=  time0= dtime=  max_bounces=  importance=
 *   num_emitters=  eye=x,y,z ray00= ray01= ray10= ray11=
 *   nodes= tris= mtls= emitters= vnormals= vcentres= qtable= qvisits=   (file paths; optional ones may be absent)
 *   instrument=0|1  filter_iters=N  c_phi0= n_phi0= p_phi0=  out=prefix  repeat=R (timing)
 *   init_color=path (optional RGBA32F image to start from)  frame0=K (first frame number)
 * Outputs: <out>.color.f32 .normal.i8 .pos.f16 [.tridx.i32 .visits.i32 .stats.i32]
 *          [.filter<k>.f16 for each pass k] and timing lines on stdout.
 */
#define _POSIX_C_SOURCE 200809L
#include <GL/gl.h>
#include <GL/glext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gl_headless.h"

#define GLFN(type, name) static type p_##name
GLFN(PFNGLCREATESHADERPROC, glCreateShader);
GLFN(PFNGLSHADERSOURCEPROC, glShaderSource);
GLFN(PFNGLCOMPILESHADERPROC, glCompileShader);
GLFN(PFNGLGETSHADERIVPROC, glGetShaderiv);
GLFN(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog);
GLFN(PFNGLCREATEPROGRAMPROC, glCreateProgram);
GLFN(PFNGLATTACHSHADERPROC, glAttachShader);
GLFN(PFNGLLINKPROGRAMPROC, glLinkProgram);
GLFN(PFNGLGETPROGRAMIVPROC, glGetProgramiv);
GLFN(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog);
GLFN(PFNGLUSEPROGRAMPROC, glUseProgram);
GLFN(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation);
GLFN(PFNGLUNIFORM1IPROC, glUniform1i);
GLFN(PFNGLUNIFORM2IPROC, glUniform2i);
GLFN(PFNGLUNIFORM1FPROC, glUniform1f);
GLFN(PFNGLUNIFORM1FVPROC, glUniform1fv);
GLFN(PFNGLUNIFORM3FVPROC, glUniform3fv);
GLFN(PFNGLGENBUFFERSPROC, glGenBuffers);
GLFN(PFNGLBINDBUFFERPROC, glBindBuffer);
GLFN(PFNGLBUFFERDATAPROC, glBufferData);
GLFN(PFNGLBINDBUFFERBASEPROC, glBindBufferBase);
GLFN(PFNGLGETBUFFERSUBDATAPROC, glGetBufferSubData);
GLFN(PFNGLTEXSTORAGE2DPROC, glTexStorage2D);
GLFN(PFNGLBINDIMAGETEXTUREPROC, glBindImageTexture);
GLFN(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute);
GLFN(PFNGLMEMORYBARRIERPROC, glMemoryBarrier);
GLFN(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers);
GLFN(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer);
GLFN(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D);
GLFN(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus);
GLFN(PFNGLDRAWBUFFERSPROC, glDrawBuffers);
GLFN(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays);
GLFN(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray);
GLFN(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer);
GLFN(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray);
GLFN(PFNGLACTIVETEXTUREPROC, glActiveTexture);
GLFN(PFNGLCLEARTEXIMAGEPROC, glClearTexImage);
static const GLubyte *(*p_glGetString)(GLenum);
static GLenum (*p_glGetError)(void);
static void (*p_glGenTextures)(GLsizei, GLuint *);
static void (*p_glBindTexture)(GLenum, GLuint);
static void (*p_glTexParameteri)(GLenum, GLenum, GLint);
static void (*p_glTexSubImage2D)(GLenum, GLint, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, const void *);
static void (*p_glTexImage2D)(GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *);
static void (*p_glGetTexImage)(GLenum, GLint, GLenum, GLenum, void *);
static void (*p_glPixelStorei)(GLenum, GLint);
static void (*p_glFinish)(void);
static void (*p_glViewport)(GLint, GLint, GLsizei, GLsizei);
static void (*p_glDrawElements)(GLenum, GLsizei, GLenum, const void *);
static void (*p_glDisable)(GLenum);

#define LOAD(name) p_##name = glh_proc(#name)
static void loadGL(void) {
    LOAD(glCreateShader); LOAD(glShaderSource); LOAD(glCompileShader); LOAD(glGetShaderiv);
    LOAD(glGetShaderInfoLog); LOAD(glCreateProgram); LOAD(glAttachShader); LOAD(glLinkProgram);
    LOAD(glGetProgramiv); LOAD(glGetProgramInfoLog); LOAD(glUseProgram); LOAD(glGetUniformLocation);
    LOAD(glUniform1i); LOAD(glUniform2i); LOAD(glUniform1f); LOAD(glUniform1fv); LOAD(glUniform3fv);
    LOAD(glGenBuffers); LOAD(glBindBuffer); LOAD(glBufferData); LOAD(glBindBufferBase); LOAD(glGetBufferSubData); LOAD(glTexStorage2D);
    LOAD(glBindImageTexture); LOAD(glDispatchCompute); LOAD(glMemoryBarrier); LOAD(glGenFramebuffers);
    LOAD(glBindFramebuffer); LOAD(glFramebufferTexture2D); LOAD(glCheckFramebufferStatus); LOAD(glDrawBuffers);
    LOAD(glGenVertexArrays); LOAD(glBindVertexArray); LOAD(glVertexAttribPointer);
    LOAD(glEnableVertexAttribArray); LOAD(glActiveTexture); LOAD(glClearTexImage);
    LOAD(glGetString); LOAD(glGetError); LOAD(glGenTextures); LOAD(glBindTexture); LOAD(glTexParameteri);
    LOAD(glTexSubImage2D); LOAD(glTexImage2D); LOAD(glGetTexImage); LOAD(glPixelStorei); LOAD(glFinish);
    LOAD(glViewport); LOAD(glDrawElements); LOAD(glDisable);
}

/* ---------------------------------------------------------------- job file */
#define MAXKV 96
static char *jobKeys[MAXKV], *jobVals[MAXKV];
static int jobN;

static void jobLoad(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) { fprintf(stderr, "cannot open job %s\n", path); exit(2); }
    char line[4096];
    while (fgets(line, sizeof line, f) && jobN < MAXKV) {
        char *nl = strpbrk(line, "\r\n");
        if (nl) *nl = 0;
        if (!line[0] || line[0] == '#') continue;
        char *eq = strchr(line, '=');
        if (!eq) continue;
        *eq = 0;
        jobKeys[jobN] = strdup(line);
        jobVals[jobN] = strdup(eq + 1);
        jobN++;
    }
    fclose(f);
}
static const char *jobStr(const char *k, const char *def) {
    for (int i = 0; i < jobN; i++)
        if (!strcmp(jobKeys[i], k)) return jobVals[i];
    return def;
}
static int jobInt(const char *k, int def) { const char *s = jobStr(k, NULL); return s ? atoi(s) : def; }
static float jobFloat(const char *k, float def) { const char *s = jobStr(k, NULL); return s ? strtof(s, NULL) : def; }
static void jobVec3(const char *k, float v[3]) {
    const char *s = jobStr(k, NULL);
    if (!s) { fprintf(stderr, "job: missing %s\n", k); exit(2); }
    char *e;
    for (int i = 0; i < 3; i++) { v[i] = strtof(s, &e); s = (*e == ',') ? e + 1 : e; }
}

/* ---------------------------------------------------------------- helpers */
static char *readFile(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *b = malloc((size_t)sz + 1);
    if (fread(b, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(b); return NULL; }
    b[sz] = 0;
    fclose(f);
    if (n) *n = (size_t)sz;
    return b;
}
static void writeFile(const char *prefix, const char *suffix, const void *p, size_t n) {
    char path[4096];
    snprintf(path, sizeof path, "%s%s", prefix, suffix);
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(p, 1, n, f) != n) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fclose(f);
}
static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
/* Replace the first occurrence of `what` at or after `from` by `with`; exits if absent. */
static char *replaceOnce(char *src, const char *from, const char *what, const char *with) {
    char *start = src;
    if (from) {
        start = strstr(src, from);
        if (!start) { fprintf(stderr, "instrument: anchor not found: %s\n", from); exit(4); }
    }
    char *at = strstr(start, what);
    if (!at) { fprintf(stderr, "instrument: pattern not found: %s\n", what); exit(4); }
    size_t n = strlen(src), a = strlen(what), b = strlen(with);
    char *out = malloc(n - a + b + 1);
    memcpy(out, src, (size_t)(at - src));
    memcpy(out + (at - src), with, b);
    strcpy(out + (at - src) + b, at + a);
    free(src);
    return out;
}

static GLuint compile(GLenum kind, const char *src, const char *name) {
    GLuint s = p_glCreateShader(kind);
    p_glShaderSource(s, 1, &src, NULL);
    p_glCompileShader(s);
    GLint ok = 0;
    p_glGetShaderiv(s, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[4096];
        p_glGetShaderInfoLog(s, sizeof log, NULL, log);
        fprintf(stderr, "compile failed for %s:\n%s\n", name, log);
        exit(5);
    }
    return s;
}
static GLuint linkProgram(GLuint *shaders, int n) {
    GLuint prog = p_glCreateProgram();
    for (int i = 0; i < n; i++) p_glAttachShader(prog, shaders[i]);
    p_glLinkProgram(prog);
    GLint ok = 0;
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[4096];
        p_glGetProgramInfoLog(prog, sizeof log, NULL, log);
        fprintf(stderr, "link failed:\n%s\n", log);
        exit(5);
    }
    return prog;
}
static GLuint makeTex(GLenum fmt, int w, int h) {
    GLuint t;
    p_glGenTextures(1, &t);
    p_glBindTexture(GL_TEXTURE_2D, t);
    p_glTexStorage2D(GL_TEXTURE_2D, 1, fmt, w, h);
    /* sampler state of the reference's attachments: nearest + repeat (FrameBuffer.cpp:27-30) */
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_REPEAT);
    p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_REPEAT);
    return t;
}
static GLuint ssboId[16];
static size_t ssboSize[16];
static void ssbo(int binding, const char *key) {
    const char *path = jobStr(key, NULL);
    size_t n = 0;
    char *bytes = NULL;
    if (path && path[0]) {
        bytes = readFile(path, &n);
        if (!bytes) { fprintf(stderr, "cannot read %s=%s\n", key, path); exit(2); }
    }
    static const char zeros[64];
    GLuint b;
    p_glGenBuffers(1, &b);
    p_glBindBuffer(GL_SHADER_STORAGE_BUFFER, b);
    if (n) p_glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)n, bytes, GL_STATIC_DRAW);
    else p_glBufferData(GL_SHADER_STORAGE_BUFFER, sizeof zeros, zeros, GL_STATIC_DRAW);
    p_glBindBufferBase(GL_SHADER_STORAGE_BUFFER, (GLuint)binding, b);
    ssboId[binding] = b;
    ssboSize[binding] = n;
    free(bytes);
}
static int nextPowerOfTwo(int x) { /* reference src/Tracer.cpp:3-12 */
    x--;
    x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16;
    return x + 1;
}
static void checkGL(const char *where) {
    GLenum e = p_glGetError();
    if (e) { fprintf(stderr, "GL error 0x%x at %s\n", e, where); exit(6); }
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s job.txt\n", argv[0]); return 2; }
    jobLoad(argv[1]);
    int rc = glh_init(4, 3);
    if (rc) return rc;
    loadGL();
    printf("GL_VERSION %s\nGL_RENDERER %s\n", p_glGetString(GL_VERSION), p_glGetString(GL_RENDERER));

    const char *refDir = jobStr("ref_dir", "/root/reference");
    const char *out = jobStr("out", "/tmp/ref_out");
    const int W = jobInt("width", 64), H = jobInt("height", 64);
    const int spp = jobInt("spp", 1), frame0 = jobInt("frame0", 0);
    const int instrument = jobInt("instrument", 0);
    const int filterIters = jobInt("filter_iters", 0);
    const int repeat = jobInt("repeat", 1);
    const float time0 = jobFloat("time0", 1.25f), dtime = jobFloat("dtime", 0.0625f);
    const size_t npx = (size_t)W * H;
    char path[4096];

    /* ---------------- textures */
    GLuint colorTex = makeTex(GL_RGBA32F, W, H), normTex = makeTex(GL_RGBA8_SNORM, W, H),
           posTex = makeTex(GL_RGBA16F, W, H);
    GLuint tridxTex = 0, visitsTex = 0, statsTex = 0;
    const char *initColor = jobStr("init_color", NULL);
    if (initColor) {
        size_t n;
        char *b = readFile(initColor, &n);
        if (!b || n != npx * 16) { fprintf(stderr, "bad init_color\n"); return 2; }
        p_glBindTexture(GL_TEXTURE_2D, colorTex);
        p_glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, W, H, GL_RGBA, GL_FLOAT, b);
        free(b);
    }
    const char *initNorm = jobStr("init_normal", NULL), *initPos = jobStr("init_pos", NULL);
    if (initNorm) {
        size_t n;
        char *b = readFile(initNorm, &n);
        if (!b || n != npx * 4) { fprintf(stderr, "bad init_normal\n"); return 2; }
        p_glBindTexture(GL_TEXTURE_2D, normTex);
        p_glPixelStorei(GL_UNPACK_ALIGNMENT, 1);
        p_glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, W, H, GL_RGBA, GL_BYTE, b);
        free(b);
    }
    if (initPos) {
        size_t n;
        char *b = readFile(initPos, &n);
        if (!b || n != npx * 8) { fprintf(stderr, "bad init_pos\n"); return 2; }
        p_glBindTexture(GL_TEXTURE_2D, posTex);
        p_glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, W, H, GL_RGBA, GL_HALF_FLOAT, b);
        free(b);
    }

    /* ---------------- trace program (skipped when spp == 0: filter-only job) */
    if (spp > 0) {
        /* shader=trace3 is the live kernel (reference src/Tracer.cpp:14); trace / trace1 / trace2 are the
         * reference's other kernels, each linked with random.glsl for the functions they declare. */
        const char *shader = jobStr("shader", "trace3");
        const char *files3[] = {"src/shaders/trace3.cs", "src/shaders/geometry.glsl", "src/shaders/random.glsl",
                                "src/shaders/qtablehelper.glsl"};
        const char *filesBox[] = {"src/shaders/trace.cs", "src/shaders/random.glsl"};
        const char *filesBn[] = {"src/shaders/trace1.cs", "src/shaders/random.glsl"};
        const char *filesLtc[] = {"src/shaders/trace2.cs", "src/shaders/random.glsl"};
        const char **files = files3;
        int nfiles = 4, variant = 3;
        if (!strcmp(shader, "trace")) { files = filesBox; nfiles = 2; variant = 0; }
        else if (!strcmp(shader, "trace1")) { files = filesBn; nfiles = 2; variant = 1; }
        else if (!strcmp(shader, "trace2")) { files = filesLtc; nfiles = 2; variant = 2; }
        else if (!strcmp(shader, "trace3rl")) variant = 4; /* trace3.cs with main() calling traceRL (:695-791) */
        else if (strcmp(shader, "trace3")) { fprintf(stderr, "unknown shader=%s\n", shader); return 2; }
        const char *extraImages = "uniform image2D posbuffer;\n"
                                  "layout(binding = 3, r32i) uniform iimage2D o_tridx;\n"
                                  "layout(binding = 4, r32i) uniform iimage2D o_visits;\n"
                                  "layout(binding = 5, rgba32i) uniform iimage2D o_stats;\n"
                                  "int g_iter = 0; int g_rays = 0; int g_visits = 0; int g_tests = 0;\n";
        GLuint sh[4];
        for (int i = 0; i < nfiles; i++) {
            snprintf(path, sizeof path, "%s/%s", refDir, files[i]);
            char *src = readFile(path, NULL);
            if (!src) { fprintf(stderr, "reference oracle unavailable: cannot read %s\n", path); return GLH_UNAVAILABLE; }
            /* Load-time portability patches; arithmetic untouched, never written back. */
            if (i == 0 && variant == 1) {
                /* trace1.cs:83-93 declares uint8_t[] buffers (GL_NV_gpu_shader5 / EXT_shader_8bit_storage, absent on
                 * llvmpipe): same bytes read through uint[] words, little-endian byte extraction. */
                src = replaceOnce(src, NULL, "uint8_t[] sobols;", "uint[] sobols_w;");
                src = replaceOnce(src, NULL, "uint8_t[] scrambles;", "uint[] scrambles_w;");
                src = replaceOnce(src, NULL, "uint8_t[] rankings;",
                                  "uint[] rankings_w;\n};\n"
                                  "uint byteOf(uint w, uint i) { return (w >> ((i & 3u) * 8u)) & 255u; }\n"
                                  "uint sobolsAt(uint i) { return byteOf(sobols_w[i >> 2u], i); }\n"
                                  "uint scramblesAt(uint i) { return byteOf(scrambles_w[i >> 2u], i); }\n"
                                  "uint rankingsAt(uint i) { return byteOf(rankings_w[i >> 2u], i); }\n"
                                  "struct unused_tail_ { int x;");
                src = replaceOnce(src, "float sampleBlueNoise(", "rankings[sampleDimension + pxv]",
                                  "rankingsAt(sampleDimension + pxv)");
                src = replaceOnce(src, "float sampleBlueNoise(", "sobols[sampleDimension + (rankedSampleIndex << 8u)]",
                                  "sobolsAt(sampleDimension + (rankedSampleIndex << 8u))");
                src = replaceOnce(src, "float sampleBlueNoise(", "scrambles[(sampleDimension & 7u) + pxv]",
                                  "scramblesAt((sampleDimension & 7u) + pxv)");
            }
            if (i == 0 && variant == 2) {
                /* texture2D() is not in core 4.30 (trace2.cs:266-267) */
                src = replaceOnce(src, NULL, "texture2D(ltc_mat, uv)", "texture(ltc_mat, uv)");
                src = replaceOnce(src, NULL, "texture2D(ltc_mag, uv)", "texture(ltc_mag, uv)");
            }
            if (i == 2 && variant >= 3 && jobInt("weighted_lights", 0)) {
                /* randomEmitterSample (random.glsl:221-251) keeps its power / distance weighted pick in a comment
                 * block (:223-248) and returns a uniform pick; the block is enabled and its result returned */
                const char *fn = "int randomEmitterSample(";
                src = replaceOnce(src, fn, "/*float totalWeight = 0;", "float totalWeight = 0;");
                src = replaceOnce(src, fn, "}*/", "}");
                src = replaceOnce(src, fn, "//pdf = weights_[i]/totalWeight;", "pdf = weights_[i]/totalWeight; return i;");
            }
            if (i == 0 && variant == 3 && jobInt("light_pick_probe", 0)) {
                /* pick probe: the path ends at its first light pick and returns (light_prob, li, light_rand) */
                src = replaceOnce(src, "vec3 UniformSampleOneLight(", "return EstimateDirect(origin, dir, SI, bounces, light) / light_prob;",
                                  "return vec3(light_prob, float(li), light_rand);");
                src = replaceOnce(src, "vec3 trace(vec3 origin, vec3 dir, vec3 invdir)", "L += beta * Ld;", "return Ld;");
            }
            if (i == 0 && variant == 4) {
                /* the reference's main() calls trace(); its Q-table guided entry point is selected instead */
                src = replaceOnce(src, "void main(void)", "vec3 color = trace(eye, dir, 1.0 / dir);",
                                  "vec3 color = traceRL(eye, dir, 1.0 / dir);");
            }
            if (i == 0 && instrument && variant >= 3) {
                src = replaceOnce(src, NULL, "uniform image2D posbuffer;", extraImages);
                src = replaceOnce(src, "int intersectTriangles(", "const triangle tri = triangles[i];",
                                  "const triangle tri = triangles[i]; g_tests++;");
                src = replaceOnce(src, "void intersectGeometry(", "      SI.tridx = -1;",
                                  "      g_iter = iterations; g_rays++; g_visits += iterations; SI.tridx = -1;");
                src = replaceOnce(src, "void intersectGeometry(", "  SI.tridx = rettridx;",
                                  "  g_iter = iterations; g_rays++; g_visits += iterations; SI.tridx = rettridx;");
                src = replaceOnce(src, variant == 4 ? "vec3 traceRL(vec3 origin, vec3 dir, vec3 invdir)"
                                                    : "vec3 trace(vec3 origin, vec3 dir, vec3 invdir)",
                                  "bool foundIntersection = (SI.tridx != -1);",
                                  "bool foundIntersection = (SI.tridx != -1);\n"
                                  "    if(bounces == 0) { imageStore(o_tridx, px, ivec4(SI.tridx));"
                                  " imageStore(o_visits, px, ivec4(g_iter)); }\n");
                src = replaceOnce(src, "void main(void)", "imageStore(framebuffer, pix, vec4(finalColor, 1.0));",
                                  "imageStore(framebuffer, pix, vec4(finalColor, 1.0));\n"
                                  "  imageStore(o_stats, pix, ivec4(g_rays, g_visits, g_tests, floatBitsToInt(rand_seeds[-1 - (pix.x % 3)])));");
            }
            if (i == 0 && instrument && variant < 3) {
                /* primary-hit object id: box index, 100 + rectangle index, -1 for a miss; o_visits = number of
                 * bounces that found a hit is not recorded (the analytic kernels have no traversal). */
                const char *anchor = "vec3 trace(vec3 origin, vec3 dir) {";
                src = replaceOnce(src, NULL, "uniform image2D posbuffer;", extraImages);
                src = replaceOnce(src, anchor, "write(vec3(0.0), vec3(0.0));",
                                  "{ write(vec3(0.0), vec3(0.0)); imageStore(o_tridx, px, ivec4(-1)); }");
                if (variant == 0) {
                    src = replaceOnce(src, anchor, "write(point, normal);",
                                      "{ write(point, normal); imageStore(o_tridx, px, ivec4(hinfo.bi)); }");
                } else if (variant == 1) {
                    src = replaceOnce(src, anchor, "write(point, normalForRectangle(point, r));",
                                      "{ write(point, normalForRectangle(point, r)); imageStore(o_tridx, px, ivec4(100 + hinfo.bi)); }");
                    src = replaceOnce(src, anchor, "write(point, normal);",
                                      "{ write(point, normal); imageStore(o_tridx, px, ivec4(hinfo.bi)); }");
                } else {
                    src = replaceOnce(src, anchor, "write(point, normalForRectangle(point, r));",
                                      "{ write(point, normalForRectangle(point, r)); imageStore(o_tridx, px, ivec4(100 + hinfo.i)); }");
                    src = replaceOnce(src, anchor, "write(point, normal);",
                                      "{ write(point, normal); imageStore(o_tridx, px, ivec4(hinfo.i)); }");
                }
            }
            if (jobInt("dump_source", 0) && i == 0) fprintf(stderr, "%s\n", src);
            sh[i] = compile(GL_COMPUTE_SHADER, src, files[i]);
            free(src);
        }
        GLuint prog = linkProgram(sh, nfiles);
        GLint wg[3];
        p_glGetProgramiv(prog, GL_COMPUTE_WORK_GROUP_SIZE, wg);
        printf("work_group %d %d %d\n", wg[0], wg[1], wg[2]);

        if (instrument) {
            tridxTex = makeTex(GL_R32I, W, H);
            visitsTex = makeTex(GL_R32I, W, H);
            statsTex = makeTex(GL_RGBA32I, W, H);
        }
        ssbo(3, "nodes"); ssbo(4, "tris"); ssbo(5, "mtls"); ssbo(6, "emitters");
        ssbo(7, "vnormals"); ssbo(8, "vcentres"); ssbo(9, "qtable"); ssbo(10, "qvisits");
        if (variant == 1) { /* reference src/Tracer.cpp:50-90,169-171 */
            ssbo(0, "sobol"); ssbo(1, "scramble"); ssbo(2, "ranking");
        }
        if (variant == 2) { /* reference src/Tracer.cpp:23-48; units from trace2.cs:12-13 */
            const char *keys[2] = {"ltc1", "ltc2"};
            for (int k = 0; k < 2; k++) {
                size_t n;
                const char *lp = jobStr(keys[k], NULL);
                char *b = lp ? readFile(lp, &n) : NULL;
                if (!b || n != 64 * 64 * 16) { fprintf(stderr, "bad %s\n", keys[k]); return 2; }
                GLuint t;
                p_glGenTextures(1, &t);
                p_glActiveTexture(GL_TEXTURE0 + k);
                p_glBindTexture(GL_TEXTURE_2D, t);
                p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
                p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
                p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
                p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
                p_glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA32F, 64, 64);
                p_glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, 64, 64, GL_RGBA, GL_FLOAT, b);
                free(b);
            }
            p_glActiveTexture(GL_TEXTURE2);
        }

        float eye[3], r00[3], r01[3], r10[3], r11[3];
        jobVec3("eye", eye); jobVec3("ray00", r00); jobVec3("ray01", r01); jobVec3("ray10", r10); jobVec3("ray11", r11);

        /* seq=file: one dispatch per 8-float record {eye.xyz, dir.xyz, time, blendFactor}; all four corner rays are
         * set to dir, so that on a 1x1 image exactly one path runs per dispatch (a deterministic order for the
         * Q-table updates of traceRL).  The colour image after every dispatch is appended to <out>.seqcolor.f32. */
        size_t seqBytes = 0;
        float *seq = NULL;
        if (jobStr("seq", NULL)) {
            seq = (float *)readFile(jobStr("seq", NULL), &seqBytes);
            if (!seq || seqBytes % 32) { fprintf(stderr, "bad seq file\n"); return 2; }
        }
        const int nseq = (int)(seqBytes / 32);
        float *seqColor = nseq ? malloc((size_t)nseq * npx * 16) : NULL;
        const int ndispatch = nseq ? nseq : spp;
        for (int rep = 0; rep < repeat; rep++) {
            for (int k = 0; k < ndispatch; k++) {
                const int frame = frame0 + k;
                const double t0 = now();
                p_glUseProgram(prog);
                p_glUniform3fv(p_glGetUniformLocation(prog, "ray00"), 1, r00);
                p_glUniform3fv(p_glGetUniformLocation(prog, "ray01"), 1, r01);
                p_glUniform3fv(p_glGetUniformLocation(prog, "ray10"), 1, r10);
                p_glUniform3fv(p_glGetUniformLocation(prog, "ray11"), 1, r11);
                p_glUniform3fv(p_glGetUniformLocation(prog, "eye"), 1, eye);
                p_glUniform1f(p_glGetUniformLocation(prog, "time"), time0 + dtime * (float)frame);
                p_glUniform1f(p_glGetUniformLocation(prog, "blendFactor"), (float)frame / (frame + 1.0f));
                p_glUniform1i(p_glGetUniformLocation(prog, "importanceSampled"), jobInt("importance", 1));
                p_glUniform1i(p_glGetUniformLocation(prog, "numPoints"), 100);
                p_glUniform1i(p_glGetUniformLocation(prog, "numEmitters"), jobInt("num_emitters", 0));
                p_glUniform1i(p_glGetUniformLocation(prog, "maxPhi"), 10);
                p_glUniform1i(p_glGetUniformLocation(prog, "maxTheta"), 10);
                p_glUniform1i(p_glGetUniformLocation(prog, "maxBounces"), jobInt("max_bounces", 5));
                /* uniforms of the other kernels (commented-out setters, reference src/Tracer.cpp:147-153); a
                 * location of -1 is ignored by GL */
                p_glUniform1f(p_glGetUniformLocation(prog, "phongExponent"), jobFloat("phong_exponent", 128.0f));
                p_glUniform1f(p_glGetUniformLocation(prog, "specularFactor"), jobFloat("specular_factor", 0.2f));
                p_glUniform1f(p_glGetUniformLocation(prog, "roughness"), jobFloat("roughness", 0.5f));
                p_glUniform1i(p_glGetUniformLocation(prog, "multipleImportanceSampled"), jobInt("mis", 1));
                p_glUniform1i(p_glGetUniformLocation(prog, "useBlueNoise"), jobInt("use_blue_noise", 1));
                p_glUniform1i(p_glGetUniformLocation(prog, "frameIndex"), frame);
                if (nseq) {
                    const float *r = seq + 8 * k;
                    p_glUniform3fv(p_glGetUniformLocation(prog, "eye"), 1, r);
                    p_glUniform3fv(p_glGetUniformLocation(prog, "ray00"), 1, r + 3);
                    p_glUniform3fv(p_glGetUniformLocation(prog, "ray01"), 1, r + 3);
                    p_glUniform3fv(p_glGetUniformLocation(prog, "ray10"), 1, r + 3);
                    p_glUniform3fv(p_glGetUniformLocation(prog, "ray11"), 1, r + 3);
                    p_glUniform1f(p_glGetUniformLocation(prog, "time"), r[6]);
                    p_glUniform1f(p_glGetUniformLocation(prog, "blendFactor"), r[7]);
                }
                p_glBindImageTexture(0, colorTex, 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA32F);
                p_glBindImageTexture(1, normTex, 0, GL_FALSE, 0, GL_WRITE_ONLY, GL_RGBA8_SNORM);
                p_glBindImageTexture(2, posTex, 0, GL_FALSE, 0, GL_WRITE_ONLY, GL_RGBA16F);
                if (instrument) {
                    p_glBindImageTexture(3, tridxTex, 0, GL_FALSE, 0, GL_WRITE_ONLY, GL_R32I);
                    p_glBindImageTexture(4, visitsTex, 0, GL_FALSE, 0, GL_WRITE_ONLY, GL_R32I);
                    p_glBindImageTexture(5, statsTex, 0, GL_FALSE, 0, GL_WRITE_ONLY, GL_RGBA32I);
                }
                GLuint gx = (GLuint)(nextPowerOfTwo(W) / wg[0]), gy = (GLuint)(nextPowerOfTwo(H) / wg[1]);
                if (nseq) { gx = gx ? gx : 1; gy = gy ? gy : 1; } /* a 1x1 image still needs one work group */
                p_glDispatchCompute(gx, gy, 1);
                p_glMemoryBarrier(GL_SHADER_IMAGE_ACCESS_BARRIER_BIT);
                p_glMemoryBarrier(GL_SHADER_STORAGE_BARRIER_BIT);
                p_glFinish();
                if (!nseq || k < 4) printf("trace rep %d sample %d ms %.3f\n", rep, k, 1e3 * (now() - t0));
                fflush(stdout);
                if (nseq) {
                    p_glBindTexture(GL_TEXTURE_2D, colorTex);
                    p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, seqColor + (size_t)k * npx * 4);
                }
                if (instrument && rep == 0 && jobInt("dump_stats_per_sample", 0)) {
                    int *st = malloc(npx * 16);
                    p_glBindTexture(GL_TEXTURE_2D, statsTex);
                    p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA_INTEGER, GL_INT, st);
                    char suf[64];
                    snprintf(suf, sizeof suf, ".stats%d.i32", k);
                    writeFile(out, suf, st, npx * 16);
                    free(st);
                }
            }
        }
        checkGL("trace");
        if (nseq) writeFile(out, ".seqcolor.f32", seqColor, (size_t)nseq * npx * 16);
        for (int b = 9; b <= 10; b++) /* Q-table and visit counts after the run (qtablehelper.glsl:26-32) */
            if (ssboSize[b]) {
                void *buf = malloc(ssboSize[b]);
                p_glBindBuffer(GL_SHADER_STORAGE_BUFFER, ssboId[b]);
                p_glGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)ssboSize[b], buf);
                writeFile(out, b == 9 ? ".qtable.f32" : ".qvisits.i32", buf, ssboSize[b]);
                free(buf);
            }
    }

    /* ---------------- readback of the tracer outputs */
    p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    {
        void *buf = malloc(npx * 16);
        p_glBindTexture(GL_TEXTURE_2D, colorTex);
        p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, buf);
        writeFile(out, ".color.f32", buf, npx * 16);
        p_glBindTexture(GL_TEXTURE_2D, normTex);
        p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_BYTE, buf);
        writeFile(out, ".normal.i8", buf, npx * 4);
        p_glBindTexture(GL_TEXTURE_2D, posTex);
        p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_HALF_FLOAT, buf);
        writeFile(out, ".pos.f16", buf, npx * 8);
        if (instrument && spp > 0) {
            p_glBindTexture(GL_TEXTURE_2D, tridxTex);
            p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RED_INTEGER, GL_INT, buf);
            writeFile(out, ".tridx.i32", buf, npx * 4);
            p_glBindTexture(GL_TEXTURE_2D, visitsTex);
            p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RED_INTEGER, GL_INT, buf);
            writeFile(out, ".visits.i32", buf, npx * 4);
            p_glBindTexture(GL_TEXTURE_2D, statsTex);
            p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA_INTEGER, GL_INT, buf);
            writeFile(out, ".stats.i32", buf, npx * 16);
        }
        free(buf);
        checkGL("readback");
    }

    /* ---------------- à-trous filter (reference src/filter.cpp) */
    if (filterIters > 0) {
        GLuint sh[2];
        snprintf(path, sizeof path, "%s/src/shaders/atrous.vs", refDir);
        char *vs = readFile(path, NULL);
        snprintf(path, sizeof path, "%s/src/shaders/atrous.fs", refDir);
        char *fs = readFile(path, NULL);
        if (!vs || !fs) { fprintf(stderr, "reference oracle unavailable: atrous shaders\n"); return GLH_UNAVAILABLE; }
        sh[0] = compile(GL_VERTEX_SHADER, vs, "atrous.vs");
        sh[1] = compile(GL_FRAGMENT_SHADER, fs, "atrous.fs");
        GLuint prog = linkProgram(sh, 2);

        /* Quad (reference src/Quad.cpp:3-30) */
        const float verts[8] = {1, 1, 1, -1, -1, -1, -1, 1};
        const unsigned idx[6] = {0, 1, 3, 1, 2, 3};
        GLuint vao, vbo, ebo;
        p_glGenVertexArrays(1, &vao);
        p_glBindVertexArray(vao);
        p_glGenBuffers(1, &vbo);
        p_glBindBuffer(GL_ARRAY_BUFFER, vbo);
        p_glBufferData(GL_ARRAY_BUFFER, sizeof verts, verts, GL_STATIC_DRAW);
        p_glGenBuffers(1, &ebo);
        p_glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ebo);
        p_glBufferData(GL_ELEMENT_ARRAY_BUFFER, sizeof idx, idx, GL_STATIC_DRAW);
        p_glVertexAttribPointer(0, 2, GL_FLOAT, GL_FALSE, 8, (void *)0);
        p_glEnableVertexAttribArray(0);

        GLuint fbo, ftex[2];
        p_glGenFramebuffers(1, &fbo);
        p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
        for (int i = 0; i < 2; i++) {
            p_glGenTextures(1, &ftex[i]);
            p_glBindTexture(GL_TEXTURE_2D, ftex[i]);
            p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA16F, W, H, 0, GL_RGBA, GL_UNSIGNED_BYTE, NULL);
            p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
            p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
            p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + i, GL_TEXTURE_2D, ftex[i], 0);
        }
        if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) {
            fprintf(stderr, "filter FBO incomplete\n");
            return 6;
        }
        p_glViewport(0, 0, W, H);
        p_glDisable(GL_DEPTH_TEST);
        p_glDisable(GL_BLEND);

        p_glUseProgram(prog);
        p_glUniform1i(p_glGetUniformLocation(prog, "colMap"), 0);
        p_glUniform1i(p_glGetUniformLocation(prog, "normMap"), 1);
        p_glUniform1i(p_glGetUniformLocation(prog, "posMap"), 2);
        GLint loc = p_glGetUniformLocation(prog, "offset");
        for (int i = 0, y = -1; y <= 1; y++)
            for (int x = -1; x <= 1; x++, i++) p_glUniform2i(loc + i, x, y);
        const float kernel[9] = {1.0f / 16, 1.0f / 8, 1.0f / 16, 1.0f / 8, 1.0f / 4, 1.0f / 8, 1.0f / 16, 1.0f / 8, 1.0f / 16};
        p_glUniform1fv(p_glGetUniformLocation(prog, "kernel"), 9, kernel);
        const float c_phi0 = jobFloat("c_phi0", 3.3f), n_phi0 = jobFloat("n_phi0", 1E-2f), p_phi0 = jobFloat("p_phi0", 5.5f);

        void *buf = malloc(npx * 8);
        for (int rep = 0; rep < repeat; rep++) {
            GLuint read = colorTex, write = ftex[0];
            p_glActiveTexture(GL_TEXTURE1);
            p_glBindTexture(GL_TEXTURE_2D, normTex);
            p_glActiveTexture(GL_TEXTURE2);
            p_glBindTexture(GL_TEXTURE_2D, posTex);
            p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
            const double t0 = now();
            for (int i = 0; i < filterIters; i++) {
                p_glUniform1f(p_glGetUniformLocation(prog, "c_phi"), 1.0f / i * c_phi0);
                p_glUniform1f(p_glGetUniformLocation(prog, "p_phi"), 1.0f / (1 << i) * p_phi0);
                p_glUniform1f(p_glGetUniformLocation(prog, "n_phi"), 1.0f / (1 << i) * n_phi0);
                p_glUniform1i(p_glGetUniformLocation(prog, "stepwidth"), (1 << (i + 1)) - 1);
                p_glActiveTexture(GL_TEXTURE0);
                p_glBindTexture(GL_TEXTURE_2D, read);
                const GLenum draw[1] = {(GLenum)(write == ftex[0] ? GL_COLOR_ATTACHMENT0 : GL_COLOR_ATTACHMENT1)};
                p_glDrawBuffers(1, draw);
                p_glBindVertexArray(vao);
                p_glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ebo);
                p_glDrawElements(GL_TRIANGLES, 6, GL_UNSIGNED_INT, 0);
                if (rep == 0) {
                    p_glFinish();
                    p_glBindTexture(GL_TEXTURE_2D, write);
                    p_glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_HALF_FLOAT, buf);
                    char suf[64];
                    snprintf(suf, sizeof suf, ".filter%d.f16", i);
                    writeFile(out, suf, buf, npx * 8);
                }
                if (i == 0) read = ftex[1];
                GLuint tmp = read;
                read = write;
                write = tmp;
            }
            p_glFinish();
            printf("filter rep %d iters %d ms %.3f\n", rep, filterIters, 1e3 * (now() - t0));
        }
        free(buf);
        checkGL("filter");
    }
    return 0;
}
