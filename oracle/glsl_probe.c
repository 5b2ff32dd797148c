/* glsl_probe.c — TEST INFRASTRUCTURE.  Runs one of this repo's own probe compute shaders
 * (oracle/probes/NAME.comp) on llvmpipe with an input SSBO (binding 0, floats) and an output SSBO
 * (binding 1, floats), one work-group.  Used to measure the arithmetic semantics of the GLSL
 * implementation the reference is judged on (operation order of dot/mix/normalize, image
 * format conversions, ...), i.e. to re-assert SURVEY.md Appendix B before trusting a comparison.
 *   usage: glsl_probe shader.comp[,extra.glsl,...] in.f32 n_out out.f32 [groups_x]
 * Extra files (e.g. the reference's geometry.glsl / random.glsl, read from where they lie) are
 * compiled as further compute-shader objects and linked in, the way the reference links its own
 * program (reference src/ComputeShader.cpp:38-63), so a probe main can call the reference's
 * functions directly and known-answer vectors come from the reference's own code.
 */
#include <GL/gl.h>
#include <GL/glext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gl_headless.h"

static char *slurp(const char *p, size_t *n) {
    FILE *f = fopen(p, "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", p); exit(2); }
    fseek(f, 0, SEEK_END); long s = ftell(f); fseek(f, 0, SEEK_SET);
    char *b = malloc(s + 1);
    if (fread(b, 1, s, f) != (size_t)s) exit(2);
    b[s] = 0; fclose(f);
    if (n) *n = s;
    return b;
}

int main(int argc, char **argv) {
    if (argc != 5 && argc != 6) { fprintf(stderr, "usage: %s shader[,extra...] in.f32 n_out out.f32 [groups]\n", argv[0]); return 2; }
    int groups = argc == 6 ? atoi(argv[5]) : 1;
    int rc = glh_init(4, 3);
    if (rc) return rc;
    PFNGLCREATESHADERPROC CreateShader = glh_proc("glCreateShader");
    PFNGLSHADERSOURCEPROC ShaderSource = glh_proc("glShaderSource");
    PFNGLCOMPILESHADERPROC CompileShader = glh_proc("glCompileShader");
    PFNGLGETSHADERIVPROC GetShaderiv = glh_proc("glGetShaderiv");
    PFNGLGETSHADERINFOLOGPROC GetShaderInfoLog = glh_proc("glGetShaderInfoLog");
    PFNGLCREATEPROGRAMPROC CreateProgram = glh_proc("glCreateProgram");
    PFNGLATTACHSHADERPROC AttachShader = glh_proc("glAttachShader");
    PFNGLLINKPROGRAMPROC LinkProgram = glh_proc("glLinkProgram");
    PFNGLGETPROGRAMIVPROC GetProgramiv = glh_proc("glGetProgramiv");
    PFNGLGETPROGRAMINFOLOGPROC GetProgramInfoLog = glh_proc("glGetProgramInfoLog");
    PFNGLUSEPROGRAMPROC UseProgram = glh_proc("glUseProgram");
    PFNGLGENBUFFERSPROC GenBuffers = glh_proc("glGenBuffers");
    PFNGLBINDBUFFERPROC BindBuffer = glh_proc("glBindBuffer");
    PFNGLBUFFERDATAPROC BufferData = glh_proc("glBufferData");
    PFNGLBINDBUFFERBASEPROC BindBufferBase = glh_proc("glBindBufferBase");
    PFNGLGETBUFFERSUBDATAPROC GetBufferSubData = glh_proc("glGetBufferSubData");
    PFNGLDISPATCHCOMPUTEPROC DispatchCompute = glh_proc("glDispatchCompute");
    PFNGLMEMORYBARRIERPROC MemoryBarrier = glh_proc("glMemoryBarrier");
    PFNGLTEXSTORAGE2DPROC TexStorage2D = glh_proc("glTexStorage2D");
    PFNGLBINDIMAGETEXTUREPROC BindImageTexture = glh_proc("glBindImageTexture");
    void (*GenTextures)(GLsizei, GLuint *) = glh_proc("glGenTextures");
    void (*BindTexture)(GLenum, GLuint) = glh_proc("glBindTexture");
    void (*Finish)(void) = glh_proc("glFinish");

    size_t inBytes;
    char *in = slurp(argv[2], &inBytes);
    int nOut = atoi(argv[3]);
    GLuint prog = CreateProgram();
    GLint ok;
    char *list = strdup(argv[1]);
    for (char *tok = strtok(list, ","); tok; tok = strtok(NULL, ",")) {
        char *src = slurp(tok, NULL);
        GLuint sh = CreateShader(GL_COMPUTE_SHADER);
        const char *csrc = src;
        ShaderSource(sh, 1, &csrc, NULL);
        CompileShader(sh);
        GetShaderiv(sh, GL_COMPILE_STATUS, &ok);
        if (!ok) { char log[4096]; GetShaderInfoLog(sh, sizeof log, NULL, log); fprintf(stderr, "%s: %s\n", tok, log); return 5; }
        AttachShader(prog, sh);
        free(src);
    }
    LinkProgram(prog);
    GetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) { char log[4096]; GetProgramInfoLog(prog, sizeof log, NULL, log); fprintf(stderr, "%s\n", log); return 5; }
    GLuint b[2];
    GenBuffers(2, b);
    BindBuffer(GL_SHADER_STORAGE_BUFFER, b[0]);
    BufferData(GL_SHADER_STORAGE_BUFFER, inBytes ? inBytes : 4, in, GL_STATIC_DRAW);
    BindBufferBase(GL_SHADER_STORAGE_BUFFER, 0, b[0]);
    float *out = calloc(nOut, 4);
    BindBuffer(GL_SHADER_STORAGE_BUFFER, b[1]);
    BufferData(GL_SHADER_STORAGE_BUFFER, nOut * 4, out, GL_DYNAMIC_COPY);
    BindBufferBase(GL_SHADER_STORAGE_BUFFER, 1, b[1]);
    /* scratch images a probe may use for format-conversion tests: unit 0 rgba16f, unit 1 rgba8_snorm */
    GLuint tex[2];
    GenTextures(2, tex);
    BindTexture(GL_TEXTURE_2D, tex[0]);
    TexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA16F, 64, 1);
    BindTexture(GL_TEXTURE_2D, tex[1]);
    TexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA8_SNORM, 64, 1);
    BindImageTexture(0, tex[0], 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA16F);
    BindImageTexture(1, tex[1], 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA8_SNORM);
    UseProgram(prog);
    DispatchCompute(groups, 1, 1);
    MemoryBarrier(GL_ALL_BARRIER_BITS);
    Finish();
    BindBuffer(GL_SHADER_STORAGE_BUFFER, b[1]);
    GetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, nOut * 4, out);
    FILE *f = fopen(argv[4], "wb");
    fwrite(out, 4, nOut, f);
    fclose(f);
    return 0;
}
