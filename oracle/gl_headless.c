/* gl_headless.c — TEST INFRASTRUCTURE (oracle side), never linked into the product.
 *
 * Creates a headless OpenGL 4.x core context on Mesa's llvmpipe software rasteriser by
 * talking to swrast_dri.so through the DRI "swrast" loader interface
 * (/usr/include/GL/internal/dri_interface.h).  No X server, EGL, OSMesa or GLFW is needed.
 * This is how the reference's own GLSL (trace3.cs, geometry.glsl, random.glsl,
 * qtablehelper.glsl, atrous.fs — read at run time from --ref-dir, never copied here) is
 * executed to produce golden vectors.
 *
 * All GL entry points are fetched through _glapi_get_proc_address; see gl_headless.h.
 */
#include "gl_headless.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <GL/internal/dri_interface.h>

static void cbGetDrawableInfo(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) {
    (void)d; (void)p;
    *x = *y = 0;
    *w = *h = 64;
}
static void cbPutImage(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static void cbGetImage(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) {
    (void)d; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p;
}
static void cbPutImage2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p) {
    (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p;
}
static void cbGetImage2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p) {
    (void)d; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p;
}

static const __DRIswrastLoaderExtension swrastLoader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = cbGetDrawableInfo,
    .putImage = cbPutImage,
    .getImage = cbGetImage,
    .putImage2 = cbPutImage2,
    .getImage2 = cbGetImage2,
};
static const __DRIextension *loaderExtensions[] = {&swrastLoader.base, NULL};

static void *(*glapiGetProc)(const char *);

void *glh_proc(const char *name) {
    void *p = glapiGetProc ? glapiGetProc(name) : NULL;
    if (!p) {
        fprintf(stderr, "gl_headless: missing GL entry point %s\n", name);
        exit(3);
    }
    return p;
}

int glh_init(int major, int minor) {
    const char *candidates[] = {getenv("RT_SWRAST_DRI"), "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so",
                                "/usr/lib64/dri/swrast_dri.so", "/usr/lib/dri/swrast_dri.so"};
    void *drv = NULL;
    for (unsigned i = 0; i < sizeof(candidates) / sizeof(candidates[0]) && !drv; i++)
        if (candidates[i]) drv = dlopen(candidates[i], RTLD_NOW | RTLD_GLOBAL);
    if (!drv) {
        fprintf(stderr, "llvmpipe unavailable: cannot dlopen swrast_dri.so (%s)\n", dlerror());
        return GLH_UNAVAILABLE;
    }
    const __DRIextension **(*getExts)(void) =
        (const __DRIextension **(*)(void))dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!getExts) {
        fprintf(stderr, "llvmpipe unavailable: no __driDriverGetExtensions_swrast\n");
        return GLH_UNAVAILABLE;
    }
    const __DRIextension **exts = getExts();
    const __DRIcoreExtension *core = NULL;
    const __DRIswrastExtension *swrast = NULL;
    for (int i = 0; exts && exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension *)exts[i];
    }
    if (!core || !swrast || swrast->base.version < 4) {
        fprintf(stderr, "llvmpipe unavailable: DRI core/swrast extension missing\n");
        return GLH_UNAVAILABLE;
    }
    const __DRIconfig **configs = NULL;
    __DRIscreen *screen = swrast->createNewScreen2(0, loaderExtensions, exts, &configs, NULL);
    if (!screen || !configs || !configs[0]) {
        fprintf(stderr, "llvmpipe unavailable: createNewScreen2 failed\n");
        return GLH_UNAVAILABLE;
    }
    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, (uint32_t)major, __DRI_CTX_ATTRIB_MINOR_VERSION,
                          (uint32_t)minor};
    unsigned err = 0;
    __DRIcontext *ctx =
        swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!ctx) {
        fprintf(stderr, "llvmpipe unavailable: createContextAttribs failed (err %u)\n", err);
        return GLH_UNAVAILABLE;
    }
    __DRIdrawable *drawable = swrast->createNewDrawable(screen, configs[0], NULL);
    if (!drawable || !core->bindContext(ctx, drawable, drawable)) {
        fprintf(stderr, "llvmpipe unavailable: bindContext failed\n");
        return GLH_UNAVAILABLE;
    }
    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) {
        fprintf(stderr, "llvmpipe unavailable: libglapi.so.0 (%s)\n", dlerror());
        return GLH_UNAVAILABLE;
    }
    glapiGetProc = (void *(*)(const char *))dlsym(glapi, "_glapi_get_proc_address");
    if (!glapiGetProc) {
        fprintf(stderr, "llvmpipe unavailable: _glapi_get_proc_address\n");
        return GLH_UNAVAILABLE;
    }
    return 0;
}
