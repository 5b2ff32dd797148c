/* gl_headless.h — TEST INFRASTRUCTURE. Headless llvmpipe GL context (see gl_headless.c). */
#ifndef RT_GL_HEADLESS_H
#define RT_GL_HEADLESS_H

#define GLH_UNAVAILABLE 77 /* exit code / return value when llvmpipe cannot be brought up */

/* Creates and binds a core-profile context of at least the given version. 0 on success. */
int glh_init(int major, int minor);
/* GL entry point by name (exits with code 3 if it does not exist). */
void *glh_proc(const char *name);

#endif
