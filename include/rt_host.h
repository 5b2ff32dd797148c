/* rt_host.h — C-ABI over the host-side scene preparation of the MI355X trace path.
 *
 * The reference does this work in C++ classes that upload straight into OpenGL buffers
 * (ModelLoader → QTable → BVHContainer, reference src/main.cpp:37-40).  This header exposes
 * the same steps to callers that cannot include the C++ classes (tests, Python, other FFIs):
 * the byte streams returned by rth_scene_buffer are exactly what the reference hands to
 * glBufferData (reference src/BVH.cpp:378-397, src/QTable.cpp:138-146) and what
 * rt_upload() in rt_launcher.h expects.
 *
 * Library: raytracer_amd/lib/librt_host.so (plain g++, no GPU dependency).
 * All functions return 0 on success and a negative value on error unless stated otherwise.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene;

/* Buffer slots = the SSBO binding numbers the reference binds in src/Tracer.cpp:173-180. */
enum rth_slot {
    RTH_SLOT_NODES = 3,           /* 48 B records, reference src/BVH.cpp:310-319            */
    RTH_SLOT_TRIANGLES = 4,       /* 128 B records, reference src/BVH.cpp:283-298           */
    RTH_SLOT_MATERIALS = 5,       /* 32 B records, reference src/BVH.cpp:321-325            */
    RTH_SLOT_EMITTERS = 6,        /* 128 B records, reference src/BVH.cpp:327-341           */
    RTH_SLOT_VORONOI_NORMALS = 7, /* vec3 padded to 16 B (w=0), reference src/BVH.cpp:343-350 */
    RTH_SLOT_VORONOI_CENTRES = 8, /* vec3 padded to 16 B (w=1), reference src/BVH.cpp:352-359 */
    RTH_SLOT_QTABLE = 9,          /* float[obj][100][10][10], reference src/QTable.cpp:114-128 */
    RTH_SLOT_QVISITS = 10         /* int[obj][100], reference src/QTable.cpp:130-136        */
};

typedef struct rth_scene_info {
    int32_t num_triangles, num_nodes, num_materials, num_emitters, num_objects;
    int32_t max_leaf_triangles, depth, reserved;
    float bbox_min[3], bbox_max[3];
} rth_scene_info;

/* ModelLoader(path) (reference src/ModelLoader.cpp:3-137) with the tolerant superset rules of
 * SURVEY.md §8(a) row L.  Returns NULL if the file cannot be opened. */
rth_scene *rth_scene_load_obj(const char *path);
/* An empty scene to be filled with rth_scene_add_triangles (synthetic scenes). */
rth_scene *rth_scene_new(void);
/* Append n triangles given as 128-byte records (layout of RTH_SLOT_TRIANGLES); mtlIndex must
 * name a material added with rth_scene_add_material (or -1 for the default material). */
int rth_scene_add_triangles(rth_scene *s, const void *records, size_t n);
/* Append a material {Kd[3], Ns, Ke[3]}; returns its index. */
int rth_scene_add_material(rth_scene *s, const float Kd[3], float Ns, const float Ke[3]);
/* Harness light for emitter-less scenes: emissive quad facing -y. */
int rth_scene_add_quad_light(rth_scene *s, const float centre[3], float half_x, float half_z, const float Ke[3]);
/* Synthetic large scenes (the reference has no instancing): replace the triangles added
 * so far by nx*ny*nz randomly rotated copies on a jittered lattice, flattened into one list.  pitch = pitch_factor
 * * largest bbox extent; copy (i,j,k) sits at pitch * ((i,j,k) + jitter * (u - 0.5)); random numbers from PCG32
 * (XSH-RR 64/32, stream 54) seeded with `seed`.  Call before adding the light and before rth_scene_build. */
int rth_scene_replicate_lattice(rth_scene *s, int nx, int ny, int nz, uint64_t seed, float pitch_factor, float jitter);
/* n-th output (0-based) of that generator, for known-answer tests. */
uint32_t rth_pcg32_nth(uint64_t seed, uint64_t stream, int n);
/* QTable(numObjs).makeVoronoi(...) then BVHContainer(...) (reference src/main.cpp:38-40).
 * with_qtable = 0 skips the Voronoi/Q-table step (slots 7-10 then hold one zero record). */
int rth_scene_build(rth_scene *s, int with_qtable, uint32_t qtable_seed);
int rth_scene_get_info(const rth_scene *s, rth_scene_info *out);
/* Borrowed pointer, valid until rth_scene_free. */
int rth_scene_buffer(const rth_scene *s, int slot, const void **bytes, size_t *nbytes);
/* triangles-buffer position → index of the triangle in loader (file) order. */
int rth_scene_triangle_order(const rth_scene *s, const uint32_t **order, size_t *n);
void rth_scene_free(rth_scene *s);

/* Tracer::updateRays (reference src/Tracer.cpp:110-130) on a Camera (reference
 * src/Camera.cpp:9-17): out = ray00, ray01, ray10, ray11 (xyz each, 12 floats). */
int rth_camera_rays(const float eye[3], const float look_at[3], const float up[3], float fov_deg, float width,
                    float height, float z_near, float z_far, float out_rays[12]);

/* Multi-GPU tiling rule of the headless host (raytracer_amd/host/tiling.hpp; no counterpart in the single-GPU
 * reference): rank's tile [x0,x1) x [y0,y1) of a width x height frame split over `world` GPUs when the filter halo is
 * `halo` pixels.  out_rect = {x0, y0, x1, y1}; out_grid (may be NULL) = {columns, rows}. */
int rth_tile_rect(int width, int height, int world, int rank, int halo, int32_t out_rect[4], int32_t out_grid[2]);

/* The state machine of the reference's input handler (reference src/glfw_setter.cpp:77-237,304-310) applied to one
 * scripted event: key = a GLFW key code (printable keys: their ASCII code) or 0 with scroll != 0 for a scroll event.
 * state = {eye[3], lookAt[3], up[3], fov, exposure} (11 floats, in/out); ints = {frameNumber, maxBounces, iterations,
 * filterOn} (in/out).  Returns 0. */
int rth_input_event(int key, double scroll, float state[11], int32_t ints[4]);

/* Mouse-look (reference src/glfw_setter.cpp:239-302): n cursor positions xy = {x0, y0, x1, y1, ...} fed in order to a
 * fresh handler on the camera state = {eye[3], lookAt[3], up[3]} (lookAt in/out).  The first position only sets the
 * reference point; every later one turns the camera by 0.05 degrees per pixel and restarts the accumulation.  Returns the
 * frame number the handler leaves (frameNumber if the camera never turned, else 0), or -1 on a bad argument. */
int rth_mouse_events(const double *xy, int n, float state[9], int32_t frameNumber);

/* FrameBuffer::saveToImage (reference src/FrameBuffer.cpp:89-107): 8-bit RGBA PNG, rows
 * flipped vertically when flip != 0. */
int rth_write_png(const char *path, int width, int height, const uint8_t *rgba, int flip);

const char *rth_version(void);

#ifdef __cplusplus
}
#endif
#endif
