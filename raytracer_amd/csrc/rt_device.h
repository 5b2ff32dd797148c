// rt_device.h — device-side data layout and fp32 helpers shared by the gfx950 kernels.
//
// HBM layout (built by rt_upload from the reference's std430 streams, see rt_launcher.hip):
//   boxes   float4[2*N]   node i → {min.xyz, bits(w0)}, {max.xyz, bits(w1)}                 32 B, 32-B aligned
//           packed form (every leaf has <= 32 triangles and the triangle stream < 2^26 records, true for any tree
//           the reference's builder makes up to 67 M triangles): w0 = missNext; w1 = hitNext for an inner node,
//           0x80000000 | (numTris-1) << 26 | firstTri for a leaf (whose hitNext equals its missNext), so that one
//           32-byte read gives the box, both links and the leaf range;
//           plain form (anything else): w0 = hitNext, w1 = missNext, leaf range from `leaves`
//   leaves  int2[N]       node i → {firstTri, numTris}                                    8 B
//   tris36  float[9*T]    triangle j → v0, v1-v0, v2-v0                                    36 B, no padding (4-B aligned:
//           read with two 16-byte loads and one 4-byte load, loadTri36 below; round 2 kept 48-byte records of three
//           float4 — the vector-memory data path of a CU returns 16 dwords per clock and a dwordx3 load occupies it like a
//           dwordx4, so a test cost 48 of its clocks against 36 now, and on room.obj that path is 94 % busy: DESIGN.md 4.2)
//   triPairs float4[5*S/2] packed scenes: the same values leaf by leaf, two triangles per 80-B record, interleaved
//           component by component for packed fp32 arithmetic (RtScene::triPairs); a packed leaf word's range is in
//           SLOTS of this stream (every leaf starts at an even slot), slotTri[slot] = triangle index
//   tris128 the reference's 128-B triangle records, untouched (normals, mtlIndex, ...)   shading only
//   mtls    the reference's 32-B material records; emitters: 128-B records
//   spec    uint16[32*N] decision-tree table of the row-mode walk (small packed trees only, see RtScene::spec)
// A 32-B box record never straddles a 64-B/128-B line (the reference's 48-B records do); the
// intersection loops read 48 B per triangle instead of the first 48 B of a 128-B stride.
//
// Arithmetic rules (must match the reference shaders as executed by the oracle):
// every translation unit is compiled with -ffp-contract=off, IEEE divide/sqrt, and vector
// built-ins are spelt in the measured operation order (dot summed from the last component).
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_launcher.h"

struct RtTri128 { // reference trace3.cs:73-87
    float v0[3], w0, v1[3], w1, v2[3], w2;
    float n0[3], p0, n1[3], p1, n2[3];
    int mtlIndex;
    float uv0[2], uv1[2], uv2[2];
    int objId;
    float pad;
};
static_assert(sizeof(RtTri128) == 128, "triangle record");

struct RtMtl { // reference trace3.cs:95-100
    float Kd[3], Ns, Ke[3];
    int emitter;
};
static_assert(sizeof(RtMtl) == 32, "material record");

struct RtNode48 { // reference trace3.cs:54-65
    float min[3], pad0, max[3];
    int hitNext, missNext, firstTri, numTris;
    float pad1;
};
static_assert(sizeof(RtNode48) == 48, "node record");

struct RtScene {
    const float4 *boxes;
    const int2 *leaves;
    const float *tris36;
    const RtTri128 *tris128;
    const RtMtl *mtls;
    const RtTri128 *emitters;
    int numNodes, numTris, numMtls, numEmitters;
    // Nodes [0, cacheNodes) are read from the block's LDS copy (trace3_device.h: fillNodeCache / loadNodeT); set per
    // launch by the launcher of a kernel that fills the cache, 0 otherwise.  Nodes are numbered breadth-first
    // (reference BVH.cpp:223-241), so a prefix of the node array is exactly the top levels every ray walks.
    int cacheNodes;
    int packedLeaves; // boxes are in the packed form (see the layout above)
    // Decision-tree table of the row-mode walk (wavefront.hip), only for packed trees small enough to live in LDS
    // (numNodes <= RT_SPEC_MAX_NODES), else null: 32 uint16 per node n.  Heap position p stands for the node reached
    // from n by the decisions p spells (0 = n itself; 2p+1 = after "enter" at p, 2p+2 = after "skip" at p; a leaf's two
    // successors are the same node).  Entries 1..14: the node at position p (with n itself the 15 nodes a round tests
    // at once), 15..30: the nodes the walk can stand on after four decisions; 0xFFFF = the walk has ended there.
    // Entry 0: bit p set = position p (0..14) has a node; entry 31: bit p set = that node is a leaf.
    const uint16_t *spec;
    // Packed scenes only (trace3_device.h: testPair), read by the PAIRS variant of the extend kernel: the leaves'
    // triangles as 80-byte PAIR records {v0, e1, e2 of slots 2P and 2P+1, component by component}, every leaf starting
    // at an even slot; the triangle index of every slot (-1: the padding half of a leaf of odd size); and a second set
    // of packed node records, equal to `boxes` except that a leaf word holds the leaf's first SLOT.
    const float4 *triPairs;
    const float4 *boxesP;
    const int *slotTri;
    int numSlots;
};
#define RT_SPEC_MAX_NODES 896
#define RT_SPEC_STRIDE 32

struct RtImages {
    float4 *color;   // RGBA32F
    uint32_t *normal; // RGBA8_SNORM packed, byte 0 = x
    uint2 *pos;      // RGBA16F packed
    int *tridx, *visits;
    int4 *stats;
    int width, height;
};

// Flat cells.  The image is divided into 16 x 16 pixel cells (cell (cx, cy) = pixels 16 cx .. 16 cx + 15 of rows
// 16 cy .. 16 cy + 15).  A cell is FLAT when all its texels are equal bit for bit in colour, normal and position (the
// sky behind a scene: trace3.cs:639-647 writes one constant there).  The wavefront pipeline's blend kernel, the one
// kernel that commits a frame's colour, records which cells it left flat and their texel.  rt_filter turns the record
// into a LEVEL per cell — the number of filter passes whose result for every pixel of the cell depends on texels of
// one value only — and the results themselves (filter_kernel.hip: rt_flat_levels_kernel); a pass stores those results
// for the tiles whose cells all have them instead of staging and filtering texel by texel.  The launcher keeps track
// of whether the record still describes the images (rt_launcher.hip: flatValid).
#define RT_CELL_SHIFT 4
#define RT_FLAT_MAX_PASSES 6
struct RtFlatCells {
    uint8_t *flag;      // [cellsY][cellsX]; null: no record is kept
    float4 *color;      // the texel of a flat cell: RGBA32F colour,
    uint32_t *normal;   // packed normal,
    uint2 *pos;         // packed position
    int cellsX, cellsY;
};
// What one filter pass reads of it.  level null: the pass filters every texel.
struct RtFilterCells {
    const uint8_t *level;   // [cellsY][cellsX] passes 0 .. level-1 of this rt_filter call have a per-cell result
    const uint2 *texel;     // [cellsY][cellsX] this pass's result for the cells with level > pass
    int cellsX, pass;
};

// Global counters are sharded: every wave adds to the shard its block/wave number selects, each shard
// on a 128-byte line of its own, and rt_counters_read sums the shards.  One word shared by every wave
// of a launch serialises at the memory side (measured: ~88 atomics/us on one address, which cost the
// first version of the trace kernels milliseconds per launch).
#define RT_COUNTER_SHARDS 256
struct __attribute__((aligned(128))) RtCountersDev {
    unsigned long long rays, node_visits, tri_tests, shaded_hits, paths, primary_visits, primary_tests;
    unsigned long long rows_rays, rows_visits, rows_tests; // the part of rays / node_visits / tri_tests done by rt_wf_extend_rows_kernel
    unsigned long long pad[6];
};
static_assert(sizeof(RtCountersDev) == 128, "one shard per 128-byte line");

// ---------------------------------------------------------------------------------------
struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 F3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
// GLSL dot() as the oracle evaluates it: x*x' + (y*y' + z*z')
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float rsq(float x) { return 1.0f / sqrtf(x); }
__device__ __forceinline__ f3 normalize(f3 v) { return v * rsq(dot(v, v)); }
__device__ __forceinline__ f3 mix(f3 a, f3 b, float t) {
    return F3(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z));
}
__device__ __forceinline__ f3 reflect(f3 I, f3 N) { return I - N * (2.0f * dot(N, I)); }
__device__ __forceinline__ f3 rcp3(f3 v) { return F3(1.0f / v.x, 1.0f / v.y, 1.0f / v.z); }
__device__ __forceinline__ f3 ld3(const float *p) { return F3(p[0], p[1], p[2]); }
__device__ __forceinline__ f3 xyz(float4 v) { return F3(v.x, v.y, v.z); }

// imageStore to rgba16f truncates (measured on the oracle); v_cvt_pkrtz_f16_f32 does exactly that.
__device__ __forceinline__ uint32_t pack_half2_rtz(float a, float b) {
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    h2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float half_bits_to_float(uint32_t h16) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)h16);
}
// imageStore to rgba8_snorm: round-half-even of clamp(x,-1,1)*127; NaN stores 0.
__device__ __forceinline__ uint32_t snorm8(float x) {
    float c = fminf(fmaxf(x, -1.0f), 1.0f);
    if (x != x) c = 0.0f;
    return (uint32_t)(__float2int_rn(c * 127.0f)) & 0xFFu;
}
__device__ __forceinline__ float snorm8_to_float(uint32_t b) {
    return fmaxf((float)(int)(signed char)(b & 0xFFu) * (1.0f / 127.0f), -1.0f);
}

#endif
