// scene.hpp — host-side scene preparation for the MI355X trace path.
//
// Mirrors the public surface of the reference's scene classes so that code written against
// them keeps compiling (class / member / method names and argument meaning):
//   Material       reference include/Material.hpp:9-15
//   Triangle       reference include/BVH.hpp:23-55,   src/BVH.cpp:3-51
//   BVHContainer   reference include/BVH.hpp:76-143,  src/BVH.cpp:58-399
//   ModelLoader    reference include/ModelLoader.hpp:25-49, src/ModelLoader.cpp:3-198
//   QTable         reference include/QTable.hpp:29-74, src/QTable.cpp:3-147
// The bodies are new.  Where the reference uploads with glBufferData, these classes keep the
// std430 byte streams in memory (and hand them to the rt_* launcher, see render.hpp); where
// the reference recurses over copied std::vector<Triangle>s, the builder here is iterative
// over an index array so that tens of millions of triangles are practical.  For any scene
// the reference can build, the node / triangle / material / emitter byte streams are
// intended to be identical (same splits, same BFS numbering, same hit/miss threading).
#ifndef RT_AMD_SCENE_HPP
#define RT_AMD_SCENE_HPP

#include <array>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "rtmath.hpp"

#define MAX_POINTS 100
#define MAX_THETA 10
#define MAX_PHI 10

class Material {
public:
    rtm::vec3 Kd, Ke;
    float Ns = 0.0f;
    int emitter = false;
};

class Triangle {
    static constexpr float ONE_THIRD = 1.0f / 3.0f;

public:
    std::array<rtm::vec3, 3> vertices;
    std::array<rtm::vec3, 3> normals;
    std::array<rtm::vec2, 3> uvs;
    int materialIndex = 0;
    int objId = 0;

    Triangle() {}
    Triangle(rtm::vec3 v1, rtm::vec3 v2, rtm::vec3 v3, rtm::vec3 n1, rtm::vec3 n2, rtm::vec3 n3, rtm::vec2 uv1,
             rtm::vec2 uv2, rtm::vec2 uv3, int materialIndex, int objId);
    rtm::vec3 centroid() const;
    rtm::vec3 max() const;
    rtm::vec3 min() const;
    void calculateNormals();
};

// One flattened node exactly as the trace kernels read it (std430 `struct node`, 48 bytes).
struct GPUNode {
    float min[3];
    float pad0;
    float max[3];
    int hitNext, missNext, firstTri, numTris;
    float pad1;
};
static_assert(sizeof(GPUNode) == 48, "node record is 48 bytes");

// One flattened triangle (std430 `struct triangle`, 128 bytes).
struct GPUTriangle {
    float v0[3], w0, v1[3], w1, v2[3], w2;
    float n0[3], p0, n1[3], p1, n2[3];
    int mtlIndex;
    float uv0[2], uv1[2], uv2[2];
    int objId;
    float pad;
};
static_assert(sizeof(GPUTriangle) == 128, "triangle record is 128 bytes");

struct GPUMaterial {
    float Kd[3], Ns, Ke[3];
    int emitter;
};
static_assert(sizeof(GPUMaterial) == 32, "material record is 32 bytes");

class BVHContainer {
public:
    static constexpr int MAX_TRIANGLES_IN_NODE = 32;

    BVHContainer(std::vector<Triangle> triangles, std::vector<Material *> materials, std::vector<Triangle> emitters,
                 std::vector<std::vector<rtm::vec3>> normals_per_voronoi,
                 std::vector<std::vector<rtm::vec3>> voronoi_centres);
    // Not in the reference: the same construction without copying the triangle list (tens of millions of
    // triangles in the synthetic scenes); `triangles` is only read during the call.
    BVHContainer(const std::vector<Triangle> *triangles, std::vector<Material *> materials,
                 std::vector<Triangle> emitters, const std::vector<std::vector<rtm::vec3>> &normals_per_voronoi,
                 const std::vector<std::vector<rtm::vec3>> &voronoi_centres);

    // SSBO "names" of the reference become slots of the launcher (binding numbers of
    // reference Tracer.cpp:173-178); the byte streams themselves are kept here.
    unsigned nodesSsbo = 3, trianglesSsbo = 4, materialsSsbo = 5, emittersSsbo = 6, normalsVoronoiSsbo = 7,
             voronoiCentresSsbo = 8;
    std::vector<Triangle> emitters;

    std::vector<GPUNode> nodesBuffer;
    std::vector<GPUTriangle> trianglesBuffer;
    std::vector<GPUMaterial> mtlBuffer;
    std::vector<GPUTriangle> emittersBuffer;
    std::vector<float> normalsVoronoiBuffer, voronoiCentresBuffer;
    // trianglesBuffer[i] is input triangle triangleOrder[i].
    std::vector<uint32_t> triangleOrder;
    int maxLeafTriangles = 0;
    int depth = 0;

private:
    void build(const std::vector<Triangle> &tris);
    void finish(const std::vector<std::vector<rtm::vec3>> &normals_per_voronoi,
                const std::vector<std::vector<rtm::vec3>> &voronoi_centres);
    std::vector<Material *> materials;
};

class ModelLoader {
public:
    std::vector<rtm::vec3> out_vertices;
    std::vector<rtm::vec2> out_uvs;
    std::vector<rtm::vec3> out_normals;
    std::vector<Triangle> triangles;
    std::vector<std::vector<Triangle>> triangles_per_obj;
    std::vector<Triangle> emitters;
    std::vector<Material *> materialsVector;
    int numObjs = 0;

    ModelLoader() {}
    ModelLoader(const char *path);
    ~ModelLoader();
    ModelLoader(const ModelLoader &) = delete;
    ModelLoader &operator=(const ModelLoader &) = delete;
    void loadObj(const char *path);
    void loadMat(const char *path);

    // Not in the reference: harness light for emitter-less scenes:
    // a 2-triangle emissive quad facing -y, appended as a new object with its own material.
    void addQuadLight(rtm::vec3 centre, float halfX, float halfZ, rtm::vec3 Ke);
    // Not in the reference: append an already-built triangle (synthetic scenes).
    void addTriangle(const Triangle &t);
    int addMaterial(const Material &m, const std::string &name);
    // Not in the reference (which has no instancing): replace the triangles loaded so far by nx*ny*nz rotated,
    // jittered copies of them on a lattice, flattened into one triangle list.
    // Copy (i,j,k) is rotated about the mesh centre by a uniformly random rotation and moved to
    // pitch * (i,j,k) + pitch * jitter * (u - 0.5), pitch = pitchFactor * largest bbox extent; the random numbers
    // come from PCG32 (XSH-RR 64/32, stream 54) seeded with `seed`, seven draws per copy in i-major order.
    // Per-object triangle lists (only used by the Q-table) are not kept for the copies.
    void replicateLattice(int nx, int ny, int nz, uint64_t seed, float pitchFactor, float jitter);
    void bounds(rtm::vec3 &lo, rtm::vec3 &hi) const;
    int defaultMaterialIndex();

private:
    std::map<std::string, int> materialIndexes;
    int defaultMaterial = -1;
    void finishTriangle(const Triangle &t);
};

class QTable {
public:
    std::vector<std::vector<rtm::vec3>> normal_per_voronoi;
    std::vector<std::vector<rtm::vec3>> voronoi_centres;

    QTable(int numObjs);
    void makeVoronoi(const std::vector<std::vector<Triangle>> &triangles_per_obj);
    void fillQtableSsbo();

    unsigned QTableSsbo = 9, QVisitsSsbo = 10;
    std::vector<float> QTableBuffer;
    std::vector<int> QVisitsBuffer;
    // The reference seeds std::mt19937 from std::random_device (QTable.cpp:111-112), so its
    // initial Q values are not reproducible; here the seed is explicit.
    uint32_t seed = 0x5EEDu;
};

#endif
