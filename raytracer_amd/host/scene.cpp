// scene.cpp — OBJ/MTL loading, BVH construction + flattening, Voronoi/Q-table initialisation.
// See scene.hpp for the reference interfaces these classes mirror.
#include "scene.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>
#include <atomic>
#include <functional>
#include <thread>

using rtm::vec2;
using rtm::vec3;

// ---------------------------------------------------------------------------------------
// Triangle (reference BVH.cpp:3-51)
// ---------------------------------------------------------------------------------------
Triangle::Triangle(vec3 v1, vec3 v2, vec3 v3, vec3 n1, vec3 n2, vec3 n3, vec2 uv1, vec2 uv2, vec2 uv3,
                   int materialIndex, int objId)
    : materialIndex(materialIndex), objId(objId) {
    vertices = {v1, v2, v3};
    normals = {n1, n2, n3};
    uvs = {uv1, uv2, uv3};
}

vec3 Triangle::centroid() const { return (vertices[0] + vertices[1] + vertices[2]) * ONE_THIRD; }

vec3 Triangle::min() const {
    vec3 m;
    for (int a = 0; a < 3; ++a) m[a] = std::min(std::min(vertices[0][a], vertices[1][a]), vertices[2][a]);
    return m;
}

vec3 Triangle::max() const {
    vec3 m;
    for (int a = 0; a < 3; ++a) m[a] = std::max(std::max(vertices[0][a], vertices[1][a]), vertices[2][a]);
    return m;
}

void Triangle::calculateNormals() {
    vec3 side1 = vertices[1] - vertices[0];
    vec3 side2 = vertices[2] - vertices[1];
    vec3 n = rtm::normalize(rtm::cross(side1, side2));
    normals[0] = normals[1] = normals[2] = n;
}

// ---------------------------------------------------------------------------------------
// BVHContainer
// ---------------------------------------------------------------------------------------
namespace {

struct BuildNode {
    vec3 lo, hi;
    int left = -1, right = -1;
    uint32_t first = 0, count = 0; // leaf range into the index array
    int depth = 0;
};

struct Pending {
    uint32_t lo, hi;
    int parent;
    bool isLeft;
    int depth;
};

// Host threads for the scene-preparation loops (BVH build, lattice replication): $RT_HOST_THREADS, else
// $OMP_NUM_THREADS (what the GPU boxes set to their CPU share), else the hardware's, at most 16.
int hostThreads() {
    for (const char *name : {"RT_HOST_THREADS", "OMP_NUM_THREADS"})
        if (const char *e = std::getenv(name)) {
            const int n = std::atoi(e);
            if (n > 0) return std::min(n, 64);
        }
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hw ? hw : 1u, 16u));
}

// body(i) for i in [0, n), items handed out in chunks from a shared counter; the results must not depend on which
// thread runs an item (every use below writes to places that belong to the item alone).
void parallelFor(size_t n, size_t chunk, const std::function<void(size_t)> &body) {
    const int threads = (int)std::min<size_t>((size_t)hostThreads(), (n + chunk - 1) / std::max<size_t>(chunk, 1));
    if (threads <= 1) {
        for (size_t i = 0; i < n; ++i) body(i);
        return;
    }
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t b = next.fetch_add(chunk);
            if (b >= n) return;
            const size_t e = std::min(n, b + chunk);
            for (size_t i = b; i < e; ++i) body(i);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
}

GPUTriangle flattenTriangle(const Triangle &t) {
    // reference BVH.cpp:287-297 (and :330-340 for emitters)
    GPUTriangle g;
    for (int a = 0; a < 3; ++a) {
        g.v0[a] = t.vertices[0][a];
        g.v1[a] = t.vertices[1][a];
        g.v2[a] = t.vertices[2][a];
        g.n0[a] = t.normals[0][a];
        g.n1[a] = t.normals[1][a];
        g.n2[a] = t.normals[2][a];
    }
    g.w0 = g.w1 = g.w2 = 1.0f;
    g.p0 = g.p1 = 0.0f;
    g.mtlIndex = t.materialIndex;
    g.uv0[0] = t.uvs[0].x; g.uv0[1] = t.uvs[0].y;
    g.uv1[0] = t.uvs[1].x; g.uv1[1] = t.uvs[1].y;
    g.uv2[0] = t.uvs[2].x; g.uv2[1] = t.uvs[2].y;
    g.objId = t.objId;
    g.pad = 0.0f;
    return g;
}

} // namespace

BVHContainer::BVHContainer(std::vector<Triangle> triangles, std::vector<Material *> materials,
                           std::vector<Triangle> emitters, std::vector<std::vector<vec3>> normals_per_voronoi,
                           std::vector<std::vector<vec3>> voronoi_centres)
    : emitters(std::move(emitters)), materials(std::move(materials)) {
    build(triangles);
    finish(normals_per_voronoi, voronoi_centres);
}

BVHContainer::BVHContainer(const std::vector<Triangle> *triangles, std::vector<Material *> materials,
                           std::vector<Triangle> emitters, const std::vector<std::vector<vec3>> &normals_per_voronoi,
                           const std::vector<std::vector<vec3>> &voronoi_centres)
    : emitters(std::move(emitters)), materials(std::move(materials)) {
    build(*triangles);
    finish(normals_per_voronoi, voronoi_centres);
}

void BVHContainer::finish(const std::vector<std::vector<vec3>> &normals_per_voronoi,
                          const std::vector<std::vector<vec3>> &voronoi_centres) {
    // reference BVH.cpp:321-325
    for (Material *m : this->materials) {
        GPUMaterial g;
        g.Kd[0] = m->Kd.x; g.Kd[1] = m->Kd.y; g.Kd[2] = m->Kd.z;
        g.Ns = m->Ns;
        g.Ke[0] = m->Ke.x; g.Ke[1] = m->Ke.y; g.Ke[2] = m->Ke.z;
        g.emitter = m->emitter;
        mtlBuffer.push_back(g);
    }
    // reference BVH.cpp:327-341
    for (const Triangle &t : this->emitters) emittersBuffer.push_back(flattenTriangle(t));
    // reference BVH.cpp:343-359
    for (const auto &obj : normals_per_voronoi)
        for (const vec3 &n : obj) {
            normalsVoronoiBuffer.insert(normalsVoronoiBuffer.end(), {n.x, n.y, n.z, 0.0f});
        }
    for (const auto &obj : voronoi_centres)
        for (const vec3 &p : obj) {
            voronoiCentresBuffer.insert(voronoiCentresBuffer.end(), {p.x, p.y, p.z, 1.0f});
        }
}

// Same tree as reference buildBVH (BVH.cpp:67-155) + buildNextLinks (:157-221) +
// allocateIndexes (:223-241) + bvhToSsbo (:243-319), built breadth-first over an index array:
//  * bounds start at the centroid of the range's first triangle, then min/max over all vertices;
//  * a range with more than MAX_TRIANGLES_IN_NODE triangles is split on the strictly longest
//    axis (x, else y if y>z, else z) at the mean centroid (sum in range order, divided by the
//    count), `centroid < mean` going left, both sides keeping their input order;
//  * nodes are numbered in BFS order, left before right; leaves emit their triangles in that order;
//  * hitNext = left child (inner) or the next subtree (leaf); missNext = the next subtree.
// Deviation (the reference would recurse forever / index an empty vector): a split that leaves
// one side empty makes a leaf instead.
void BVHContainer::build(const std::vector<Triangle> &tris) {
    const uint32_t n = (uint32_t)tris.size();
    if (n == 0) return;

    std::vector<vec3> cen(n), tmin(n), tmax(n);
    parallelFor(n, 1 << 16, [&](size_t i) {
        cen[i] = tris[i].centroid();
        tmin[i] = tris[i].min();
        tmax[i] = tris[i].max();
    });
    std::vector<uint32_t> idx(n), scratch(n);
    for (uint32_t i = 0; i < n; ++i) idx[i] = i;

    // Level by level: the nodes of a level own disjoint ranges of the index array, so they are processed by
    // several threads (each node by one thread: the centroid sum of a range must be formed in range order to give
    // the reference's split plane); numbering the nodes and queueing their children stays sequential, in the order
    // a first-in-first-out queue would produce (left before right, level after level).
    std::vector<BuildNode> nodes;
    std::vector<Pending> level{{0u, n, -1, false, 0}}, nextLevel;
    struct Outcome {
        BuildNode node;
        uint32_t nl = 0;
        bool split = false;
    };
    std::vector<Outcome> outcome;
    while (!level.empty()) {
        outcome.assign(level.size(), Outcome());
        parallelFor(level.size(), 1, [&](size_t q) {
            const Pending p = level[q];
            Outcome &o = outcome[q];
            BuildNode &node = o.node;
            node.depth = p.depth;
            node.lo = node.hi = cen[idx[p.lo]];
            for (uint32_t k = p.lo; k < p.hi; ++k) {
                const uint32_t t = idx[k];
                for (int a = 0; a < 3; ++a) {
                    node.lo[a] = std::min(node.lo[a], tmin[t][a]);
                    node.hi[a] = std::max(node.hi[a], tmax[t][a]);
                }
            }
            const uint32_t count = p.hi - p.lo;
            if (count > (uint32_t)MAX_TRIANGLES_IN_NODE) {
                const vec3 len = node.hi - node.lo;
                vec3 mean(0.0f);
                for (uint32_t k = p.lo; k < p.hi; ++k) mean += cen[idx[k]];
                mean /= (float)count;
                int axis;
                if (len.x > len.y && len.x > len.z) axis = 0;
                else if (len.y > len.z) axis = 1;
                else axis = 2;
                uint32_t nl = 0, nr = 0;
                uint32_t *right = scratch.data() + p.lo; // this node's own part of the scratch array
                for (uint32_t k = p.lo; k < p.hi; ++k) {
                    const uint32_t t = idx[k];
                    if (cen[t][axis] < mean[axis]) idx[p.lo + nl++] = t; // stable: nl <= k - p.lo
                    else right[nr++] = t;
                }
                if (nl > 0 && nr > 0) {
                    std::memcpy(&idx[p.lo + nl], right, sizeof(uint32_t) * nr);
                    o.nl = nl;
                    o.split = true;
                } else if (nl == 0) {
                    std::memcpy(&idx[p.lo], right, sizeof(uint32_t) * nr);
                }
            }
            if (!o.split) {
                node.first = p.lo;
                node.count = count;
            }
        });
        nextLevel.clear();
        for (size_t q = 0; q < level.size(); ++q) {
            const Pending &p = level[q];
            const int me = (int)nodes.size();
            nodes.push_back(outcome[q].node);
            depth = std::max(depth, p.depth);
            if (p.parent >= 0) {
                if (p.isLeft) nodes[p.parent].left = me;
                else nodes[p.parent].right = me;
            }
            if (outcome[q].split) {
                nextLevel.push_back({p.lo, p.lo + outcome[q].nl, me, true, p.depth + 1});
                nextLevel.push_back({p.lo + outcome[q].nl, p.hi, me, false, p.depth + 1});
            }
        }
        level.swap(nextLevel);
    }

    // threaded links (BFS order guarantees parents are visited before children)
    std::vector<int> skip(nodes.size(), -1);
    for (size_t i = 0; i < nodes.size(); ++i) {
        if (nodes[i].left >= 0) {
            skip[nodes[i].left] = nodes[i].right;
            skip[nodes[i].right] = skip[i];
        }
    }

    // leaves emit their triangles in node order: offsets first (sequential), records in parallel
    nodesBuffer.resize(nodes.size());
    std::vector<int> firstTri(nodes.size(), 0);
    int triangleIndex = 0;
    for (size_t i = 0; i < nodes.size(); ++i)
        if (nodes[i].left < 0) {
            firstTri[i] = triangleIndex;
            triangleIndex += (int)nodes[i].count;
            maxLeafTriangles = std::max(maxLeafTriangles, (int)nodes[i].count);
        }
    trianglesBuffer.resize(n);
    triangleOrder.resize(n);
    parallelFor(nodes.size(), 4096, [&](size_t i) {
        const BuildNode &b = nodes[i];
        GPUNode g;
        for (int a = 0; a < 3; ++a) {
            g.min[a] = b.lo[a];
            g.max[a] = b.hi[a];
        }
        g.pad0 = g.pad1 = 1.0f;
        g.missNext = skip[i];
        if (b.left >= 0) {
            g.hitNext = b.left;
            g.firstTri = 0;
            g.numTris = 0;
        } else {
            g.hitNext = skip[i];
            g.firstTri = firstTri[i];
            g.numTris = (int)b.count;
            for (uint32_t k = 0; k < b.count; ++k) {
                const uint32_t t = idx[b.first + k];
                triangleOrder[(size_t)firstTri[i] + k] = t;
                trianglesBuffer[(size_t)firstTri[i] + k] = flattenTriangle(tris[t]);
            }
        }
        nodesBuffer[i] = g;
    });
}

// ---------------------------------------------------------------------------------------
// ModelLoader
// ---------------------------------------------------------------------------------------
ModelLoader::ModelLoader(const char *path) { loadObj(path); }

ModelLoader::~ModelLoader() {
    for (Material *m : materialsVector) delete m;
}

int ModelLoader::addMaterial(const Material &m, const std::string &name) {
    materialsVector.push_back(new Material(m));
    const int id = (int)materialsVector.size() - 1;
    if (!name.empty()) materialIndexes[name] = id;
    return id;
}

int ModelLoader::defaultMaterialIndex() {
    if (defaultMaterial < 0) {
        // faces without a usable material get this instead of the
        // reference's materialsVector[-1] (ModelLoader.cpp:85-92,129).
        Material m;
        m.Kd = vec3(0.8f, 0.8f, 0.8f);
        m.Ke = vec3(0.0f);
        m.Ns = 32.0f;
        m.emitter = 0;
        defaultMaterial = addMaterial(m, "");
    }
    return defaultMaterial;
}

void ModelLoader::finishTriangle(const Triangle &t) {
    triangles.push_back(t);
    if (materialsVector[t.materialIndex]->emitter) emitters.push_back(t);
    if ((int)triangles_per_obj.size() <= t.objId) triangles_per_obj.resize(t.objId + 1);
    triangles_per_obj[t.objId].push_back(t);
    if (numObjs < t.objId + 1) numObjs = t.objId + 1;
}

void ModelLoader::addTriangle(const Triangle &t) { finishTriangle(t); }

namespace {

struct FaceVertex {
    long v = 0, vt = 0, vn = 0; // 1-based, 0 = absent
};

std::string keywordOf(const std::string &line) { return line.substr(0, line.find(' ')); }

std::string restOf(const std::string &line) {
    const size_t p = line.find(' ');
    return p == std::string::npos ? std::string() : line.substr(p + 1);
}

// "v", "v/vt", "v//vn", "v/vt/vn"; negative indices count back from the end (OBJ standard).
bool parseFaceVertex(const std::string &tok, FaceVertex &fv, long nv, long nvt, long nvn) {
    long vals[3] = {0, 0, 0};
    size_t pos = 0;
    for (int k = 0; k < 3 && pos <= tok.size(); ++k) {
        size_t slash = tok.find('/', pos);
        std::string part = tok.substr(pos, slash == std::string::npos ? std::string::npos : slash - pos);
        if (!part.empty()) {
            char *end = nullptr;
            vals[k] = std::strtol(part.c_str(), &end, 10);
            if (end == part.c_str()) return false;
        }
        if (slash == std::string::npos) break;
        pos = slash + 1;
    }
    fv.v = vals[0] < 0 ? nv + 1 + vals[0] : vals[0];
    fv.vt = vals[1] < 0 ? nvt + 1 + vals[1] : vals[1];
    fv.vn = vals[2] < 0 ? nvn + 1 + vals[2] : vals[2];
    return fv.v > 0;
}

} // namespace

void ModelLoader::loadObj(const char *path) {
    std::ifstream file(path);
    if (!file) {
        std::cerr << "ERROR::FILE::COULD NOT OPEN FILE FOR MODEL LOADING" << std::endl;
        return;
    }
    std::string pathroot(path);
    {
        const size_t slash = pathroot.find_last_of('/');
        pathroot = slash == std::string::npos ? std::string(".") : pathroot.substr(0, slash);
    }

    std::vector<vec3> temp_vertices, temp_normals;
    std::vector<vec2> temp_uvs;
    struct Face {
        FaceVertex fv[3];
        int material;
        int obj;
    };
    std::vector<Face> faces;

    std::string curMtl;
    int curObj = -1;
    bool facesBeforeFirstObject = false;
    std::string line;
    while (std::getline(file, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        const std::string key = keywordOf(line);
        if (key == "o") {
            if (curObj == -1 && facesBeforeFirstObject) curObj = 0;
            curObj++;
        } else if (key == "v") {
            vec3 v;
            sscanf(restOf(line).c_str(), "%f %f %f", &v.x, &v.y, &v.z);
            temp_vertices.push_back(v);
        } else if (key == "vt") {
            vec2 uv;
            sscanf(restOf(line).c_str(), "%f %f", &uv.x, &uv.y);
            temp_uvs.push_back(uv);
        } else if (key == "vn") {
            vec3 nn;
            sscanf(restOf(line).c_str(), "%f %f %f", &nn.x, &nn.y, &nn.z);
            temp_normals.push_back(nn);
        } else if (key == "f") {
            std::istringstream ss(restOf(line));
            std::string tok;
            std::vector<FaceVertex> poly;
            bool ok = true;
            while (ss >> tok) {
                FaceVertex fv;
                if (!parseFaceVertex(tok, fv, (long)temp_vertices.size(), (long)temp_uvs.size(),
                                     (long)temp_normals.size())) {
                    ok = false;
                    break;
                }
                poly.push_back(fv);
            }
            if (!ok || poly.size() < 3) {
                std::cerr << "ERROR::FILE::UNPARSEABLE FACE: " << line << std::endl;
                continue;
            }
            int mat;
            if (!curMtl.empty() && materialIndexes.count(curMtl)) mat = materialIndexes[curMtl];
            else mat = defaultMaterialIndex();
            if (curObj < 0) facesBeforeFirstObject = true;
            const int obj = curObj < 0 ? 0 : curObj;
            // triangles as-is; larger polygons as a fan (superset of the reference, which reads 3 corners)
            for (size_t k = 1; k + 1 < poly.size(); ++k) faces.push_back({{poly[0], poly[k], poly[k + 1]}, mat, obj});
        } else if (key == "mtllib") {
            const std::string mtlpath = pathroot + "/" + restOf(line);
            loadMat(mtlpath.c_str());
        } else if (key == "usemtl") {
            curMtl = restOf(line);
        }
    }

    for (const Face &f : faces) {
        Triangle t;
        bool haveNormals = true;
        for (int k = 0; k < 3; ++k) {
            const FaceVertex &fv = f.fv[k];
            if (fv.v < 1 || fv.v > (long)temp_vertices.size()) {
                std::cerr << "ERROR::FILE::VERTEX INDEX OUT OF RANGE" << std::endl;
                t.vertices[k] = vec3(0.0f);
            } else {
                t.vertices[k] = temp_vertices[fv.v - 1];
            }
            out_vertices.push_back(t.vertices[k]);
            if (fv.vt >= 1 && fv.vt <= (long)temp_uvs.size()) t.uvs[k] = temp_uvs[fv.vt - 1];
            else t.uvs[k] = vec2(0.0f, 0.0f);
            out_uvs.push_back(t.uvs[k]);
            if (fv.vn >= 1 && fv.vn <= (long)temp_normals.size()) t.normals[k] = temp_normals[fv.vn - 1];
            else haveNormals = false;
        }
        if (!haveNormals) t.calculateNormals();
        for (int k = 0; k < 3; ++k) out_normals.push_back(t.normals[k]);
        t.materialIndex = f.material;
        t.objId = f.obj;
        finishTriangle(t);
    }
    if (curObj + 1 > numObjs) numObjs = curObj + 1;
    if ((int)triangles_per_obj.size() < numObjs) triangles_per_obj.resize(numObjs);
}

void ModelLoader::loadMat(const char *path) {
    std::ifstream file(path);
    if (!file) {
        std::cerr << "ERROR::FILE::COULD NOT OPEN FILE FOR MODEL LOADING" << std::endl;
        return;
    }
    Material *cur = nullptr;
    std::string line;
    while (std::getline(file, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        const std::string key = keywordOf(line);
        const std::string rest = restOf(line);
        if (key == "newmtl") {
            // value-initialised like the reference's `new Material()`: Kd = Ke = 0, Ns = 0
            cur = materialsVector[addMaterial(Material(), rest)];
        } else if (cur && (key == "Kd" || key == "Ke")) {
            vec3 c;
            sscanf(rest.c_str(), "%f %f %f", &c.x, &c.y, &c.z);
            if (key == "Kd") cur->Kd = c;
            else {
                cur->Ke = c;
                if (c.x > 0.0 || c.y > 0.0 || c.z > 0.0) cur->emitter = true;
            }
        } else if (cur && key == "Ns") {
            cur->Ns = std::strtof(rest.c_str(), nullptr);
        }
    }
}

void ModelLoader::bounds(vec3 &lo, vec3 &hi) const {
    // min / max are exact whatever the order: blocks of triangles are reduced by several threads, then combined
    const size_t block = 1 << 16, blocks = (triangles.size() + block - 1) / block;
    std::vector<vec3> los(blocks, vec3(1e30f)), his(blocks, vec3(-1e30f));
    parallelFor(blocks, 1, [&](size_t b) {
        vec3 l(1e30f), h(-1e30f);
        const size_t end = std::min(triangles.size(), (b + 1) * block);
        for (size_t i = b * block; i < end; ++i)
            for (int k = 0; k < 3; ++k)
                for (int a = 0; a < 3; ++a) {
                    l[a] = std::min(l[a], triangles[i].vertices[k][a]);
                    h[a] = std::max(h[a], triangles[i].vertices[k][a]);
                }
        los[b] = l;
        his[b] = h;
    });
    lo = vec3(1e30f);
    hi = vec3(-1e30f);
    for (size_t b = 0; b < blocks; ++b)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], los[b][a]);
            hi[a] = std::max(hi[a], his[b][a]);
        }
}

namespace {
struct Pcg32 { // PCG XSH-RR 64/32 (O'Neill), the generator of the pcg32 demo: srandom(seed, stream)
    uint64_t state = 0, inc = 0;
    Pcg32(uint64_t seed, uint64_t stream) {
        inc = (stream << 1u) | 1u;
        next();
        state += seed;
        next();
    }
    uint32_t next() {
        const uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        const uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
    }
    float unit() { return (float)(next() >> 8) * (1.0f / 16777216.0f); } // [0,1), 24 bits
};
} // namespace

uint32_t rt_pcg32_nth(uint64_t seed, uint64_t stream, int n) {
    Pcg32 g(seed, stream);
    uint32_t v = 0;
    for (int i = 0; i <= n; ++i) v = g.next();
    return v;
}

void ModelLoader::replicateLattice(int nx, int ny, int nz, uint64_t seed, float pitchFactor, float jitter) {
    if (nx < 1 || ny < 1 || nz < 1 || triangles.empty()) return;
    const std::vector<Triangle> base = triangles;
    vec3 lo, hi;
    bounds(lo, hi);
    const vec3 centre = (lo + hi) * 0.5f, ext = hi - lo;
    const float pitch = pitchFactor * std::max(ext.x, std::max(ext.y, ext.z));
    triangles.clear();
    emitters.clear();
    // the random numbers are drawn in copy order by one generator (cheap); the copies are then written by several threads
    struct Placement {
        float R[3][3];
        vec3 offset;
    };
    const size_t copies = (size_t)nx * ny * nz;
    std::vector<Placement> place(copies);
    Pcg32 rng(seed, 54u);
    const float TWO_PI = 6.28318530717958647692f;
    size_t c = 0;
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int k = 0; k < nz; ++k, ++c) {
                // uniformly random rotation (Shoemake's quaternion from three uniform numbers)
                const float u1 = rng.unit(), u2 = rng.unit(), u3 = rng.unit();
                const float a = std::sqrt(1.0f - u1), b = std::sqrt(u1);
                const float qx = a * std::sin(TWO_PI * u2), qy = a * std::cos(TWO_PI * u2), qz = b * std::sin(TWO_PI * u3),
                            qw = b * std::cos(TWO_PI * u3);
                const float R[3][3] = {
                    {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)},
                    {2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)},
                    {2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)}};
                std::memcpy(place[c].R, R, sizeof R);
                const float jx = rng.unit() - 0.5f, jy = rng.unit() - 0.5f, jz = rng.unit() - 0.5f;
                (void)rng.next(); // seventh draw: reserved (keeps copies on 7-draw boundaries)
                place[c].offset = vec3(pitch * ((float)i + jitter * jx), pitch * ((float)j + jitter * jy),
                                       pitch * ((float)k + jitter * jz));
            }
    triangles.resize(copies * base.size(), base[0]);
    parallelFor(copies, 64, [&](size_t ci) {
        const Placement &pl = place[ci];
        auto rot = [&](const vec3 &v) {
            return vec3(pl.R[0][0] * v.x + pl.R[0][1] * v.y + pl.R[0][2] * v.z,
                        pl.R[1][0] * v.x + pl.R[1][1] * v.y + pl.R[1][2] * v.z,
                        pl.R[2][0] * v.x + pl.R[2][1] * v.y + pl.R[2][2] * v.z);
        };
        Triangle *out = triangles.data() + ci * base.size();
        for (size_t b = 0; b < base.size(); ++b) {
            Triangle t = base[b];
            for (int v = 0; v < 3; ++v) {
                t.vertices[v] = rot(base[b].vertices[v] - centre) + centre + pl.offset;
                t.normals[v] = rot(base[b].normals[v]);
            }
            out[b] = t;
        }
    });
    bool anyEmitter = false;
    for (const Triangle &t : base) anyEmitter = anyEmitter || materialsVector[t.materialIndex]->emitter;
    if (anyEmitter)
        for (const Triangle &t : triangles)
            if (materialsVector[t.materialIndex]->emitter) emitters.push_back(t);
}

void ModelLoader::addQuadLight(vec3 c, float hx, float hz, vec3 Ke) {
    Material m;
    m.Kd = vec3(0.8f, 0.8f, 0.8f);
    m.Ns = 225.0f; // as the reference's "Light" material (res/Scene/room.mtl)
    m.Ke = Ke;
    m.emitter = (Ke.x > 0.0f || Ke.y > 0.0f || Ke.z > 0.0f);
    const int mat = addMaterial(m, "__harness_light");
    const int obj = numObjs;
    const vec3 nrm(0.0f, -1.0f, 0.0f);
    const vec3 a(c.x - hx, c.y, c.z - hz), b(c.x + hx, c.y, c.z - hz), cc(c.x + hx, c.y, c.z + hz),
        d(c.x - hx, c.y, c.z + hz);
    // wound so that cross(v1-v0, v2-v0) points down (-y): front faces look at the scene
    finishTriangle(Triangle(a, b, cc, nrm, nrm, nrm, vec2(0, 0), vec2(1, 0), vec2(1, 1), mat, obj));
    finishTriangle(Triangle(a, cc, d, nrm, nrm, nrm, vec2(0, 0), vec2(1, 1), vec2(0, 1), mat, obj));
}

// ---------------------------------------------------------------------------------------
// QTable (reference QTable.cpp)
// ---------------------------------------------------------------------------------------
namespace {

float radicalInverse(unsigned int bits) {
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    // the reference multiplies by a double literal and narrows (QTable.cpp:10)
    return (float)(float(bits) * 2.3283064365386963e-10);
}

vec2 hammersley(unsigned int i, unsigned int N) { return vec2(float(i) / float(N), radicalInverse(i)); }

vec3 uv3(const vec2 &a, const vec2 &b) { return vec3(a.x - b.x, a.y - b.y, 0.0f); }

} // namespace

QTable::QTable(int numObjs) {
    normal_per_voronoi.resize(numObjs);
    voronoi_centres.resize(numObjs);
}

// reference QTable.cpp:55-107: for each Hammersley point, the first triangle of each object
// whose UV footprint contains it gives the cell's normal and 3-D centre; none → zero vectors.
void QTable::makeVoronoi(const std::vector<std::vector<Triangle>> &triangles_per_obj) {
    for (int i = 0; i < MAX_POINTS; i++) {
        const vec2 hamm = hammersley(i, MAX_POINTS);
        for (size_t j = 0; j < triangles_per_obj.size() && j < voronoi_centres.size(); j++) {
            bool found = false;
            for (const Triangle &t : triangles_per_obj[j]) {
                const vec3 e1 = uv3(t.uvs[1], t.uvs[0]);
                const vec3 e2 = uv3(t.uvs[2], t.uvs[0]);
                const vec3 n = rtm::cross(e1, e2);
                const vec3 p1 = uv3(hamm, t.uvs[0]);
                const vec3 edge2 = uv3(t.uvs[2], t.uvs[1]);
                const vec3 p2 = uv3(hamm, t.uvs[1]);
                const vec3 edge3 = uv3(t.uvs[0], t.uvs[2]);
                const vec3 p3 = uv3(hamm, t.uvs[2]);
                const vec3 c1 = rtm::cross(e1, p1);
                const vec3 c2 = rtm::cross(edge2, p2);
                const vec3 c3 = rtm::cross(edge3, p3);
                if (rtm::dot(c1, c2) > 0 && rtm::dot(c2, c3) > 0 && rtm::dot(c3, c1) > 0) {
                    normal_per_voronoi[j].push_back(
                        rtm::normalize(rtm::cross(t.vertices[1] - t.vertices[0], t.vertices[2] - t.vertices[0])));
                    const float u = rtm::dot(n, c2) / rtm::dot(n, n);
                    const float v = rtm::dot(n, c3) / rtm::dot(n, n);
                    const vec3 pos = t.vertices[0] * (1 - u - v) + t.vertices[1] * u + t.vertices[2] * v;
                    voronoi_centres[j].push_back(pos);
                    found = true;
                    break;
                }
            }
            if (!found) {
                normal_per_voronoi[j].push_back(vec3(0.0f));
                voronoi_centres[j].push_back(vec3(0.0f));
            }
        }
    }
    fillQtableSsbo();
}

void QTable::fillQtableSsbo() {
    std::mt19937 gen(seed);
    std::uniform_real_distribution<> dist(0, 1);
    QTableBuffer.clear();
    QVisitsBuffer.clear();
    for (const auto &obj : voronoi_centres)
        for (size_t j = 0; j < obj.size(); j++)
            for (int k = 0; k < MAX_THETA * MAX_PHI; k++) QTableBuffer.push_back((float)dist(gen));
    for (const auto &obj : voronoi_centres)
        for (size_t j = 0; j < obj.size(); j++) QVisitsBuffer.push_back(0);
}
