// rt_comm.hip — multi-GPU assembly of a frame from per-rank screen-space tiles (include/rt_launcher.h:
// rt_comm_unique_id / rt_comm_init / rt_gather_tiles).  The reference is a single-GPU program; this is the step
// SURVEY.md 8(e) adds: scene replicated, every rank traces and filters its tile (plus halo) locally, then ONE gather.
//
// Design for xGMI (point-to-point links, no switch): the root posts one ncclRecv per peer and every peer one ncclSend
// inside a single ncclGroup, so each tile travels over its sender's own link to the root concurrently — never a ring.
// The collective and the root's pasting run on a second HIP stream; the trace stream and that stream are ordered with
// events only (no host wait), staging buffers are allocated once and double-buffered, so gather(k) overlaps
// trace(k+1).  RCCL is opened at run time (dlopen) to keep single-GPU users free of it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <type_traits>

#include "rt_context.h"

namespace {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
};

const RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // Only dlsym'd pointers are used, so the library is opened RTLD_LOCAL: its nccl* symbols must not become visible to
        // libraries loaded later (a PyTorch imported AFTER this call binds its own bundled RCCL / HIP runtime; with
        // RTLD_GLOBAL it resolved nccl* against this copy and the process aborted at exit with a double free).
        // A copy the process already holds is used as it is (RTLD_NOLOAD probe: PyTorch's, when it was imported first),
        // so that one process runs one RCCL.
        void *lib = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (lib) break;
        }
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (lib) break;
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!lib) {
            api.why = std::string("cannot open librccl.so.1: ") + dlerror();
            return;
        }
        bool all = true;
        auto sym = [&](const char *n, auto &fn) {
            void *p = dlsym(lib, n);
            if (!p) {
                all = false;
                api.why = std::string("librccl does not export ") + n;
            }
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(p);
        };
        sym("ncclGetUniqueId", api.GetUniqueId);
        sym("ncclCommInitRank", api.CommInitRank);
        sym("ncclCommDestroy", api.CommDestroy);
        sym("ncclGroupStart", api.GroupStart);
        sym("ncclGroupEnd", api.GroupEnd);
        sym("ncclSend", api.Send);
        sym("ncclRecv", api.Recv);
        sym("ncclGetErrorString", api.GetErrorString);
        api.ok = all;
    });
    return api;
}

int ncclFail(rt_ctx *c, ncclResult_t r, const char *what) {
    return fail(c, RT_EHIP, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error"));
}
#define NCCL_TRY(ctx, call)                                    \
    do {                                                       \
        ncclResult_t r_ = (call);                              \
        if (r_ != ncclSuccess) return ncclFail(ctx, r_, #call); \
    } while (0)

int ensureCommStream(rt_ctx *c) {
    rt_ctx::Comm &m = c->comm;
    if (m.ready) return RT_OK;
    HIP_TRY(c, hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(c, hipEventCreateWithFlags(&m.tileReady[k], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&m.done[k], hipEventDisableTiming));
    }
    m.ready = true;
    return RT_OK;
}

} // namespace

// called by rt_destroy / rt_resize (rt_launcher.hip)
void rtCommQuiesce(rt_ctx *c) {
    if (c->comm.ready) (void)hipStreamSynchronize(c->comm.stream);
}
void rtCommRelease(rt_ctx *c) {
    rt_ctx::Comm &m = c->comm;
    rtCommQuiesce(c);
    if (m.nccl && rccl().ok) (void)rccl().CommDestroy((ncclComm_t)m.nccl);
    m.nccl = nullptr;
    for (int k = 0; k < 2; ++k) {
        releaseBuf(m.send[k]);
        for (DevBuf &b : m.recv[k]) releaseBuf(b);
        m.recv[k].clear();
        if (m.tileReady[k]) (void)hipEventDestroy(m.tileReady[k]);
        if (m.done[k]) (void)hipEventDestroy(m.done[k]);
        m.tileReady[k] = m.done[k] = nullptr;
        m.used[k] = false;
    }
    if (m.stream) (void)hipStreamDestroy(m.stream);
    m.stream = nullptr;
    m.ready = false;
    m.nranks = 1;
    m.rank = 0;
    m.last = -1;
    releaseBuf(c->assembled);
    c->assembledPixelBytes = 0;
}
// make the context's stream wait (on the device) for the most recent gather: before RT_IMAGE_ASSEMBLED is read
int rtCommOrderAfterGather(rt_ctx *c) {
    rt_ctx::Comm &m = c->comm;
    if (m.ready && m.last >= 0) HIP_TRY(c, hipStreamWaitEvent(c->stream, m.done[m.last], 0));
    return RT_OK;
}

extern "C" {

int rt_comm_unique_id(void *id, size_t nbytes) {
    if (!id || nbytes != RT_COMM_ID_BYTES) return RT_EINVAL;
    static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "RT_COMM_ID_BYTES");
    const RcclApi &api = rccl();
    if (!api.ok) return RT_ENODEV;
    ncclUniqueId u;
    if (api.GetUniqueId(&u) != ncclSuccess) return RT_EHIP;
    std::memcpy(id, &u, sizeof u);
    return RT_OK;
}

int rt_comm_init(rt_ctx *c, int nranks, int rank, const void *id, size_t nbytes) {
    if (!c) return RT_EINVAL;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(c, RT_EINVAL, "rt_comm_init: rank / nranks out of range");
    if (nranks > 1 && !id) return fail(c, RT_EINVAL, "rt_comm_init: more than one rank needs the id of rt_comm_unique_id");
    if (id && nbytes != RT_COMM_ID_BYTES) return fail(c, RT_EINVAL, "rt_comm_init: the id is RT_COMM_ID_BYTES long");
    int rc = bind(c);
    if (rc) return rc;
    if (c->comm.nccl || c->comm.ready) rtCommRelease(c);
    if ((rc = ensureCommStream(c))) return rc;
    c->comm.nranks = nranks;
    c->comm.rank = rank;
    if (id) {
        const RcclApi &api = rccl();
        if (!api.ok) return fail(c, RT_ENODEV, "rt_comm_init: " + api.why);
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof u);
        ncclComm_t comm = nullptr;
        NCCL_TRY(c, api.CommInitRank(&comm, nranks, u, rank));
        c->comm.nccl = comm;
    }
    return RT_OK;
}

int rt_comm_destroy(rt_ctx *c) {
    if (!c) return RT_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    rtCommRelease(c);
    return RT_OK;
}

int rt_gather_tiles(rt_ctx *c, int image, const rt_rect *rects, int nrects, int root) {
    if (!c || !rects) return RT_EINVAL;
    rt_ctx::Comm &m = c->comm;
    if (nrects != m.nranks) return fail(c, RT_EINVAL, "rt_gather_tiles: one rect per rank (call rt_comm_init first)");
    if (root < 0 || root >= m.nranks) return fail(c, RT_EINVAL, "rt_gather_tiles: root out of range");
    if (image == RT_IMAGE_ASSEMBLED) return fail(c, RT_EINVAL, "rt_gather_tiles: cannot gather the assembled image");
    if (m.nranks > 1 && !m.nccl) return fail(c, RT_ESTATE, "rt_gather_tiles: no communicator (rt_comm_init)");
    const ImageRef im = imageOf(c, image);
    if (!im.p) return fail(c, RT_ESTATE, "rt_gather_tiles: image not available");
    for (int r = 0; r < nrects; ++r)
        if (!rectOk(c, rects[r])) return fail(c, RT_EINVAL, "rt_gather_tiles: rect of rank " + std::to_string(r) + " outside the image");
    int rc = bind(c);
    if (rc) return rc;
    if ((rc = ensureCommStream(c))) return rc;
    const size_t pb = im.pixelBytes;
    auto rowBytes = [&](const rt_rect &r) { return (size_t)(r.x1 - r.x0) * pb; };
    auto rows = [&](const rt_rect &r) { return (size_t)(r.y1 - r.y0); };
    auto bytesOf = [&](const rt_rect &r) { return rowBytes(r) * rows(r); };
    const int k = (int)(m.gathers & 1u);
    const rt_rect mine = rects[m.rank];
    const bool isRoot = m.rank == root;

    // staging memory: grows only when a tile is larger than any seen before (normally the first two calls)
    if ((rc = ensure(c, m.send[k], bytesOf(mine)))) return rc;
    if (isRoot) {
        if ((int)m.recv[k].size() != m.nranks) m.recv[k].resize((size_t)m.nranks);
        for (int r = 0; r < m.nranks; ++r)
            if (r != root && (rc = ensure(c, m.recv[k][r], bytesOf(rects[r])))) return rc;
        const size_t need = (size_t)c->width * c->height * pb;
        if (!c->assembled.p || c->assembled.bytes < need || c->assembledPixelBytes != pb) {
            rtCommQuiesce(c);
            if ((rc = ensure(c, c->assembled, need))) return rc;
            c->assembledPixelBytes = pb;
            HIP_TRY(c, hipMemsetAsync(c->assembled.p, 0, need, m.stream));
        }
    }

    // 1. trace stream: once gather k-2 has left this staging buffer, pack the tile (rows of the image -> dense rows)
    // (the pack reads an image the next frame's commit overwrites, unless it is a lane's own filtered image: that commit
    // waits for everything queued here, not just for the filter's first pass — rt_filter)
    if (!(image == RT_IMAGE_FILTERED && c->filteredLane >= 0)) c->earlyFence = nullptr;
    if (m.used[k]) HIP_TRY(c, hipStreamWaitEvent(c->stream, m.done[k], 0));
    if (bytesOf(mine)) {
        const char *imgAt = (const char *)im.p + ((size_t)mine.y0 * c->width + mine.x0) * pb;
        HIP_TRY(c, hipMemcpy2DAsync(m.send[k].p, rowBytes(mine), imgAt, (size_t)c->width * pb, rowBytes(mine), rows(mine),
                                    hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(c, hipEventRecord(m.tileReady[k], c->stream));

    // 2. gather stream: sends / receives in one group, then (root) the pasting
    HIP_TRY(c, hipStreamWaitEvent(m.stream, m.tileReady[k], 0));
    if (m.nranks > 1) {
        const RcclApi &api = rccl();
        NCCL_TRY(c, api.GroupStart());
        ncclResult_t first = ncclSuccess;
        if (isRoot) {
            for (int r = 0; r < m.nranks && first == ncclSuccess; ++r)
                if (r != root && bytesOf(rects[r]))
                    first = api.Recv(m.recv[k][r].p, bytesOf(rects[r]), ncclChar, r, (ncclComm_t)m.nccl, m.stream);
        } else if (bytesOf(mine)) {
            first = api.Send(m.send[k].p, bytesOf(mine), ncclChar, root, (ncclComm_t)m.nccl, m.stream);
        }
        const ncclResult_t end = api.GroupEnd(); // always closed, also after a failed post
        if (first != ncclSuccess) return ncclFail(c, first, "ncclSend / ncclRecv");
        if (end != ncclSuccess) return ncclFail(c, end, "ncclGroupEnd");
    }
    if (isRoot) {
        for (int r = 0; r < m.nranks; ++r) {
            const rt_rect &t = rects[r];
            if (!bytesOf(t)) continue;
            const void *src = r == root ? m.send[k].p : m.recv[k][r].p;
            char *dst = (char *)c->assembled.p + ((size_t)t.y0 * c->width + t.x0) * pb;
            HIP_TRY(c, hipMemcpy2DAsync(dst, (size_t)c->width * pb, src, rowBytes(t), rowBytes(t), rows(t),
                                        hipMemcpyDeviceToDevice, m.stream));
        }
    }
    HIP_TRY(c, hipEventRecord(m.done[k], m.stream));
    m.used[k] = true;
    m.last = k;
    m.gathers++;
    return RT_OK;
}

} // extern "C"
