// Multi-GPU inside the library (SURVEY.md 8e, 8b "required extension"): ray batches shard embarrassingly, the ONE exchange of
// the path is a root-to-all broadcast of the scene — the Bvh::serialize byte stream (reference bvh.h:221-229) written into HBM
// straight from the resident nodes (wire.hip) + the BVH-ordered primitive array — over RCCL (xGMI). No other collective exists.
//
//   one process per GPU : bvh_amd_comm_unique_id / bvh_amd_comm_create (ncclCommInitRank) + bvhXX_broadcast   — what
//                         bvh_amd/parallel.py and bench.py --gpus N drive (torch.distributed only carries the 128-byte id);
//   one process, N GPUs : bvhXX_replicate (ncclCommInitAll + one grouped ncclBroadcast) — what a C / C++ user of the reference
//                         API (test/c_api_example.c, test/benchmark.cpp) calls before tracing a ray shard per device.
//
// No payload byte visits the host on any rank: the receivers turn the received device buffer into resident nodes + traversal
// records with deserialize_from_device (device-to-device copies, id narrowing, structural validation, relayout kernels).
#include "common.h"

#include <rccl/rccl.h>

#include <cstdlib>
#include <memory>
#include <vector>

struct bvh_amd_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1, device = -1;
    bool owned = true;
};

namespace bvh_amd {

namespace {

#define BVH_NCCL_TRY(expr)                                                                          \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess)                                                                      \
            return ::bvh_amd::fail(BVH_AMD_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)

struct DeviceGuard {                                           // the calling thread's device is restored on every path out
    int saved = -1;
    DeviceGuard() { (void)hipGetDevice(&saved); }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

struct SceneMeta {                                             // what the receivers need before they can post their receives
    unsigned long long stream_bytes, prim_bytes, dim, is_double;
};

// Non-root side of the exchange: the received stream -> a BvhImpl on the current device. The stream buffer is released here.
template <typename T>
BvhImpl<T>* adopt_stream(void* d_stream, size_t bytes, int dim, hipStream_t stream) {
    BvhImpl<T>* b = deserialize_from_device<T>(d_stream, bytes, dim, stream);
    (void)hipFree(d_stream);
    return b;
}

template <typename T>
BvhImpl<T>* broadcast_scene(bvh_amd_comm* c, int root, BvhImpl<T>* bvh, int dim, const void* d_prims, size_t prim_bytes, void** d_prims_out,
                            size_t* prim_bytes_out, hipStream_t stream, const std::string* root_error = nullptr)
{
    if (!c || !c->comm) { set_error("broadcast: null communicator"); return nullptr; }
    if (root < 0 || root >= c->size) { set_error("broadcast: root rank out of range"); return nullptr; }
    if (!d_prims_out) { set_error("broadcast: d_prims_out is required"); return nullptr; }
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) { set_error("broadcast: the communicator belongs to another device than the current one"); return nullptr; }
    const bool is_root = c->rank == root;
    // a test knob: the root ALSO runs the receiving side on the broadcast buffers and returns that copy (exercises the whole
    // non-root path on a single GPU, where RCCL refuses two ranks on one device)
    static const bool loopback = getenv("BVH_AMD_BROADCAST_LOOPBACK") && atoi(getenv("BVH_AMD_BROADCAST_LOOPBACK")) != 0;
    // every check that can fail on the root alone comes BEFORE the first collective: the other ranks must not be left waiting
    SceneMeta meta = {0, 0, 0, 0};
    if (is_root) {
        if (root_error) { set_error(*root_error); meta.dim = 0; }
        else if (!bvh) { set_error("broadcast: the root rank must pass its BVH"); meta.dim = 0; }
        else if (bvh->dim != dim) { set_error("broadcast: BVH dimension does not match the entry point"); meta.dim = 0; }
        else if (bvh->device != cur) { set_error("broadcast: the root's BVH lives on another device than the current one"); meta.dim = 0; }
        else if (prim_bytes && !d_prims) { set_error("broadcast: null primitive array"); meta.dim = 0; }
        else {
            meta.stream_bytes = wire_size<T>(*bvh); meta.prim_bytes = prim_bytes; meta.dim = static_cast<unsigned long long>(dim);
            meta.is_double = sizeof(T) == 8;
        }
    }
    SceneMeta* d_meta = nullptr;
    BVH_HIP_TRY_PTR(hipMalloc(&d_meta, sizeof(SceneMeta)));
    auto fail_ptr = [&](const std::string& msg) -> BvhImpl<T>* { set_error(msg); (void)hipFree(d_meta); return nullptr; };
    if (is_root && hipMemcpyAsync(d_meta, &meta, sizeof(meta), hipMemcpyHostToDevice, stream) != hipSuccess) return fail_ptr("broadcast: copying the header failed");
    ncclResult_t r = ncclBroadcast(d_meta, d_meta, sizeof(SceneMeta), ncclUint8, root, c->comm, stream);
    if (r != ncclSuccess) return fail_ptr(std::string("broadcast: ncclBroadcast(header): ") + ncclGetErrorString(r));
    if (hipMemcpyAsync(&meta, d_meta, sizeof(meta), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
        return fail_ptr("broadcast: reading the header failed");
    (void)hipFree(d_meta);
    d_meta = nullptr;
    if (meta.dim == 0) {                                       // the root refused (its message is in ITS last_error)
        if (!is_root) set_error("broadcast: the root rank had nothing valid to send");
        return nullptr;
    }
    if (meta.dim != static_cast<unsigned long long>(dim) || meta.is_double != (sizeof(T) == 8 ? 1ull : 0ull)) {
        set_error("broadcast: the root sends another BVH family (scalar type / dimension) than this entry point receives");
        return nullptr;                                        // (every rank of a correct program calls the same entry point)
    }
    // payload: the serialized BVH, then the primitives; both device buffers, grouped into one RCCL launch
    void* d_stream = nullptr;
    void* d_recv_prims = nullptr;
    BVH_HIP_TRY_PTR(hipMalloc(&d_stream, meta.stream_bytes));
    auto drop = [&]() { if (d_stream) (void)hipFree(d_stream); if (d_recv_prims) (void)hipFree(d_recv_prims); };
    if (is_root) {
        // (serialize_to_device wants resident nodes: the C wrappers below call nodes_resident first)
        if (serialize_to_device<T>(*bvh, d_stream, meta.stream_bytes, stream) != meta.stream_bytes) { drop(); return nullptr; }
    }
    const bool receives = !is_root || loopback;
    if (receives && meta.prim_bytes) {
        if (hipMalloc(&d_recv_prims, meta.prim_bytes) != hipSuccess) { drop(); set_error("broadcast: out of device memory for the primitives"); return nullptr; }
    }
    const void* send_prims = is_root ? d_prims : d_recv_prims;
    void* recv_prims = receives ? d_recv_prims : const_cast<void*>(d_prims);     // in place on a root that keeps its own copy
    r = ncclGroupStart();
    if (r == ncclSuccess) r = ncclBroadcast(d_stream, d_stream, meta.stream_bytes, ncclUint8, root, c->comm, stream);
    if (r == ncclSuccess && meta.prim_bytes) r = ncclBroadcast(send_prims, recv_prims, meta.prim_bytes, ncclUint8, root, c->comm, stream);
    ncclResult_t r2 = ncclGroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) { drop(); set_error(std::string("broadcast: ncclBroadcast(payload): ") + ncclGetErrorString(r)); return nullptr; }
    if (prim_bytes_out) *prim_bytes_out = meta.prim_bytes;
    if (!receives) {
        if (hipStreamSynchronize(stream) != hipSuccess) { drop(); set_error("broadcast: stream synchronisation failed"); return nullptr; }
        (void)hipFree(d_stream);
        *d_prims_out = const_cast<void*>(d_prims);
        return bvh;
    }
    BvhImpl<T>* out = adopt_stream<T>(d_stream, meta.stream_bytes, dim, stream);      // synchronises the stream, frees d_stream
    d_stream = nullptr;
    if (!out) { drop(); return nullptr; }
    *d_prims_out = d_recv_prims;
    return out;
}

// One process, N devices. `devices[i]` (NULL: 0..n-1) receives a copy; the entry of the BVH's own device gets the original.
template <typename T>
int replicate_scene(BvhImpl<T>* bvh, int dim, const void* d_prims, size_t prim_bytes, int n_devices, const int* devices, BvhImpl<T>** bvhs_out,
                    void** d_prims_out)
{
    if (!bvh || !bvhs_out || !d_prims_out || n_devices < 1) return fail(BVH_AMD_ERR_ARG, "replicate: bad argument");
    if (bvh->dim != dim) return fail(BVH_AMD_ERR_ARG, "replicate: BVH dimension does not match the entry point");
    if (prim_bytes && !d_prims) return fail(BVH_AMD_ERR_ARG, "replicate: null primitive array");
    int have = 0;
    BVH_HIP_TRY(hipGetDeviceCount(&have), BVH_AMD_ERR_HIP);
    std::vector<int> devs(static_cast<size_t>(n_devices));
    int root = -1;
    for (int i = 0; i < n_devices; ++i) {
        devs[i] = devices ? devices[i] : i;
        if (devs[i] < 0 || devs[i] >= have) return fail(BVH_AMD_ERR_ARG, "replicate: no such device");
        for (int j = 0; j < i; ++j) if (devs[j] == devs[i]) return fail(BVH_AMD_ERR_ARG, "replicate: a device is listed twice");
        if (devs[i] == bvh->device) root = i;
    }
    if (root < 0) return fail(BVH_AMD_ERR_ARG, "replicate: the BVH's own device must be one of the devices");
    for (int i = 0; i < n_devices; ++i) { bvhs_out[i] = nullptr; d_prims_out[i] = nullptr; }
    bvhs_out[root] = bvh; d_prims_out[root] = const_cast<void*>(d_prims);
    if (n_devices == 1) return BVH_AMD_OK;
    DeviceGuard guard;
    const size_t stream_bytes = wire_size<T>(*bvh);
    std::vector<ncclComm_t> comms(static_cast<size_t>(n_devices), nullptr);
    std::vector<hipStream_t> streams(static_cast<size_t>(n_devices), nullptr);
    std::vector<void*> bufs(static_cast<size_t>(n_devices), nullptr);
    int rc = BVH_AMD_OK;
    auto cleanup = [&](int code) {
        for (int i = 0; i < n_devices; ++i) {
            (void)hipSetDevice(devs[i]);
            if (streams[i]) { (void)hipStreamSynchronize(streams[i]); (void)hipStreamDestroy(streams[i]); }
            if (bufs[i]) (void)hipFree(bufs[i]);
            if (comms[i]) (void)ncclCommDestroy(comms[i]);
            if (code != BVH_AMD_OK && i != root) {
                if (d_prims_out[i]) { (void)hipFree(d_prims_out[i]); d_prims_out[i] = nullptr; }
                if (bvhs_out[i]) { delete bvhs_out[i]; bvhs_out[i] = nullptr; }
            }
        }
        return code;
    };
    BVH_NCCL_TRY(ncclCommInitAll(comms.data(), n_devices, devs.data()));
    for (int i = 0; i < n_devices && rc == BVH_AMD_OK; ++i) {
        hipError_t e = hipSetDevice(devs[i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(&bufs[i], stream_bytes);
        if (e == hipSuccess && i != root && prim_bytes) e = hipMalloc(&d_prims_out[i], prim_bytes);
        if (e != hipSuccess) rc = fail(BVH_AMD_ERR_HIP, std::string("replicate: ") + hipGetErrorString(e));
    }
    if (rc) return cleanup(rc);
    (void)hipSetDevice(devs[root]);
    if (serialize_to_device<T>(*bvh, bufs[root], stream_bytes, streams[root]) != stream_bytes) return cleanup(BVH_AMD_ERR_HIP);
    ncclResult_t r = ncclGroupStart();
    for (int i = 0; i < n_devices && r == ncclSuccess; ++i) {
        r = ncclBroadcast(bufs[root], bufs[i], stream_bytes, ncclUint8, root, comms[i], streams[i]);
        if (r == ncclSuccess && prim_bytes) r = ncclBroadcast(d_prims, d_prims_out[i], prim_bytes, ncclUint8, root, comms[i], streams[i]);
    }
    ncclResult_t r2 = ncclGroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return cleanup(fail(BVH_AMD_ERR_HIP, std::string("replicate: ncclBroadcast: ") + ncclGetErrorString(r)));
    for (int i = 0; i < n_devices; ++i) {
        if (i == root) continue;
        if (hipSetDevice(devs[i]) != hipSuccess) return cleanup(fail(BVH_AMD_ERR_HIP, "replicate: hipSetDevice"));
        bvhs_out[i] = deserialize_from_device<T>(bufs[i], stream_bytes, dim, streams[i]);   // synchronises streams[i]
        if (!bvhs_out[i]) return cleanup(BVH_AMD_ERR_HIP);
    }
    return cleanup(BVH_AMD_OK);
}

} // namespace

} // namespace bvh_amd

using namespace bvh_amd;

extern "C" {

BVH_AMD_API int bvh_amd_device_select(int device) {
    BVH_HIP_TRY(hipSetDevice(device), BVH_AMD_ERR_HIP);
    return BVH_AMD_OK;
}

BVH_AMD_API int bvh_amd_device_current(void) {
    int d = -1;
    BVH_HIP_TRY(hipGetDevice(&d), BVH_AMD_ERR_HIP);
    return d;
}

BVH_AMD_API int bvh_amd_comm_unique_id(void* id_out) {
    if (!id_out) return fail(BVH_AMD_ERR_ARG, "comm_unique_id: null output");
    static_assert(sizeof(ncclUniqueId) == BVH_AMD_COMM_ID_BYTES);
    ncclUniqueId id;
    BVH_NCCL_TRY(ncclGetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return BVH_AMD_OK;
}

BVH_AMD_API struct bvh_amd_comm* bvh_amd_comm_create(const void* id_bytes, int n_ranks, int rank) {
    if (!id_bytes || n_ranks < 1 || rank < 0 || rank >= n_ranks) { set_error("comm_create: bad argument"); return nullptr; }
    auto c = std::make_unique<bvh_amd_comm>();
    if (hipGetDevice(&c->device) != hipSuccess) { set_error("comm_create: no current device"); return nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&c->comm, n_ranks, id, rank);
    if (r != ncclSuccess) { set_error(std::string("comm_create: ncclCommInitRank: ") + ncclGetErrorString(r)); return nullptr; }
    c->rank = rank; c->size = n_ranks;
    return c.release();
}

BVH_AMD_API struct bvh_amd_comm* bvh_amd_comm_adopt(void* nccl_comm) {
    if (!nccl_comm) { set_error("comm_adopt: null ncclComm_t"); return nullptr; }
    auto c = std::make_unique<bvh_amd_comm>();
    c->comm = static_cast<ncclComm_t>(nccl_comm);
    c->owned = false;
    ncclResult_t r = ncclCommUserRank(c->comm, &c->rank);
    if (r == ncclSuccess) r = ncclCommCount(c->comm, &c->size);
    if (r == ncclSuccess) r = ncclCommCuDevice(c->comm, &c->device);
    if (r != ncclSuccess) { set_error(std::string("comm_adopt: ") + ncclGetErrorString(r)); return nullptr; }
    return c.release();
}

BVH_AMD_API void bvh_amd_comm_destroy(struct bvh_amd_comm* c) {
    if (!c) return;
    if (c->owned && c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
}

BVH_AMD_API int bvh_amd_comm_rank(const struct bvh_amd_comm* c) { return c ? c->rank : fail(BVH_AMD_ERR_ARG, "comm_rank: null communicator"); }
BVH_AMD_API int bvh_amd_comm_size(const struct bvh_amd_comm* c) { return c ? c->size : fail(BVH_AMD_ERR_ARG, "comm_size: null communicator"); }
BVH_AMD_API void* bvh_amd_comm_handle(const struct bvh_amd_comm* c) { return c ? c->comm : nullptr; }

BVH_AMD_API int bvh_amd_comm_broadcast(struct bvh_amd_comm* c, void* d_buf, size_t bytes, int root, void* stream) {
    if (!c || !c->comm || (bytes && !d_buf)) return fail(BVH_AMD_ERR_ARG, "comm_broadcast: bad argument");
    if (bytes == 0) return BVH_AMD_OK;
    BVH_NCCL_TRY(ncclBroadcast(d_buf, d_buf, bytes, ncclUint8, root, c->comm, static_cast<hipStream_t>(stream)));
    return BVH_AMD_OK;
}

// (the root's reference-layout nodes are made resident first, host edits pushed and re-validated: capi.hip nodes_resident)
#define BVH_AMD_BROADCAST(S, T, DIM)                                                                                                   \
    BVH_AMD_API struct bvh##S* bvh##S##_broadcast(struct bvh_amd_comm* c, int root, struct bvh##S* bvh, const void* d_prims, size_t prim_bytes, \
                                                  void** d_prims_out, size_t* prim_bytes_out, void* stream) {                          \
        BvhImpl<T>* b = reinterpret_cast<BvhImpl<T>*>(bvh);                                                                             \
        std::string why;                                       /* a root that cannot send still takes part: the header tells the others */ \
        const bool refused = b && c && c->rank == root && nodes_resident<T>(*b) != BVH_AMD_OK;                                          \
        if (refused) why = std::string("broadcast: ") + bvh_amd_last_error();                                                           \
        return reinterpret_cast<bvh##S*>(broadcast_scene<T>(c, root, b, DIM, d_prims, prim_bytes, d_prims_out, prim_bytes_out,          \
                                                            static_cast<hipStream_t>(stream), refused ? &why : nullptr));               \
    }                                                                                                                                   \
    BVH_AMD_API int bvh##S##_replicate(struct bvh##S* bvh, const void* d_prims, size_t prim_bytes, int n_devices, const int* devices,   \
                                       struct bvh##S** bvhs_out, void** d_prims_out) {                                                  \
        BvhImpl<T>* b = reinterpret_cast<BvhImpl<T>*>(bvh);                                                                             \
        if (b) {                                                                                                                        \
            DeviceGuard guard;                                                                                                          \
            BVH_HIP_TRY(hipSetDevice(b->device), BVH_AMD_ERR_HIP);                                                                      \
            const int rc = nodes_resident<T>(*b);                                                                                       \
            if (rc) return rc;                                                                                                          \
        }                                                                                                                               \
        return replicate_scene<T>(b, DIM, d_prims, prim_bytes, n_devices, devices, reinterpret_cast<BvhImpl<T>**>(bvhs_out), d_prims_out); \
    }

BVH_AMD_BROADCAST(3f, float, 3)
BVH_AMD_BROADCAST(3d, double, 3)
BVH_AMD_BROADCAST(2f, float, 2)
BVH_AMD_BROADCAST(2d, double, 2)
#undef BVH_AMD_BROADCAST

} // extern "C"
