// Multi-GPU inside the library (SURVEY.md 8e, 8b "required extension"): ray batches shard embarrassingly, the ONE exchange of
// the path is a root-to-all broadcast of the scene — the Bvh::serialize byte stream (reference bvh.h:221-229) written into HBM
// straight from the resident nodes (wire.hip) + the BVH-ordered primitive array — over RCCL (xGMI). No other collective exists.
//
//   one process per GPU : bvh_amd_comm_unique_id / bvh_amd_comm_create (ncclCommInitRank) + bvhXX_broadcast   — what
//                         bvh_amd/parallel.py and bench.py --gpus N drive (torch.distributed only carries the 128-byte id);
//   one process, N GPUs : bvhXX_replicate (ncclCommInitAll once per list of devices + one grouped ncclBroadcast) — what a C / C++
//                         user of the reference API (test/c_api_example.c, test/benchmark.cpp) calls before tracing a ray shard
//                         per device.
// librccl is opened on first use (dlopen): nothing here is needed, or loaded, by a single-GPU program.
//
// No payload byte visits the host on any rank: the receivers turn the received device buffer into resident nodes + traversal
// records with deserialize_from_device (device-to-device copies, id narrowing, structural validation, relayout kernels).
#include "common.h"

#include <rccl/rccl.h>                                         // types and prototypes only: the library itself is opened on first use

#include <dlfcn.h>

#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

struct bvh_amd_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1, device = -1;
    bool owned = true;
    // 64 bytes of device memory on the communicator's device: the 32-byte header of a broadcast and the status word the ranks agree
    // on before any payload moves. Allocated with the communicator, so that no rank can fail an allocation between two collectives.
    void* d_words = nullptr;
};

namespace bvh_amd {

namespace {

// RCCL is opened on first use of bvh_amd_comm_* / bvhXX_broadcast / bvhXX_replicate: single-GPU users of libbvh_amd.so
// neither need librccl to be installed nor pay for loading it. Search: $BVH_AMD_RCCL_LIB, the SONAME through the loader's own
// path (which also finds a librccl the process has already loaded, e.g. PyTorch's), $ROCM_PATH/lib, /opt/rocm/lib.
struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string loaded_from, error;
};

const Rccl* rccl() {
    static const Rccl* table = []() -> const Rccl* {
        static Rccl r;
        std::vector<std::string> names;
        if (const char* e = getenv("BVH_AMD_RCCL_LIB")) names.push_back(e);
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        if (const char* e = getenv("ROCM_PATH")) { names.push_back(std::string(e) + "/lib/librccl.so.1"); names.push_back(std::string(e) + "/lib/librccl.so"); }
        names.push_back("/opt/rocm/lib/librccl.so.1");
        names.push_back("/opt/rocm/lib/librccl.so");
        void* dll = nullptr;
        for (const std::string& n : names) {
            dll = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (dll) { r.loaded_from = n; break; }
            if (const char* why = dlerror()) r.error += (r.error.empty() ? "" : "; ") + std::string(why);
        }
        if (!dll) return &r;
        bool ok = true;
        auto sym = [&](const char* name) { void* p = dlsym(dll, name); if (!p) { ok = false; r.error = std::string("librccl lacks ") + name; } return p; };
#define BVH_RCCL_SYM(F) r.F = reinterpret_cast<decltype(r.F)>(sym("nccl" #F))
        BVH_RCCL_SYM(GetUniqueId); BVH_RCCL_SYM(CommInitRank); BVH_RCCL_SYM(CommInitAll); BVH_RCCL_SYM(CommDestroy); BVH_RCCL_SYM(CommUserRank);
        BVH_RCCL_SYM(CommCount); BVH_RCCL_SYM(CommCuDevice); BVH_RCCL_SYM(Broadcast); BVH_RCCL_SYM(AllReduce); BVH_RCCL_SYM(GroupStart);
        BVH_RCCL_SYM(GroupEnd); BVH_RCCL_SYM(GetErrorString);
#undef BVH_RCCL_SYM
        if (ok) r.error.clear(); else r.Broadcast = nullptr;
        return &r;
    }();
    return table;
}
bool rccl_ready() {
    if (rccl()->Broadcast) return true;
    set_error("RCCL is not available (librccl could not be opened: " + rccl()->error + "); the multi-GPU entry points need it");
    return false;
}

#define BVH_NCCL_TRY(expr)                                                                          \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess)                                                                      \
            return ::bvh_amd::fail(BVH_AMD_ERR_HIP, std::string(#expr) + ": " + rccl()->GetErrorString(r_)); \
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

// The exchange, rank by rank (every failure that ONE rank can have alone is turned into something ALL ranks learn before the
// payload is posted, so that nobody is left waiting in a collective its peers never enter):
//   1. header  : ncclBroadcast of 32 bytes from the root — stream bytes, primitive bytes, dimension, scalar type; a root that has
//                nothing valid to send says so here (dim = 0) and every rank returns;
//   2. local   : every rank checks the family against its entry point, allocates its receive buffers, the root serializes;
//   3. status  : ncclAllReduce(min) of one word — 1 = ready, 0 = failed under 2. Any 0: every rank releases what it allocated and
//                returns NULL (the failing rank keeps its own message in bvh_amd_last_error);
//   4. payload : one grouped ncclBroadcast of the stream and the primitives.
// Header and status live in the communicator's own 64 device bytes: no allocation happens between two collectives.
template <typename T>
BvhImpl<T>* broadcast_scene(bvh_amd_comm* c, int root, BvhImpl<T>* bvh, int dim, const void* d_prims, size_t prim_bytes, void** d_prims_out,
                            size_t* prim_bytes_out, hipStream_t stream, const std::string* root_error = nullptr)
{
    if (!c || !c->comm || !c->d_words) { set_error("broadcast: null communicator"); return nullptr; }
    if (!rccl_ready()) return nullptr;
    const Rccl& nccl = *rccl();
    if (root < 0 || root >= c->size) { set_error("broadcast: root rank out of range"); return nullptr; }
    if (!d_prims_out) { set_error("broadcast: d_prims_out is required"); return nullptr; }
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) { set_error("broadcast: the communicator belongs to another device than the current one"); return nullptr; }
    const bool is_root = c->rank == root;
    // a test knob: the root ALSO runs the receiving side on the broadcast buffers and returns that copy (exercises the whole
    // non-root path on a single GPU, where RCCL refuses two ranks on one device)
    static const bool loopback = BVH_DEV_INT("BVH_AMD_BROADCAST_LOOPBACK", 0) != 0;
    // test knob: this rank pretends its local step 2 failed (BVH_AMD_BROADCAST_FAIL_RANK = a rank number): the status round must
    // make every rank return instead of hanging
    static const int fail_rank = BVH_DEV_INT("BVH_AMD_BROADCAST_FAIL_RANK", -1);
    // 1. header. Every check that can fail on the root alone comes before it and is reported THROUGH it.
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
    SceneMeta* d_meta = static_cast<SceneMeta*>(c->d_words);
    int* d_status = reinterpret_cast<int*>(static_cast<char*>(c->d_words) + sizeof(SceneMeta));
    static_assert(sizeof(SceneMeta) + sizeof(int) <= 64);
    // (a failed host <-> device copy of the header cannot be reported to the peers before the collective it feeds: the root then
    //  sends dim = 0 from a zeroed buffer, the others read garbage-free zeros and everybody returns)
    bool header_ok = true;
    if (is_root) header_ok = hipMemcpyAsync(d_meta, &meta, sizeof(meta), hipMemcpyHostToDevice, stream) == hipSuccess;
    if (is_root && !header_ok) (void)hipMemsetAsync(d_meta, 0, sizeof(SceneMeta), stream);
    // From the header on EVERY rank leaves through the status round, whatever happened to it (round 5): a rank whose header
    // call failed used to return while its peers went on to the all-reduce, and a rank that could not read the header back went on
    // alone while the others returned on "the root has nothing to send" (dim == 0). All of these are status 0 now.
    std::string why;
    ncclResult_t r = nccl.Broadcast(d_meta, d_meta, sizeof(SceneMeta), ncclUint8, root, c->comm, stream);
    if (r != ncclSuccess) why = std::string("broadcast: ncclBroadcast(header): ") + nccl.GetErrorString(r);
    SceneMeta got = {0, 0, 0, 0};
    const bool read_ok = why.empty() && hipMemcpyAsync(&got, d_meta, sizeof(got), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
    // 2. local preparation
    void* d_stream = nullptr;
    void* d_recv_prims = nullptr;
    auto drop = [&]() { if (d_stream) (void)hipFree(d_stream); if (d_recv_prims) (void)hipFree(d_recv_prims); d_stream = d_recv_prims = nullptr; };
    const bool receives = !is_root || loopback;
    if (!why.empty()) {}
    else if (!read_ok) why = "broadcast: reading the header failed";
    else if (got.dim == 0)                                     // the root refused (its message is in ITS last_error)
        why = !is_root ? "broadcast: the root rank had nothing valid to send" : !header_ok ? "broadcast: copying the header to the device failed" : bvh_amd_last_error();
    else if (got.dim != static_cast<unsigned long long>(dim) || got.is_double != (sizeof(T) == 8 ? 1ull : 0ull))
        why = "broadcast: the root sends another BVH family (scalar type / dimension) than this entry point receives";
    else if (fail_rank == c->rank) why = "broadcast: fault injection (developer build) made this rank fail its preparation";
    else if (hipMalloc(&d_stream, got.stream_bytes) != hipSuccess) { d_stream = nullptr; why = "broadcast: out of device memory for the serialized BVH"; }
    else if (is_root && serialize_to_device<T>(*bvh, d_stream, got.stream_bytes, stream) != got.stream_bytes) why = std::string("broadcast: ") + bvh_amd_last_error();
    else if (receives && got.prim_bytes && hipMalloc(&d_recv_prims, got.prim_bytes) != hipSuccess) { d_recv_prims = nullptr; why = "broadcast: out of device memory for the primitives"; }
    (void)hipGetLastError();
    // 3. status: does EVERY rank stand ready?
    const int mine = why.empty() ? 1 : 0;
    int all = 0;
    bool status_ok = hipMemcpyAsync(d_status, &mine, sizeof(int), hipMemcpyHostToDevice, stream) == hipSuccess;
    if (!status_ok) (void)hipMemsetAsync(d_status, 0, sizeof(int), stream);                 // (an unreadable status counts as "failed")
    r = nccl.AllReduce(d_status, d_status, 1, ncclInt32, ncclMin, c->comm, stream);
    if (r != ncclSuccess) { drop(); set_error(std::string("broadcast: ncclAllReduce(status): ") + nccl.GetErrorString(r)); return nullptr; }
    if (hipMemcpyAsync(&all, d_status, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) all = 0;
    if (all != 1) {
        drop();
        set_error(!why.empty() ? why : "broadcast: another rank could not prepare its side of the exchange (allocation, family mismatch or serialization; see its last error)");
        return nullptr;
    }
    // 4. payload: the serialized BVH, then the primitives; both device buffers, grouped into one RCCL launch
    const void* send_prims = is_root ? d_prims : d_recv_prims;
    void* recv_prims = receives ? d_recv_prims : const_cast<void*>(d_prims);     // in place on a root that keeps its own copy
    r = nccl.GroupStart();
    if (r == ncclSuccess) r = nccl.Broadcast(d_stream, d_stream, got.stream_bytes, ncclUint8, root, c->comm, stream);
    if (r == ncclSuccess && got.prim_bytes) r = nccl.Broadcast(send_prims, recv_prims, got.prim_bytes, ncclUint8, root, c->comm, stream);
    ncclResult_t r2 = nccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) { drop(); set_error(std::string("broadcast: ncclBroadcast(payload): ") + nccl.GetErrorString(r)); return nullptr; }
    if (prim_bytes_out) *prim_bytes_out = got.prim_bytes;
    if (!receives) {
        if (hipStreamSynchronize(stream) != hipSuccess) { drop(); set_error("broadcast: stream synchronisation failed"); return nullptr; }
        (void)hipFree(d_stream);
        *d_prims_out = const_cast<void*>(d_prims);
        return bvh;
    }
    BvhImpl<T>* out = adopt_stream<T>(d_stream, got.stream_bytes, dim, stream);      // synchronises the stream, frees d_stream
    d_stream = nullptr;
    if (!out) { drop(); return nullptr; }
    *d_prims_out = d_recv_prims;
    return out;
}

// The communicators of bvhXX_replicate, one set per list of devices, kept for the life of the process (ncclCommInitAll costs
// hundreds of milliseconds on an 8-GPU node): a program that replicates scene after scene pays it once.
// bvh_amd_comm_cache_clear() / bvh_amd_release_cached_memory() destroy them. A set is used by one call at a time (the mutex is
// held across the collective: RCCL communicators are not to be driven by two threads at once).
struct CommCache {
    std::mutex mutex;
    std::map<std::vector<int>, std::vector<ncclComm_t>> sets;
};
CommCache& comm_cache() { static CommCache* c = new CommCache; return *c; }     // (never destroyed: no RCCL calls from static destructors)

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
    if (!rccl_ready()) return BVH_AMD_ERR_HIP;
    const Rccl& nccl = *rccl();
    DeviceGuard guard;
    const size_t stream_bytes = wire_size<T>(*bvh);
    std::vector<hipStream_t> streams(static_cast<size_t>(n_devices), nullptr);
    std::vector<void*> bufs(static_cast<size_t>(n_devices), nullptr);
    int rc = BVH_AMD_OK;
    auto cleanup = [&](int code) {
        for (int i = 0; i < n_devices; ++i) {
            (void)hipSetDevice(devs[i]);
            if (streams[i]) { (void)hipStreamSynchronize(streams[i]); (void)hipStreamDestroy(streams[i]); }
            if (bufs[i]) (void)hipFree(bufs[i]);
            if (code != BVH_AMD_OK && i != root) {
                if (d_prims_out[i]) { (void)hipFree(d_prims_out[i]); d_prims_out[i] = nullptr; }
                if (bvhs_out[i]) { delete bvhs_out[i]; bvhs_out[i] = nullptr; }
            }
        }
        return code;
    };
    // everything that can fail without RCCL happens before the first collective (one thread drives all ranks here: a failure
    // simply means no collective is issued at all)
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
    {
        CommCache& cache = comm_cache();
        std::lock_guard<std::mutex> lock(cache.mutex);
        std::vector<ncclComm_t>& comms = cache.sets[devs];
        if (comms.empty()) {
            comms.assign(static_cast<size_t>(n_devices), nullptr);
            const ncclResult_t ri = nccl.CommInitAll(comms.data(), n_devices, devs.data());
            if (ri != ncclSuccess) {
                cache.sets.erase(devs);
                return cleanup(fail(BVH_AMD_ERR_HIP, std::string("replicate: ncclCommInitAll: ") + nccl.GetErrorString(ri)));
            }
        }
        ncclResult_t r = nccl.GroupStart();
        for (int i = 0; i < n_devices && r == ncclSuccess; ++i) {
            r = nccl.Broadcast(bufs[root], bufs[i], stream_bytes, ncclUint8, root, comms[i], streams[i]);
            if (r == ncclSuccess && prim_bytes) r = nccl.Broadcast(d_prims, d_prims_out[i], prim_bytes, ncclUint8, root, comms[i], streams[i]);
        }
        ncclResult_t r2 = nccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r == ncclSuccess)                                  // the communicators go back to the cache idle
            for (int i = 0; i < n_devices; ++i) { (void)hipSetDevice(devs[i]); if (hipStreamSynchronize(streams[i]) != hipSuccess) r = ncclUnhandledCudaError; }
        if (r != ncclSuccess) {                                // a communicator that failed mid-collective is not reused
            for (ncclComm_t cm : comms) if (cm) (void)nccl.CommDestroy(cm);
            cache.sets.erase(devs);
            return cleanup(fail(BVH_AMD_ERR_HIP, std::string("replicate: ncclBroadcast: ") + nccl.GetErrorString(r)));
        }
    }
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
    if (!rccl_ready()) return BVH_AMD_ERR_HIP;
    static_assert(sizeof(ncclUniqueId) == BVH_AMD_COMM_ID_BYTES);
    ncclUniqueId id;
    BVH_NCCL_TRY(rccl()->GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return BVH_AMD_OK;
}

// the communicator's 64 device bytes (header + status of bvhXX_broadcast), on the communicator's device
static bool comm_words(bvh_amd_comm& c) {
    DeviceGuard guard;
    if (hipSetDevice(c.device) != hipSuccess || hipMalloc(&c.d_words, 64) != hipSuccess || hipMemset(c.d_words, 0, 64) != hipSuccess) {
        if (c.d_words) (void)hipFree(c.d_words);
        c.d_words = nullptr;
        (void)hipGetLastError();
        set_error("communicator: no device memory for the exchange header");
        return false;
    }
    return true;
}

BVH_AMD_API struct bvh_amd_comm* bvh_amd_comm_create(const void* id_bytes, int n_ranks, int rank) {
    if (!id_bytes || n_ranks < 1 || rank < 0 || rank >= n_ranks) { set_error("comm_create: bad argument"); return nullptr; }
    if (!rccl_ready()) return nullptr;
    auto c = std::make_unique<bvh_amd_comm>();
    if (hipGetDevice(&c->device) != hipSuccess) { set_error("comm_create: no current device"); return nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    ncclResult_t r = rccl()->CommInitRank(&c->comm, n_ranks, id, rank);
    if (r != ncclSuccess) { set_error(std::string("comm_create: ncclCommInitRank: ") + rccl()->GetErrorString(r)); return nullptr; }
    c->rank = rank; c->size = n_ranks;
    if (!comm_words(*c)) { (void)rccl()->CommDestroy(c->comm); return nullptr; }
    return c.release();
}

BVH_AMD_API struct bvh_amd_comm* bvh_amd_comm_adopt(void* nccl_comm) {
    if (!nccl_comm) { set_error("comm_adopt: null ncclComm_t"); return nullptr; }
    if (!rccl_ready()) return nullptr;
    auto c = std::make_unique<bvh_amd_comm>();
    c->comm = static_cast<ncclComm_t>(nccl_comm);
    c->owned = false;
    ncclResult_t r = rccl()->CommUserRank(c->comm, &c->rank);
    if (r == ncclSuccess) r = rccl()->CommCount(c->comm, &c->size);
    if (r == ncclSuccess) r = rccl()->CommCuDevice(c->comm, &c->device);
    if (r != ncclSuccess) { set_error(std::string("comm_adopt: ") + rccl()->GetErrorString(r)); return nullptr; }
    if (!comm_words(*c)) return nullptr;
    return c.release();
}

BVH_AMD_API void bvh_amd_comm_destroy(struct bvh_amd_comm* c) {
    if (!c) return;
    if (c->d_words) { DeviceGuard guard; if (hipSetDevice(c->device) == hipSuccess) (void)hipFree(c->d_words); }
    if (c->owned && c->comm && rccl()->CommDestroy) (void)rccl()->CommDestroy(c->comm);
    delete c;
}

BVH_AMD_API int bvh_amd_comm_rank(const struct bvh_amd_comm* c) { return c ? c->rank : fail(BVH_AMD_ERR_ARG, "comm_rank: null communicator"); }
BVH_AMD_API int bvh_amd_comm_size(const struct bvh_amd_comm* c) { return c ? c->size : fail(BVH_AMD_ERR_ARG, "comm_size: null communicator"); }
BVH_AMD_API void* bvh_amd_comm_handle(const struct bvh_amd_comm* c) { return c ? c->comm : nullptr; }

BVH_AMD_API int bvh_amd_comm_broadcast(struct bvh_amd_comm* c, void* d_buf, size_t bytes, int root, void* stream) {
    if (!c || !c->comm || (bytes && !d_buf)) return fail(BVH_AMD_ERR_ARG, "comm_broadcast: bad argument");
    if (bytes == 0) return BVH_AMD_OK;
    if (!rccl_ready()) return BVH_AMD_ERR_HIP;
    BVH_NCCL_TRY(rccl()->Broadcast(d_buf, d_buf, bytes, ncclUint8, root, c->comm, static_cast<hipStream_t>(stream)));
    return BVH_AMD_OK;
}

// The communicators bvhXX_replicate keeps per list of devices (see CommCache): destroys them; returns how many sets there were.
BVH_AMD_API int bvh_amd_comm_cache_clear(void) {
    CommCache& cache = comm_cache();
    std::lock_guard<std::mutex> lock(cache.mutex);
    const int n = static_cast<int>(cache.sets.size());
    for (auto& kv : cache.sets)
        for (ncclComm_t cm : kv.second) if (cm && rccl()->CommDestroy) (void)rccl()->CommDestroy(cm);
    cache.sets.clear();
    return n;
}

// Where RCCL was found ("" before the first multi-GPU call or when it could not be opened; then bvh_amd_last_error says why)
BVH_AMD_API const char* bvh_amd_rccl_library(void) { return rccl()->Broadcast ? rccl()->loaded_from.c_str() : ""; }

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
