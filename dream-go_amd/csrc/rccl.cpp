// The one collective of the path, inside the library: an RCCL broadcast of the packed weight image (include/dgnn.h,
// "multi-GPU").  Replaces the per-device host upload of every tensor (src/libdg_nn/tensor.rs:123-141, run lazily by each
// device's first forward) when weights are loaded or refreshed: the rank that read the file sends ONE contiguous image
// over xGMI, the receivers take the launch scalars (gates, temperature) from the header it carries.
//
// librccl is opened at run time (dlopen), not linked: a host that never broadcasts -- the single-GPU engine, the CPU
// tests -- loads libdgnn_hip.so with nothing but the HIP runtime, and a host that already has RCCL in the process (a
// PyTorch harness, for one) shares that copy instead of loading a second one.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "dgnn_internal.h"
#include "kernels.h"

namespace dgnn {
namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;
    decltype(&ncclCommUserRank) comm_user_rank = nullptr;
    decltype(&ncclBroadcast) broadcast = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};

thread_local int g_last_rccl_error = 0;

const Rccl* rccl()
{
    static Rccl lib;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* off = std::getenv("DGNN_RCCL_DISABLE");                   // tests: behave as if librccl were not installed
        if (off && std::atoi(off) == 1) return;
        // a copy that is already in the process first (same soname: PyTorch-ROCm ships one), then the ROCm installation
        const char* const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        void* h = nullptr;
        for (const char* name : names) if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;
        if (!h) for (const char* name : names) if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
        Rccl r;
        r.handle = h;
#define DGNN_SYM(field, symbol) r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, #symbol))
        DGNN_SYM(get_unique_id, ncclGetUniqueId);
        DGNN_SYM(comm_init_rank, ncclCommInitRank);
        DGNN_SYM(comm_init_all, ncclCommInitAll);
        DGNN_SYM(comm_destroy, ncclCommDestroy);
        DGNN_SYM(comm_count, ncclCommCount);
        DGNN_SYM(comm_user_rank, ncclCommUserRank);
        DGNN_SYM(broadcast, ncclBroadcast);
        DGNN_SYM(all_gather, ncclAllGather);
        DGNN_SYM(group_start, ncclGroupStart);
        DGNN_SYM(group_end, ncclGroupEnd);
        DGNN_SYM(error_string, ncclGetErrorString);
#undef DGNN_SYM
        if (r.get_unique_id && r.comm_init_rank && r.comm_init_all && r.comm_destroy && r.comm_count && r.comm_user_rank &&
            r.broadcast && r.all_gather && r.group_start && r.group_end && r.error_string)
            lib = r;
    });
    return lib.handle ? &lib : nullptr;
}

#define DGNN_NCCL(expr)                                  \
    do {                                                 \
        const ncclResult_t nccl_status__ = (expr);       \
        if (nccl_status__ != ncclSuccess) {              \
            g_last_rccl_error = (int)nccl_status__;      \
            return DGNN_COMM_ERROR;                      \
        }                                                \
    } while (0)

#define DGNN_HIP_(expr)                                  \
    do {                                                 \
        const hipError_t hip_status__ = (expr);          \
        if (hip_status__ != hipSuccess) {                \
            set_last_hip_error((int)hip_status__);       \
            return DGNN_HIP_ERROR;                       \
        }                                                \
    } while (0)

}  // namespace

struct Comm {
    ncclComm_t comm = nullptr;
    int device = 0;
    int rank = 0;
    int num_ranks = 0;
    bool owned = true;
};

}  // namespace dgnn

using namespace dgnn;

extern "C" {

int dgnn_rccl_available(void) { return rccl() ? 1 : 0; }

int dgnn_last_rccl_error(void) { return g_last_rccl_error; }

const char* dgnn_rccl_error_string(int code)
{
    const Rccl* r = rccl();
    return r ? r->error_string((ncclResult_t)code) : "librccl not found";
}

int dgnn_rccl_unique_id(uint8_t* id_out)
{
    if (!id_out) return DGNN_BAD_ARGUMENT;
    const Rccl* r = rccl();
    if (!r) return DGNN_COMM_ERROR;
    static_assert(sizeof(ncclUniqueId) == DGNN_RCCL_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    DGNN_NCCL(r->get_unique_id(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return DGNN_OK;
}

int dgnn_rccl_comm_create(int device, int num_ranks, int rank, const uint8_t* id, dgnn_comm** out)
{
    if (!id || !out || num_ranks <= 0 || rank < 0 || rank >= num_ranks) return DGNN_BAD_ARGUMENT;
    *out = nullptr;
    const Rccl* r = rccl();
    if (!r) return DGNN_COMM_ERROR;
    int count = 0;
    DGNN_HIP_(dgnn::device_count(&count));
    if (device < 0 || device >= count) return DGNN_BAD_ARGUMENT;
    DGNN_HIP_(dgnn::set_device(device));
    Comm* c = new (std::nothrow) Comm();
    if (!c) return DGNN_OUT_OF_MEMORY;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    const ncclResult_t status = r->comm_init_rank(&c->comm, num_ranks, uid, rank);
    if (status != ncclSuccess) { g_last_rccl_error = (int)status; delete c; return DGNN_COMM_ERROR; }
    c->device = device;
    c->rank = rank;
    c->num_ranks = num_ranks;
    *out = reinterpret_cast<dgnn_comm*>(c);
    return DGNN_OK;
}

int dgnn_rccl_comm_wrap(void* nccl_comm, int device, dgnn_comm** out)
{
    if (!nccl_comm || !out) return DGNN_BAD_ARGUMENT;
    *out = nullptr;
    const Rccl* r = rccl();
    if (!r) return DGNN_COMM_ERROR;
    Comm* c = new (std::nothrow) Comm();
    if (!c) return DGNN_OUT_OF_MEMORY;
    c->comm = (ncclComm_t)nccl_comm;
    c->owned = false;
    c->device = device;
    ncclResult_t status = r->comm_count(c->comm, &c->num_ranks);
    if (status == ncclSuccess) status = r->comm_user_rank(c->comm, &c->rank);
    if (status != ncclSuccess) { g_last_rccl_error = (int)status; delete c; return DGNN_COMM_ERROR; }
    *out = reinterpret_cast<dgnn_comm*>(c);
    return DGNN_OK;
}

int dgnn_rccl_comm_info(const dgnn_comm* comm, int* num_ranks, int* rank, int* device)
{
    if (!comm) return DGNN_BAD_ARGUMENT;
    const Comm* c = reinterpret_cast<const Comm*>(comm);
    if (num_ranks) *num_ranks = c->num_ranks;
    if (rank) *rank = c->rank;
    if (device) *device = c->device;
    return DGNN_OK;
}

void dgnn_rccl_comm_destroy(dgnn_comm* comm)
{
    if (!comm) return;
    Comm* c = reinterpret_cast<Comm*>(comm);
    const Rccl* r = rccl();
    if (r && c->owned && c->comm) {
        (void)dgnn::set_device(c->device);
        (void)r->comm_destroy(c->comm);
    }
    delete c;
}

// Collective.  Order of business, so that no failure leaves a rank computing on something it should not:
//   1. every rank's (image bytes, architecture hash) is all-gathered and compared: ranks that do not hold the same architecture
//      and packing would post broadcasts of different sizes -- a hang or a buffer overrun.  On a mismatch EVERY rank returns
//      DGNN_BAD_ARGUMENT (each of them sees all descriptors) and nothing has been touched;
//   2. receivers take the image into a STAGING buffer, not into the copy their kernels read;
//   3. dgnn_network_set_weight_image checks the header of what arrived and only then replaces the host image, the device copy
//      and the launch scalars (gates, temperature): a transfer that failed or delivered a foreign image changes nothing.
int dgnn_broadcast_weights(dgnn_network* net, dgnn_comm* comm, int root)
{
    if (!net || !comm) return DGNN_BAD_ARGUMENT;
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (root < 0 || root >= c->num_ranks) return DGNN_BAD_ARGUMENT;
    const Rccl* r = rccl();
    if (!r) return DGNN_COMM_ERROR;

    uint64_t mine[2];
    int status = dgnn_network_image_descriptor(net, mine);
    if (status != DGNN_OK) return status;
    const size_t num_bytes = (size_t)mine[0];
    DGNN_HIP_(dgnn::set_device(c->device));
    hipStream_t stream = nullptr;
    DGNN_HIP_(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

    // 1. agree on what is about to move
    std::vector<uint64_t> all((size_t)2 * c->num_ranks, 0);
    uint64_t* d_desc = nullptr;
    hipError_t hip = hipMalloc((void**)&d_desc, sizeof(uint64_t) * 2 * ((size_t)c->num_ranks + 1));
    ncclResult_t nccl = ncclSuccess;
    if (hip == hipSuccess) hip = hipMemcpyAsync(d_desc, mine, sizeof(mine), hipMemcpyHostToDevice, stream);
    if (hip == hipSuccess) nccl = r->all_gather(d_desc, d_desc + 2, 2, ncclUint64, c->comm, stream);
    if (hip == hipSuccess && nccl == ncclSuccess) hip = hipMemcpyAsync(all.data(), d_desc + 2, sizeof(uint64_t) * all.size(), hipMemcpyDeviceToHost, stream);
    if (hip == hipSuccess && nccl == ncclSuccess) hip = hipStreamSynchronize(stream);
    if (d_desc) (void)hipFree(d_desc);
    if (nccl != ncclSuccess) { (void)hipStreamDestroy(stream); g_last_rccl_error = (int)nccl; return DGNN_COMM_ERROR; }
    if (hip != hipSuccess) { (void)hipStreamDestroy(stream); set_last_hip_error((int)hip); return DGNN_HIP_ERROR; }
    for (int k = 0; k < c->num_ranks; ++k)
        if (all[2 * k] != all[2 * (size_t)root] || all[2 * k + 1] != all[2 * (size_t)root + 1]) { (void)hipStreamDestroy(stream); return DGNN_BAD_ARGUMENT; }

    // 2. the transfer: the root sends the copy its kernels read (uploaded on first use), a receiver takes it into staging
    void* buffer = nullptr;
    bool staging = false;
    if (c->rank == root) {
        size_t n = 0;
        status = dgnn_network_weight_image(net, c->device, &buffer, &n);
        if (status != DGNN_OK) { (void)hipStreamDestroy(stream); return status; }
    } else {
        hip = hipMalloc(&buffer, num_bytes);
        if (hip != hipSuccess) { (void)hipStreamDestroy(stream); set_last_hip_error((int)hip); return DGNN_HIP_ERROR; }
        staging = true;
    }
    nccl = r->broadcast(buffer, buffer, num_bytes, ncclUint8, root, c->comm, stream);
    hip = hipStreamSynchronize(stream);
    (void)hipStreamDestroy(stream);
    if (nccl != ncclSuccess) { g_last_rccl_error = (int)nccl; status = DGNN_COMM_ERROR; }
    else if (hip != hipSuccess) { set_last_hip_error((int)hip); status = DGNN_HIP_ERROR; }
    // 3. receivers: header check, then the image and ITS gates and temperature become this network's (nothing to do on the root)
    else if (staging) status = dgnn_network_set_weight_image(net, c->device, buffer, num_bytes, 1);
    if (staging) (void)hipFree(buffer);
    return status;
}

int dgnn_broadcast_weights_devices(dgnn_network* net, const int* devices, int num_devices)
{
    if (!net || num_devices <= 0) return DGNN_BAD_ARGUMENT;
    std::vector<int> list;
    if (devices) list.assign(devices, devices + num_devices);
    else for (int d = 0; d < num_devices; ++d) list.push_back(d);
    for (size_t i = 0; i < list.size(); ++i)
        for (size_t j = 0; j < i; ++j) if (list[i] == list[j]) return DGNN_BAD_ARGUMENT;   // one rank per device

    // the first device gets the host image over PCIe (if it does not hold it yet); with one device that is all there is to do
    std::vector<void*> images(list.size(), nullptr);
    size_t num_bytes = 0;
    int status = dgnn_network_weight_image(net, list[0], &images[0], &num_bytes);
    if (status != DGNN_OK || list.size() == 1) return status;

    // the library and the communicators BEFORE any receiver is touched: without them nothing has changed and the devices
    // upload the host image themselves on first use (the reference's way, tensor.rs:123-141)
    // (test build, DGNN_FAKE_DEVICES: several logical devices share a GPU, which ncclCommInitAll refuses; the one grouped broadcast is
    // then stood in for by device-to-device copies from the root's range -- everything around it, the reserved ranges, the commit that
    // makes them valid, the per-device state, is the code an 8-GPU node runs)
    const bool fake = fake_devices() > 0;
    const Rccl* r = fake ? nullptr : rccl();
    if (!fake && !r) return DGNN_COMM_ERROR;
    std::vector<ncclComm_t> comms(list.size(), nullptr);
    if (!fake) DGNN_NCCL(r->comm_init_all(comms.data(), (int)list.size(), list.data()));
    std::vector<hipStream_t> streams(list.size(), nullptr);
    // the others reserve their range and receive over xGMI; a reserved range counts as empty until the transfer is known to be
    // good (engine.cpp: image_valid), so a failure below cannot leave a device computing on garbage
    for (size_t i = 1; i < list.size() && status == DGNN_OK; ++i) status = dgnn_network_weight_image_reserve(net, list[i], &images[i], &num_bytes);
    for (size_t i = 0; i < list.size() && status == DGNN_OK; ++i) {
        if (dgnn::set_device(list[i]) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) != hipSuccess)
            status = DGNN_HIP_ERROR;
    }
    if (status == DGNN_OK && fake) {
        for (size_t i = 1; i < list.size() && status == DGNN_OK; ++i) {
            (void)dgnn::set_device(list[i]);
            if (hipMemcpyAsync(images[i], images[0], num_bytes, hipMemcpyDeviceToDevice, streams[i]) != hipSuccess) status = DGNN_HIP_ERROR;
        }
    } else if (status == DGNN_OK) {
        ncclResult_t st = r->group_start();                 // one thread, several ranks: the calls must be fused in a group
        for (size_t i = 0; i < list.size() && st == ncclSuccess; ++i) {
            (void)dgnn::set_device(list[i]);
            st = r->broadcast(images[i], images[i], num_bytes, ncclUint8, 0, comms[i], streams[i]);
        }
        const ncclResult_t ended = r->group_end();
        if (st == ncclSuccess) st = ended;
        if (st != ncclSuccess) { g_last_rccl_error = (int)st; status = DGNN_COMM_ERROR; }
    }
    for (size_t i = 0; i < list.size(); ++i) {
        if (streams[i]) {
            (void)dgnn::set_device(list[i]);
            if (hipStreamSynchronize(streams[i]) != hipSuccess && status == DGNN_OK) status = DGNN_HIP_ERROR;
            (void)hipStreamDestroy(streams[i]);
        }
        if (comms[i]) (void)r->comm_destroy(comms[i]);
    }
    // good: the receivers' ranges are the image now.  Bad: they stay marked empty and are filled from the host on first use.
    if (status == DGNN_OK) for (size_t i = 1; i < list.size(); ++i) (void)dgnn_network_weight_image_commit(net, list[i]);
    return status;
}

}  // extern "C"
