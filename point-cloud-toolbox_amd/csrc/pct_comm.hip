// Multi-GPU exchange behind the C ABI: the cloud shards by point-index range, the only exchange is one
// all-gather of the float32 coordinate shards (12 B / point) -- RCCL over xGMI, on a stream of its own so that the
// gather of the next cloud overlaps the kernels of the current one.  No PyTorch: one process per GPU, launched by
// anything that sets RANK / WORLD_SIZE / LOCAL_RANK (torchrun as a launcher only); rank 0 creates the RCCL unique id
// (pct_comm_unique_id) and the host side hands its 128 bytes to the other ranks (dist.py: a TCP socket).
//
// librccl.so is opened at run time, on the first call that needs it: a single-GPU process (and the CPU-only build
// container) never loads it, and the library resolves against the ROCm installation the process already uses.
#include "pct_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

struct pct_comm {
    void* lib = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;      // the exchange runs here, beside the handle's compute stream
    hipEvent_t done = nullptr, ready = nullptr;
    bool in_flight = false;            // an exchange has been enqueued and nobody has waited for it yet (one `done` event)
    void* scratch = nullptr;           // 64 B of device memory for the scalar reductions
    void* padded = nullptr;            // world x max(counts) floats: receive buffer of the padded all-gather (unequal shards)
    size_t padded_cap = 0;
    int64_t issued[4] = {0, 0, 0, 0};  // collectives issued: [0] ncclAllGather in place, [1] padded ncclAllGather + compaction,
                                       // [2] groups of per-rank ncclBroadcast, [3] ncclAllReduce
    ncclResult_t (*InitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*Destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*ErrorString)(ncclResult_t) = nullptr;
};

namespace {

void* open_rccl(char* err, size_t err_len) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (!lib && err) snprintf(err, err_len, "cannot open librccl.so: %s", dlerror());
    return lib;
}

template <class F>
bool sym(void* lib, const char* name, F* out) {
    *out = (F)dlsym(lib, name);
    return *out != nullptr;
}

#define PCT_NCCL(ctx, c, call)                                                                                  \
    do {                                                                                                        \
        ncclResult_t r_ = (call);                                                                               \
        if (r_ != ncclSuccess)                                                                                  \
            return pct_fail((ctx), PCT_ERR_HIP, "%s failed: %s", #call, (c)->ErrorString ? (c)->ErrorString(r_) : "?"); \
    } while (0)

// padded all-gather -> back-to-back shards: rank r's slot holds counts[r] floats at r * pitch; off[r] = its place
// in the gathered array (off[world] = total).  One pass, 16 B per lane where the alignment allows.
struct ShardOffsets { int64_t v[65]; };
__global__ __launch_bounds__(256) void k_unpad(const float* __restrict__ padded, int64_t pitch, ShardOffsets off,
                                               int world, float* __restrict__ out) {
    const int r = blockIdx.y;
    if (r >= world) return;
    const int64_t n = off.v[r + 1] - off.v[r];
    const float* src = padded + (int64_t)r * pitch;
    float* dst = out + off.v[r];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

int need_comm(pct_ctx* ctx) {
    if (!ctx) return PCT_ERR_INVALID;
    ctx->err[0] = 0;
    if (!ctx->comm) return pct_fail(ctx, PCT_ERR_INVALID, "no communicator: call pct_comm_init first");
    PCT_HIP(ctx, hipSetDevice(ctx->device));
    return PCT_OK;
}

}  // namespace

void pct_comm_release(pct_ctx* ctx) {
    pct_comm* c = ctx->comm;
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && c->Destroy) (void)c->Destroy(c->comm);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->padded) (void)hipFree(c->padded);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    // (the library stays open: RCCL keeps process-wide state)
    delete c;
    ctx->comm = nullptr;
}

extern "C" {

int pct_comm_unique_id(void* id128) {
    if (!id128) return PCT_ERR_INVALID;
    void* lib = open_rccl(nullptr, 0);
    if (!lib) return PCT_ERR_HIP;
    ncclResult_t (*get)(ncclUniqueId*) = nullptr;
    if (!sym(lib, "ncclGetUniqueId", &get)) return PCT_ERR_HIP;
    ncclUniqueId id;
    if (get(&id) != ncclSuccess) return PCT_ERR_HIP;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return PCT_OK;
}

int pct_comm_init(pct_ctx* ctx, int32_t rank, int32_t world, const void* id128) {
    if (!ctx) return PCT_ERR_INVALID;
    ctx->err[0] = 0;
    if (!id128 || world < 1 || rank < 0 || rank >= world) return pct_fail(ctx, PCT_ERR_INVALID, "bad rank %d of %d", rank, world);
    if (ctx->comm) return pct_fail(ctx, PCT_ERR_INVALID, "the handle already has a communicator");
    PCT_HIP(ctx, hipSetDevice(ctx->device));
    pct_comm* c = new pct_comm();
    c->lib = open_rccl(ctx->err, sizeof(ctx->err));
    if (!c->lib) { delete c; return PCT_ERR_HIP; }
    const bool ok = sym(c->lib, "ncclCommInitRank", &c->InitRank) && sym(c->lib, "ncclCommDestroy", &c->Destroy) &&
                    sym(c->lib, "ncclAllGather", &c->AllGather) && sym(c->lib, "ncclBroadcast", &c->Broadcast) &&
                    sym(c->lib, "ncclAllReduce", &c->AllReduce) && sym(c->lib, "ncclGroupStart", &c->GroupStart) &&
                    sym(c->lib, "ncclGroupEnd", &c->GroupEnd) && sym(c->lib, "ncclGetErrorString", &c->ErrorString);
    if (!ok) { delete c; return pct_fail(ctx, PCT_ERR_HIP, "librccl.so lacks an expected symbol"); }
    c->rank = rank;
    c->world = world;
    ctx->comm = c;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&c->scratch, 64);
    if (e != hipSuccess) {
        pct_comm_release(ctx);
        return pct_fail(ctx, PCT_ERR_HIP, "communicator resources: %s", hipGetErrorString(e));
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    const ncclResult_t r = c->InitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        const char* msg = c->ErrorString(r);
        c->comm = nullptr;
        pct_comm_release(ctx);
        return pct_fail(ctx, PCT_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, msg);
    }
    return PCT_OK;
}

int pct_comm_destroy(pct_ctx* ctx) {
    if (!ctx) return PCT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    pct_comm_release(ctx);
    return PCT_OK;
}

// All-gather of float32 shards: rank r contributes counts[r] floats from dev_send, dev_recv receives them back to
// back in rank order.  Asynchronous: enqueued on the exchange stream, which first waits for what the compute stream
// has been given so far (a buffer the previous pass still reads is not overwritten under it).  ONE exchange may be in
// flight per handle (one completion event): a second call before pct_comm_wait / pct_comm_synchronize is refused.
// Equal shards: ncclAllGather straight into dev_recv.  Unequal shards (C5's 20 022 479 points over 8 ranks; a cloud
// smaller than the world leaves ranks with nothing): every shard padded to the largest -- one ncclAllGather into a
// scratch buffer, one compaction pass (k_unpad) -- the standard form, one collective whatever the sizes.
// PCT_COMM_FORCE = allgather | padded | bcast picks the form regardless (bcast: one ncclBroadcast per non-empty rank
// inside a group; kept as an alternative and exercised by the tests), so that every form can be run with one rank.
int pct_comm_allgather_f32(pct_ctx* ctx, const void* dev_send, void* dev_recv, const int64_t* counts) {
    PCT_TRY(need_comm(ctx));
    pct_comm* c = ctx->comm;
    if (!dev_send || !dev_recv || !counts) return pct_fail(ctx, PCT_ERR_INVALID, "null exchange buffer");
    if (c->in_flight) return pct_fail(ctx, PCT_ERR_INVALID, "an exchange is already in flight on this handle: pct_comm_wait first");
    bool equal = true;
    int64_t most = 0, total = 0;
    for (int r = 0; r < c->world; ++r) {
        if (counts[r] < 0) return pct_fail(ctx, PCT_ERR_INVALID, "negative shard size");
        equal = equal && counts[r] == counts[0];
        most = counts[r] > most ? counts[r] : most;
        total += counts[r];
    }
    enum { AG = 0, PADDED = 1, BCAST = 2 };
    int mode = equal ? AG : PADDED;
    if (const char* f = pct_getenv("PCT_COMM_FORCE")) {
        if (!strcmp(f, "bcast")) mode = BCAST;
        else if (!strcmp(f, "padded")) mode = PADDED;
        else if (!strcmp(f, "allgather") && equal) mode = AG;
    }
    if (total == 0) return PCT_OK;                         // nothing to move (and nothing to wait for)
    PCT_HIP(ctx, hipEventRecord(c->ready, ctx->stream));
    PCT_HIP(ctx, hipStreamWaitEvent(c->stream, c->ready, 0));
    if (mode == AG) {
        PCT_NCCL(ctx, c, c->AllGather(dev_send, dev_recv, (size_t)counts[0], ncclFloat32, c->comm, c->stream));
    } else if (mode == PADDED) {
        if (c->world > 64) return pct_fail(ctx, PCT_ERR_INVALID, "more than 64 ranks");
        const size_t need = (size_t)c->world * (size_t)most * sizeof(float);
        if (need > c->padded_cap) {
            PCT_HIP(ctx, hipStreamSynchronize(c->stream));
            if (c->padded) (void)hipFree(c->padded);
            c->padded = nullptr; c->padded_cap = 0;
            PCT_HIP(ctx, hipMalloc(&c->padded, need + need / 8));
            c->padded_cap = need + need / 8;
        }
        ShardOffsets off;
        off.v[0] = 0;
        for (int r = 0; r < c->world; ++r) off.v[r + 1] = off.v[r] + counts[r];
        // my shard into my slot is the collective's job too: ncclAllGather reads `most` floats from the send buffer, so
        // the send side goes through the slot (a shard shorter than the largest must not be read past its end)
        float* mine = (float*)c->padded + (size_t)c->rank * (size_t)most;
        if (counts[c->rank] > 0)
            PCT_HIP(ctx, hipMemcpyAsync(mine, dev_send, (size_t)counts[c->rank] * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        PCT_NCCL(ctx, c, c->AllGather(mine, c->padded, (size_t)most, ncclFloat32, c->comm, c->stream));
        const int bx = (int)((most + 255) / 256 < 2048 ? (most + 255) / 256 : 2048);
        PCT_LAUNCH(k_unpad, dim3(bx > 0 ? bx : 1, c->world), dim3(256), 0, c->stream, (const float*)c->padded, most,
                           off, c->world, (float*)dev_recv);
        PCT_HIP(ctx, hipGetLastError());
    } else {
        // one broadcast per rank that has something, fused by the group
        PCT_NCCL(ctx, c, c->GroupStart());
        int64_t off = 0;
        for (int r = 0; r < c->world; ++r) {
            float* dst = (float*)dev_recv + off;
            if (counts[r] > 0) {
                const ncclResult_t rr = c->Broadcast(r == c->rank ? dev_send : (const void*)dst, dst, (size_t)counts[r], ncclFloat32, r, c->comm, c->stream);
                if (rr != ncclSuccess) {
                    (void)c->GroupEnd();
                    return pct_fail(ctx, PCT_ERR_HIP, "ncclBroadcast failed: %s", c->ErrorString(rr));
                }
            }
            off += counts[r];
        }
        PCT_NCCL(ctx, c, c->GroupEnd());
    }
    ++c->issued[mode];
    PCT_HIP(ctx, hipEventRecord(c->done, c->stream));
    c->in_flight = true;
    return PCT_OK;
}

// Collectives this handle has issued so far: out4 = {ncclAllGather straight into the receive buffer, padded
// ncclAllGather + compaction, groups of per-rank ncclBroadcast, ncclAllReduce} -- what a report may claim.
int pct_comm_counters(pct_ctx* ctx, int64_t* out4) {
    PCT_TRY(need_comm(ctx));
    if (!out4) return pct_fail(ctx, PCT_ERR_INVALID, "null output");
    memcpy(out4, ctx->comm->issued, sizeof(ctx->comm->issued));
    return PCT_OK;
}

// The compute stream waits (on the device, the host does not block) for the last exchange.
int pct_comm_wait(pct_ctx* ctx) {
    PCT_TRY(need_comm(ctx));
    if (ctx->comm->in_flight) PCT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->comm->done, 0));
    ctx->comm->in_flight = false;
    return PCT_OK;
}

// The host blocks until the exchange stream is idle.
int pct_comm_synchronize(pct_ctx* ctx) {
    PCT_TRY(need_comm(ctx));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->comm->stream));
    ctx->comm->in_flight = false;
    return PCT_OK;
}

// values[0..n) <- reduction over all ranks (op: 0 = sum, 2 = max, 3 = min), n <= 8; blocking.  n == 0: a barrier.
int pct_comm_allreduce_f64(pct_ctx* ctx, double* values, int32_t n, int32_t op) {
    PCT_TRY(need_comm(ctx));
    pct_comm* c = ctx->comm;
    if (n < 0 || n > 8 || (n > 0 && !values) || (op != 0 && op != 2 && op != 3)) return pct_fail(ctx, PCT_ERR_INVALID, "bad reduction");
    double tmp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int m = n > 0 ? n : 1;
    for (int i = 0; i < n; ++i) tmp[i] = values[i];
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));           // a barrier also means: my kernels are done
    PCT_HIP(ctx, hipMemcpyAsync(c->scratch, tmp, m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PCT_NCCL(ctx, c, c->AllReduce(c->scratch, c->scratch, (size_t)m, ncclFloat64, (ncclRedOp_t)op, c->comm, c->stream));
    ++c->issued[3];
    PCT_HIP(ctx, hipMemcpyAsync(tmp, c->scratch, m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PCT_HIP(ctx, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i) values[i] = tmp[i];
    return PCT_OK;
}

}  // extern "C"
