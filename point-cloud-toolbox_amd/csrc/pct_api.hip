// C ABI of the curvature path (include/pct_hip.h).  Host-side orchestration
// only: buffer ownership, launch order, hipEvent timing, status mapping.
#include "pct_internal.h"

#include <execinfo.h>
#include <math.h>
#include <mutex>
#include <signal.h>
#include <stdarg.h>
#include <stdlib.h>
#include <sys/prctl.h>
#include <unistd.h>

int pct_fail(pct_ctx* ctx, int code, const char* fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

extern char** environ;
const char* pct_getenv(const char* name) {
    struct Entry { const char* name; const char* value; };
    static std::mutex m;
    static Entry cache[48];
    static int used = 0;
    static uintptr_t fp = 0;
    std::lock_guard<std::mutex> g(m);
    uintptr_t now = 1469598103934665603ull;
    for (char** e = environ; e && *e; ++e) now = (now ^ (uintptr_t)*e) * 1099511628211ull;     // setenv allocates a new string
    if (now != fp) { fp = now; used = 0; }
    for (int i = 0; i < used; ++i)
        if (cache[i].name == name) return cache[i].value;
    const char* v = getenv(name);
    if (used < 48) cache[used++] = Entry{name, v};
    return v;
}

void pct_release(pct_buf* b) {
    if (b->p) (void)hipFree(b->p);
    b->p = nullptr;
    b->cap = 0;
}

int pct_reserve(pct_ctx* ctx, pct_buf* b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b->cap >= bytes) return PCT_OK;
    if (b->p) {
        PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        pct_release(b);
    }
    size_t want = bytes + bytes / 8 + 256;   // a little head-room for repeated calls with growing sizes
    // (debugging: exact sizes -- every growing request moves the buffer, so code that keeps a pointer across a
    // reserve, or counts on the spare bytes behind a buffer, shows)
    if (pct_getenv("PCT_NO_HEADROOM")) want = bytes;
    hipError_t e = hipMalloc(&b->p, want);
    if (e != hipSuccess) {
        b->p = nullptr;
        return pct_fail(ctx, PCT_ERR_OOM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b->cap = want;
    return PCT_OK;
}

static void release(pct_buf* b) { pct_release(b); }

static float ev_ms(const hipEvent_t* ev, int a, int b) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev[a], ev[b]) != hipSuccess) ms = 0.f;
    return ms;
}
static float ev_ms(pct_ctx* ctx, int a, int b) { return ev_ms(ctx->ev, a, b); }

// pinned words the kernels of a fused call write for the host: sweep statistics (8 x u64) and the SVD row count, one set
// per parity of the call (pct_set_async: the next call's kernels must not overwrite what the pending one's left)
static unsigned long long* stat_slot(pct_ctx* ctx, int par) { return (unsigned long long*)(ctx->pin + (par ? 256 : 192)); }
static long long* svd_slot(pct_ctx* ctx, int par) { return (long long*)(ctx->pin + 2048 + 8 * par); }

// bookkeeping of the pending fused call, whose kernels have finished: timings from its event set, statistics, the
// uneven-density verdict for the next sweep
static void finish_pending(pct_ctx* ctx, const hipEvent_t* ev) {
    pct_timings t = ctx->tm_snap;
    const unsigned long long* c = stat_slot(ctx, ctx->pend_par);
    t.ring_fallbacks = (int64_t)c[0];
    t.lds_overflows = (int64_t)c[1];
    t.flushes = (int64_t)c[2];
    t.candidate_steps = (int64_t)c[3];
    t.redone_queries = (int64_t)c[4];
    t.grid_ms = ev_ms(ev, 2, 3);
    t.knn_ms = ev_ms(ev, 3, 4);
    t.knn_fast_ms = ctx->pend_sorted ? ev_ms(ev, 3, 7) : 0.f;
    t.fit_ms = ev_ms(ev, 4, 6);
    t.total_ms = ev_ms(ev, 2, 6);
    t.fit_svd_rows = *svd_slot(ctx, ctx->pend_par);
    const int64_t redone = (int64_t)(unsigned)(c[7] & 0xFFFFFFFFull);
    if (ctx->pend_sorted && !ctx->pend_levels) ctx->uneven = redone * 20 > ctx->pend_owned && ctx->pend_owned >= 65536;
    else if (ctx->pend_levels && t.levels <= 1) ctx->uneven = false;
    ctx->tm_done = t;
    ctx->pending = false;
}

// every entry point but an asynchronous pct_curvature: wait for the pending call and do its bookkeeping first
static int drain(pct_ctx* ctx) {
    if (!ctx->pending) return PCT_OK;
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    finish_pending(ctx, ctx->ev);
    ctx->tm = ctx->tm_done;
    PCT_HIP(ctx, e);
    return PCT_OK;
}

extern "C" {

const char* pct_version(void) { return "pct_hip 0.1 (gfx950)"; }

int pct_device_count(int* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    if (count) *count = n;
    return e == hipSuccess ? PCT_OK : PCT_ERR_NO_DEVICE;
}

// PCT_ABORT_TRACE=1: a SIGABRT anywhere in the process first writes the NATIVE backtrace and the name of the thread
// that raised it to stderr, then goes on to whoever handled the signal before (Python's fault handler, the default
// action).  Twice in two rounds a GPU test run died with nothing but "Fatal Python error: Aborted" in its log -- no
// GPU fault line, no glibc diagnostic: the raiser is some library's bare abort(), possibly on a runtime
// helper thread that a Python traceback cannot show.  Tests, bench.py and smoke() switch this on.
extern "C++" { const char* pct_last_launch = "(none)"; }
static struct sigaction g_prev_abrt;
static int g_abrt_fd = 2;       // PCT_ABORT_TRACE=<fd>: a descriptor of the REAL stderr (a test runner that captures fd 2
                                // -- pytest -- would swallow the trace with the dying process; tests/conftest.py dups it
                                // before capturing starts)
// Best effort only: backtrace() and backtrace_symbols_fd() are not async-signal-safe (an abort raised from inside
// malloc or the unwinder can hang here instead of dying) -- the hook is OPT-IN (PCT_ABORT_TRACE: tests, bench.py and
// smoke() set it; a host application that loads the library does not get its SIGABRT disposition touched).
static volatile sig_atomic_t g_in_abort_trace = 0;
static void abort_trace(int) {
    if (g_in_abort_trace) {                    // a second abort while tracing the first: give up tracing
        sigaction(SIGABRT, &g_prev_abrt, nullptr);
        raise(SIGABRT);
        return;
    }
    g_in_abort_trace = 1;
    static const char head[] = "\n[pct] SIGABRT -- native backtrace of the raising thread";
    const int fd = g_abrt_fd;
    (void)!write(fd, head, sizeof(head) - 1);
    char name[32] = {0};
    if (prctl(PR_GET_NAME, name, 0, 0, 0) == 0) {
        (void)!write(fd, " (", 2);
        (void)!write(fd, name, strnlen(name, sizeof(name)));
        (void)!write(fd, ")", 1);
    }
    (void)!write(fd, ":\n", 2);
    static const char last[] = "[pct] last kernel launched by the library: ";
    (void)!write(fd, last, sizeof(last) - 1);
    const char* ll = __atomic_load_n(&pct_last_launch, __ATOMIC_RELAXED);
    (void)!write(fd, ll, strlen(ll));
    (void)!write(fd, "\n", 1);
    void* frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, fd);
    sigaction(SIGABRT, &g_prev_abrt, nullptr);
    raise(SIGABRT);
}
static void install_abort_trace() {
    static std::once_flag once;
    std::call_once(once, [] {
        const char* e = getenv("PCT_ABORT_TRACE");
        if (!e) return;
        const int fd = atoi(e);
        if (fd > 2) g_abrt_fd = fd;
        void* warm[4];
        (void)backtrace(warm, 4);                 // loads the unwinder now, not inside the handler
        struct sigaction sa;
        memset(&sa, 0, sizeof(sa));
        sa.sa_handler = abort_trace;
        sigemptyset(&sa.sa_mask);
        sa.sa_flags = 0;
        sigaction(SIGABRT, &sa, &g_prev_abrt);
    });
}

int pct_create(int device, pct_ctx** out) {
    if (!out) return PCT_ERR_INVALID;
    *out = nullptr;
    install_abort_trace();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return PCT_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return PCT_ERR_NO_DEVICE;
    pct_ctx* ctx = new pct_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return PCT_ERR_HIP;
    }
    for (auto& e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            delete ctx;
            return PCT_ERR_HIP;
        }
    for (int i = 2; i <= 7; ++i)               // the second set of timing events (pct_set_async)
        if (hipEventCreate(&ctx->ev_prev[i]) != hipSuccess) {
            delete ctx;
            return PCT_ERR_HIP;
        }
    if (hipHostMalloc((void**)&ctx->pin, 4096, hipHostMallocMapped) != hipSuccess) {
        delete ctx;
        return PCT_ERR_OOM;
    }
    memset(ctx->pin, 0, 4096);
    *out = ctx;
    return PCT_OK;
}

void pct_destroy(pct_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    pct_comm_release(ctx);
    pct_buf* all[] = {&ctx->xyz, &ctx->pts4, &ctx->pts4d, &ctx->cell_of, &ctx->cell_cnt, &ctx->cell_fill,
                      &ctx->scan_tmp, &ctx->occ, &ctx->redo, &ctx->row_of, &ctx->owned_pos, &ctx->cell_own, &ctx->cell_oth, &ctx->own_start, &ctx->sorted4, &ctx->sorted4d, &ctx->red, &ctx->nbr_pos,
                      &ctx->nbr_dist, &ctx->nbr_cnt, &ctx->counters, &ctx->coefs, &ctx->K, &ctx->H, &ctx->H2,
                      &ctx->stage_a, &ctx->stage_b, &ctx->stage_c, &ctx->stage_d, &ctx->row_done, &ctx->redo_m, &ctx->flag_buf, &ctx->dens_buf, &ctx->pub_pos, &ctx->pub_dist, &ctx->pub_cnt, &ctx->qpts4, &ctx->fit_flag, &ctx->lvl_src,
                      &ctx->tree_codes, &ctx->tree_vals, &ctx->tree_lvl, &ctx->tree_head, &ctx->tree_marks, &ctx->tree_seg, &ctx->tree_runs, &ctx->tree_range, &ctx->tree_bucket, &ctx->tree_tmp};
    for (pct_buf* b : all) release(b);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    for (auto& e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->ev_prev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* pct_last_error(const pct_ctx* ctx) { return ctx ? ctx->err : "null context"; }

static int begin_call(pct_ctx* ctx, bool wait_for_pending = true) {
    if (!ctx) return PCT_ERR_INVALID;
    ctx->err[0] = 0;
    PCT_HIP(ctx, hipSetDevice(ctx->device));
    if (wait_for_pending) PCT_TRY(drain(ctx));
    return PCT_OK;
}

static int new_cloud(pct_ctx* ctx, int64_t n) {
    if (n <= 0 || n > (int64_t)INT32_MAX - 1024) return pct_fail(ctx, PCT_ERR_INVALID, "cloud size %lld out of range", (long long)n);
    ctx->n = n;
    ctx->q_begin = 0;
    ctx->q_end = n;
    ctx->slab_parts = ctx->slab_part = 0;
    ctx->slab_split_valid = false;
    ctx->grid_valid = ctx->knn_valid = ctx->fit_valid = ctx->pts4_valid = ctx->qpts4_valid = false;
    ctx->has_f64 = false;
    ctx->no_cull = ctx->culled = false;
    ctx->retries = 0;
    ctx->tm = pct_timings{};
    return PCT_OK;
}

int pct_set_points_f32(pct_ctx* ctx, const float* xyz, int64_t n) {
    PCT_TRY(begin_call(ctx));
    if (!xyz) return pct_fail(ctx, PCT_ERR_INVALID, "null coordinates");
    PCT_TRY(new_cloud(ctx, n));
    PCT_TRY(pct_reserve(ctx, &ctx->xyz, (size_t)n * 3 * sizeof(float)));
    PCT_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    PCT_HIP(ctx, hipMemcpyAsync(ctx->xyz.p, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    ctx->xyz_view = (const float*)ctx->xyz.p;
    PCT_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tm.upload_ms = ev_ms(ctx, 0, 1);
    return PCT_OK;
}

int pct_set_points_f64(pct_ctx* ctx, const double* xyz, int64_t n) {
    PCT_TRY(begin_call(ctx));
    if (!xyz) return pct_fail(ctx, PCT_ERR_INVALID, "null coordinates");
    PCT_TRY(new_cloud(ctx, n));
    PCT_TRY(pct_reserve(ctx, &ctx->stage_a, (size_t)n * 3 * sizeof(double)));
    PCT_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    PCT_HIP(ctx, hipMemcpyAsync(ctx->stage_a.p, xyz, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PCT_TRY(pct_pack_points_f64(ctx, (const double*)ctx->stage_a.p));
    PCT_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->has_f64 = true;
    ctx->tm.upload_ms = ev_ms(ctx, 0, 1);
    return PCT_OK;
}

int pct_set_points_device_f32(pct_ctx* ctx, const void* dev_xyz, int64_t n) {
    PCT_TRY(begin_call(ctx));
    if (!dev_xyz) return pct_fail(ctx, PCT_ERR_INVALID, "null coordinates");
    PCT_TRY(new_cloud(ctx, n));
    PCT_TRY(pct_reserve(ctx, &ctx->xyz, (size_t)n * 3 * sizeof(float)));
    PCT_HIP(ctx, hipMemcpyAsync(ctx->xyz.p, dev_xyz, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->xyz_view = (const float*)ctx->xyz.p;
    return PCT_OK;
}

int pct_use_points_device_f32(pct_ctx* ctx, const void* dev_xyz, int64_t n) {
    PCT_TRY(begin_call(ctx));
    if (!dev_xyz) return pct_fail(ctx, PCT_ERR_INVALID, "null coordinates");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, dev_xyz) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != ctx->device) {
        (void)hipGetLastError();
        return pct_fail(ctx, PCT_ERR_INVALID, "pct_use_points_device_f32: not a device pointer of device %d", ctx->device);
    }
    PCT_TRY(new_cloud(ctx, n));
    ctx->xyz_view = (const float*)dev_xyz;
    return PCT_OK;
}

int pct_set_query_range(pct_ctx* ctx, int64_t begin, int64_t end) {
    PCT_TRY(begin_call(ctx));
    if (ctx->n <= 0) return pct_fail(ctx, PCT_ERR_INVALID, "no cloud loaded");
    if (begin < 0 || end > ctx->n || begin > end) return pct_fail(ctx, PCT_ERR_INVALID, "bad query range [%lld,%lld)", (long long)begin, (long long)end);
    ctx->q_begin = begin;
    ctx->q_end = end;
    ctx->slab_parts = ctx->slab_part = 0;
    ctx->slab_split_valid = false;
    ctx->knn_valid = ctx->fit_valid = ctx->grid_valid = false;   // the cell order depends on the owned range
    ctx->no_cull = false;
    ctx->retries = 0;
    return PCT_OK;
}

int pct_set_query_slab(pct_ctx* ctx, int32_t part, int32_t parts) {
    PCT_TRY(begin_call(ctx));
    if (ctx->n <= 0) return pct_fail(ctx, PCT_ERR_INVALID, "no cloud loaded");
    if (parts > PCT_SLAB_PARTS_MAX || (parts >= 1 && (part < 0 || part >= parts)))
        return pct_fail(ctx, PCT_ERR_INVALID, "bad slab %d of %d (at most %d parts)", part, parts, PCT_SLAB_PARTS_MAX);
    if (parts >= 1 && (ctx->has_f64 || ctx->n < 4096))
        return pct_fail(ctx, PCT_ERR_INVALID, "slab ownership serves float32 clouds of at least 4096 points (use index ranges)");
    ctx->q_begin = 0;
    ctx->q_end = ctx->n;                 // until the build has cut the cloud
    ctx->slab_parts = parts >= 1 ? parts : 0;     // (one part: the whole cloud as one slab -- the same code path, for a world of one rank)
    ctx->slab_part = parts >= 1 ? part : 0;
    ctx->slab_split_valid = false;
    ctx->knn_valid = ctx->fit_valid = ctx->grid_valid = false;
    ctx->no_cull = false;
    ctx->retries = 0;
    return PCT_OK;
}

// by-index getters have no meaning while the rows are those of a slab
static int refuse_in_slab_mode(pct_ctx* ctx, const char* what) {
    if (ctx->slab_parts >= 1)
        return pct_fail(ctx, PCT_ERR_INVALID, "%s: this handle owns a slab, not an index range (pct_slab_records / pct_scatter_records)", what);
    return PCT_OK;
}

int pct_slab_counts(pct_ctx* ctx, int64_t* counts, int32_t parts) {
    PCT_TRY(begin_call(ctx));
    if (ctx->slab_parts < 1 || !ctx->slab_split_valid || !ctx->fit_valid)
        return pct_fail(ctx, PCT_ERR_INVALID, "pct_slab_counts: no slab pass on this handle (pct_set_query_slab, pct_curvature)");
    if (!counts || parts != ctx->slab_parts) return pct_fail(ctx, PCT_ERR_INVALID, "pct_slab_counts: %d parts asked, the cloud was cut into %d", parts, ctx->slab_parts);
    for (int p = 0; p < parts; ++p) counts[p] = ctx->slab_counts[p];
    return PCT_OK;
}

namespace {
// rows of the slab in table (cell) order -> (public index, K, H)
__global__ __launch_bounds__(256) void k_slab_records(const float4* __restrict__ sorted4, const int* __restrict__ owned_pos,
                                                      const float* __restrict__ K, const float* __restrict__ H, int64_t rows,
                                                      float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    out[3 * r + 0] = sorted4[owned_pos[r]].w;          // (the public index as int32 bits)
    out[3 * r + 1] = K[r];
    out[3 * r + 2] = H[r];
}

__global__ __launch_bounds__(256) void k_scatter_records(const float* __restrict__ rec, int64_t n_rec, int64_t begin, int64_t end,
                                                         float* __restrict__ K, float* __restrict__ H, unsigned long long* __restrict__ hits) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool hit = false;
    if (i < n_rec) {
        const int64_t pub = (int64_t)__float_as_int(rec[3 * i + 0]);
        hit = pub >= begin && pub < end;
        if (hit) {
            K[pub - begin] = rec[3 * i + 1];
            H[pub - begin] = rec[3 * i + 2];
        }
    }
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(hits, (unsigned long long)__popcll(m));
}
}  // namespace

int pct_slab_records(pct_ctx* ctx, float* dev_records, int64_t capacity_rows, int64_t* rows_out) {
    PCT_TRY(begin_call(ctx));
    if (ctx->slab_parts < 1 || !ctx->slab_split_valid || !ctx->fit_valid || !ctx->knn_sorted_space || !ctx->fit_row_order)
        return pct_fail(ctx, PCT_ERR_INVALID, "pct_slab_records: no slab pass on this handle (pct_set_query_slab, pct_curvature)");
    const int64_t rows = ctx->q_end - ctx->q_begin;
    if (rows_out) *rows_out = rows;
    if (rows > capacity_rows || (rows > 0 && !dev_records))
        return pct_fail(ctx, PCT_ERR_INVALID, "pct_slab_records: %lld rows, room for %lld", (long long)rows, (long long)capacity_rows);
    if (rows == 0) return PCT_OK;
    PCT_LAUNCH(k_slab_records, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, (const float4*)ctx->sorted4.p,
               (const int*)ctx->owned_pos.p, (const float*)ctx->K.p, (const float*)ctx->H.p, rows, dev_records);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}

int pct_scatter_records(pct_ctx* ctx, const float* dev_records, int64_t n_records, int64_t begin, int64_t end, float* dev_K, float* dev_H) {
    PCT_TRY(begin_call(ctx));
    if (n_records < 0 || begin < 0 || end < begin || (n_records > 0 && !dev_records) || (end > begin && (!dev_K || !dev_H)))
        return pct_fail(ctx, PCT_ERR_INVALID, "pct_scatter_records: bad arguments");
    PCT_TRY(pct_reserve(ctx, &ctx->counters, 64 * sizeof(unsigned long long)));
    unsigned long long* d_hits = (unsigned long long*)ctx->counters.p + 32;
    PCT_HIP(ctx, hipMemsetAsync(d_hits, 0, sizeof(unsigned long long), ctx->stream));
    if (n_records > 0)
        PCT_LAUNCH(k_scatter_records, dim3((unsigned)((n_records + 255) / 256)), dim3(256), 0, ctx->stream, dev_records, n_records, begin,
                   end, dev_K, dev_H, d_hits);
    PCT_HIP(ctx, hipGetLastError());
    unsigned long long* h = (unsigned long long*)(ctx->pin + 176);
    PCT_HIP(ctx, hipMemcpyAsync(h, d_hits, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((int64_t)*h != end - begin)
        return pct_fail(ctx, PCT_ERR_INVALID, "pct_scatter_records: %llu of the %lld records fall into rows [%lld,%lld) -- the slabs do not partition the cloud",
                        *h, (long long)n_records, (long long)begin, (long long)end);
    return PCT_OK;
}

int pct_set_grid_param(pct_ctx* ctx, double occupancy_factor) {
    PCT_TRY(begin_call(ctx));
    ctx->occupancy_factor = occupancy_factor > 0 ? occupancy_factor : 0.0;
    return PCT_OK;
}

int pct_set_stats(pct_ctx* ctx, int32_t enable) {
    PCT_TRY(begin_call(ctx));
    ctx->collect_stats = enable != 0;
    return PCT_OK;
}

// neighbour sweep without timing bookkeeping; events 2..4 bracket grid / sweep
static int run_knn(pct_ctx* ctx, int32_t k, double eps, int32_t algo, bool fuse_fit = false) {
    ctx->levels_fitted = false;
    ctx->skip_dist_req = fuse_fit;
    if (ctx->n <= 0) return pct_fail(ctx, PCT_ERR_INVALID, "no cloud loaded");
    if (k < 1 || k > PCT_K_MAX) return pct_fail(ctx, PCT_ERR_INVALID, "k=%d outside [1,%d]", k, PCT_K_MAX);
    if ((int64_t)k + 1 > ctx->n) return pct_fail(ctx, PCT_ERR_K_TOO_LARGE, "k+1=%d exceeds the cloud size %lld", k + 1, (long long)ctx->n);
    if (!(eps >= 0) || isinf(eps)) eps = 0;
    bool auto_req = algo == PCT_KNN_AUTO;
    if (algo == PCT_KNN_AUTO) algo = ctx->n >= 4096 ? PCT_KNN_GRID : PCT_KNN_BRUTE;
    // cKDTree.query takes any k (pct:83).  The fast sweeps sort lists of one or two registers per lane (k <= 127);
    // longer rows go through the wave-per-query sweeps, whose running list is 64 R wide for any power of two R: the
    // exact sweep over the cell list (any cloud size), the exhaustive sweep where that was asked for.
    if (k > 127) {
        if (algo != PCT_KNN_BRUTE) algo = PCT_KNN_GRID_EXACT;
        auto_req = false;
    }
    if (algo != PCT_KNN_GRID && algo != PCT_KNN_BRUTE && algo != PCT_KNN_GRID_EXACT && algo != PCT_KNN_GRID_LEVELS && algo != PCT_KNN_TREE)
        return pct_fail(ctx, PCT_ERR_INVALID, "unknown algorithm %d", algo);
    if (ctx->slab_parts >= 1) {            // one cell list over the slab and its margin (include/pct_hip.h)
        if (!fuse_fit || (algo != PCT_KNN_GRID && algo != PCT_KNN_GRID_EXACT))
            return pct_fail(ctx, PCT_ERR_INVALID, "slab ownership: pct_curvature with PCT_KNN_AUTO / PCT_KNN_GRID only");
        auto_req = false;
    }
    ctx->knn_valid = ctx->fit_valid = false;
    ctx->k = k;
    ctx->eps = eps;
    PCT_HIP(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    ctx->tm.levels = 0;
    ctx->tm.algo = algo;
    const auto run_levels = [&]() -> int {
        ctx->tm.algo = PCT_KNN_GRID_LEVELS;
        PCT_HIP(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
        ctx->levels_fuse_fit = fuse_fit;
        const int lst = pct_knn_levels(ctx, k, eps);
        ctx->levels_fuse_fit = false;
        PCT_TRY(lst);
        PCT_HIP(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
        PCT_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
        ctx->tm.knn_launches = ctx->tm.levels;
        ctx->knn_valid = true;
        ctx->last_levels = true;
        return PCT_OK;
    };
    // the hierarchical cell list takes whole clouds; a shard asked of it goes down the chain of cell lists
    const bool tree_ok = ctx->q_begin == 0 && ctx->q_end == ctx->n && ctx->n < ((int64_t)1 << 26);
    const auto run_tree = [&]() -> int {
        ctx->tm.algo = PCT_KNN_TREE;
        ctx->last_levels = false;
        bool usable = false;
        PCT_TRY(pct_build_tree(ctx, k, eps, &usable));
        if (!usable && ctx->tree_hint_mismatch) return PCT_OK;       // (a remembered verdict that does not fit this cloud: the caller goes on)
        if (!usable) return run_levels();          // (extents or eps outside what the float32 pre-selection can square)
        PCT_HIP(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
        PCT_TRY(pct_launch_knn_tree(ctx, k, eps));
        PCT_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
        ctx->tm.knn_launches = 1;
        ctx->knn_valid = true;
        return PCT_OK;
    };
    if (algo == PCT_KNN_TREE) return tree_ok ? run_tree() : run_levels();
    // a handle fed a stream of similar clouds: what the census said about the last one of this size still holds
    if (auto_req && tree_ok && ctx->auto_tree_n == ctx->n && (++ctx->auto_tree_calls & 15) != 0 && !pct_getenv("PCT_NO_TREE") &&
        !pct_getenv("PCT_NO_AUTO_LEVELS")) {
        // (the verdict holds for clouds of this size AND this bounding box, within 2 % per face -- the same test the
        // speculative cell-list build applies to "a stream of similar clouds")
        ctx->tree_check_bbox = true;
        ctx->tree_hint_mismatch = false;
        const int st = run_tree();
        ctx->tree_check_bbox = false;
        if (!ctx->tree_hint_mismatch) return st;
        ctx->tree_hint_mismatch = false;
        ctx->auto_tree_n = 0;
        ctx->tm.algo = algo;
    }
    if (algo == PCT_KNN_GRID_LEVELS) return run_levels();
    ctx->last_levels = false;
    const bool grid = algo == PCT_KNN_GRID || algo == PCT_KNN_GRID_EXACT;
    if (grid) {
        ctx->grid_skewed = false;
        ctx->auto_probe = auto_req && algo == PCT_KNN_GRID && tree_ok && ctx->n >= 16384 && !pct_getenv("PCT_NO_TREE") && !pct_getenv("PCT_NO_AUTO_LEVELS");
        ctx->auto_probe_tree = ctx->auto_probe;
        const int bst = pct_build_grid(ctx, k, eps);
        ctx->auto_probe = false;
        PCT_TRY(bst);
        if (ctx->grid_skewed) {                    // the uniform list was not built: far too many cells per point
            ctx->grid_skewed = false;
            const int st = run_tree();
            if (st == PCT_OK && ctx->tm.algo == PCT_KNN_TREE) { ctx->auto_tree_n = ctx->n; memcpy(ctx->auto_tree_bbox, ctx->tree_bbox, sizeof(ctx->tree_bbox)); }
            return st;
        }
        // PCT_KNN_AUTO on a whole cloud: is one cell size enough?  A point of a cloud of even density shares its cell
        // with about as many points as a non-empty cell holds on average; where the density spans decades the first
        // figure (size-biased) runs away from the second.  Only then the work items are counted: the share of queries
        // whose 27-cell stencil overflows the staging area or cannot hold k+1 points -- they would all go through the
        // wave-per-query exact sweep -- and how many stencil cells the others find non-empty (about 9-13 on a surface,
        // up to 27 in a volume, where the chain of cell lists does not pay).
        if (auto_req && algo == PCT_KNN_GRID && ctx->q_begin == 0 && ctx->q_end == ctx->n && ctx->n >= 65536 && ctx->n < ((int64_t)1 << 29) &&
            ctx->nonempty_cells > 0 && !pct_getenv("PCT_NO_AUTO_LEVELS")) {
            const double skew = ctx->tm.occupancy * (double)ctx->nonempty_cells / (double)ctx->n;
            if (pct_getenv("PCT_GRID_DEBUG")) fprintf(stderr, "[auto] occupancy %.1f, %lld non-empty cells, skew %.2f\n", ctx->tm.occupancy, (long long)ctx->nonempty_cells, skew);
            const double skew_min = ctx->auto_probe_tree ? 1.25 : 1.5;      // (the chain needs a wider spread to pay)
            if (!(skew > skew_min)) ctx->auto_tree_n = 0;
            if (skew > skew_min) {
                unsigned long long c[4];
                PCT_TRY(pct_item_census(ctx, k, c));
                const double q = (double)(c[0] ? c[0] : 1), fail = (double)(c[1] + c[2]) / q, fine = (double)c[0] - (double)(c[1] + c[2]);
                const double cells = fine > 0 ? (double)c[3] / fine : 27.0;
                if (pct_getenv("PCT_GRID_DEBUG")) fprintf(stderr, "[auto] census: %llu queries, %llu overflow, %llu short, %.1f non-empty stencil cells\n", c[0], c[1], c[2], cells);
                // (the hierarchical list costs ~1.7x a uniform one whatever the density; every query the uniform list
                // would hand to the exact sweep costs about as much as twelve it answers itself)
                // (the hierarchical list's build -- a dozen launches, three read-backs -- costs ~0.45 ms more than the
                // uniform one whatever the cloud's size, a query handed to the exact sweep ~12 ns: below a million
                // points the predicted share must be larger for the switch to pay)
                const double fail_min = fmax(0.08, 37500.0 / (double)ctx->n);
                if (ctx->auto_probe_tree ? fail > fail_min && cells < 15.0 : fail > 0.30 && fine > 0.02 * q && cells < 15.0) {
                    if (ctx->auto_probe_tree) {
                        const int st = run_tree();
                        if (st == PCT_OK && ctx->tm.algo == PCT_KNN_TREE) { ctx->auto_tree_n = ctx->n; memcpy(ctx->auto_tree_bbox, ctx->tree_bbox, sizeof(ctx->tree_bbox)); }
                        return st;
                    }
                    return run_levels();
                }
                ctx->auto_tree_n = 0;
            }
        }
    } else {
        float bbox[6];
        PCT_TRY(pct_pack_points(ctx, bbox));
        ctx->tm.grid_iters = 0;
        ctx->tm.cells = ctx->tm.occupied_cells = 0;
        ctx->tm.grid_points = ctx->n;
        ctx->tm.cell_size = 0;
    }
    PCT_HIP(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    if (grid)
        PCT_TRY(pct_launch_knn_grid(ctx, k, eps, algo == PCT_KNN_GRID_EXACT));
    else
        PCT_TRY(pct_launch_knn_brute(ctx, k, eps));
    PCT_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    ctx->tm.knn_launches = 1;
    ctx->knn_valid = true;
    return PCT_OK;
}

static int finish_knn_stats(pct_ctx* ctx, bool* beyond_limits) {
    unsigned long long* c = stat_slot(ctx, 0);                           // pinned: a plain DMA, no staging
    // (the fused call's fit kernel has already written them there: one launch less in the step's tail)
    if (!ctx->stats_mirrored)
        PCT_HIP(ctx, hipMemcpyAsync(c, ctx->counters.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));   // [7] low word: rows on the redo list
    ctx->stats_mirrored = false;
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tm.ring_fallbacks = (int64_t)c[0];
    ctx->tm.lds_overflows = (int64_t)c[1];
    ctx->tm.flushes = (int64_t)c[2];
    ctx->tm.candidate_steps = (int64_t)c[3];
    if (pct_getenv("PCT_TREE_DEBUG") && ctx->collect_stats)
        fprintf(stderr, "[tree] redone %llu, up-levelled %llu, candidate steps %llu, costliest exact query: %llu steps (row %llu)\n", c[4], c[0], c[3],
                c[6] >> 32, c[6] & 0xffffffffull);
    ctx->tm.redone_queries = (int64_t)c[4];
    ctx->tm.grid_ms = ev_ms(ctx, 2, 3);
    ctx->tm.knn_ms = ev_ms(ctx, 3, 4);
    ctx->tm.knn_fast_ms = ctx->knn_sorted_space ? ev_ms(ctx, 3, 7) : 0.f;
    // a sharded handle that left far points out of its grid met a query those points could matter to
    *beyond_limits = ctx->knn_sorted_space && ctx->culled && c[5] > 0;
    // uneven density: remember it for the next sweep on this handle (and forget it when a chain needed one pass)
    const int64_t owned = ctx->q_end - ctx->q_begin;
    const int64_t redone = (int64_t)(unsigned)(c[7] & 0xFFFFFFFFull);
    if (ctx->knn_sorted_space && !ctx->last_levels) ctx->uneven = redone * 20 > owned && owned >= 65536;
    else if (ctx->last_levels && ctx->tm.levels <= 1) ctx->uneven = false;
    ctx->tm.limit_retries = ctx->retries;
    return PCT_OK;
}

int pct_knn(pct_ctx* ctx, int32_t k, double eps, int32_t algo) {
    PCT_TRY(begin_call(ctx));
    bool again = false;
    PCT_TRY(run_knn(ctx, k, eps, algo));
    PCT_TRY(finish_knn_stats(ctx, &again));
    if (again) {                      // rare: redo with every point in the grid
        ctx->no_cull = true;
        ctx->retries = 1;
        ctx->cull_box_valid = false;   // the cached box was too small for this cloud: measure it again next time
        PCT_TRY(run_knn(ctx, k, eps, algo));
        PCT_TRY(finish_knn_stats(ctx, &again));
    }
    ctx->tm.fit_ms = 0;
    ctx->tm.total_ms = ev_ms(ctx, 2, 4);
    return PCT_OK;
}

int pct_fit(pct_ctx* ctx) {
    PCT_TRY(begin_call(ctx));
    PCT_TRY(refuse_in_slab_mode(ctx, "pct_fit"));
    if (!ctx->knn_valid) return pct_fail(ctx, PCT_ERR_NO_NEIGHBORS, "plant the neighbour table first (pct_knn)");
    PCT_HIP(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    PCT_TRY(pct_launch_fit_table(ctx));
    PCT_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tm.fit_ms = ev_ms(ctx, 5, 6);
    ctx->tm.fit_svd_rows = *(const long long*)(ctx->pin + 2048);
    ctx->fit_rows = ctx->q_end - ctx->q_begin;
    ctx->fit_valid = true;
    ctx->fit_cloud_aligned = true;
    return PCT_OK;
}

int pct_set_async(pct_ctx* ctx, int32_t enable) {
    PCT_TRY(begin_call(ctx));
    ctx->async_mode = enable != 0;
    return PCT_OK;
}

// pct_curvature of a stream of clouds (pct_set_async): enqueue and return.  The previous call's bookkeeping is done here,
// after the cell list's mid-build wait -- which lies behind all of that call's kernels on the stream --, without a wait
// of its own.
static int curvature_async(pct_ctx* ctx, int32_t k, double eps, int32_t algo) {
    const bool had = ctx->pending;
    if (had)          // the pending call keeps its timing events; this one records into the other set
        for (int i = 2; i <= 7; ++i) { hipEvent_t t = ctx->ev[i]; ctx->ev[i] = ctx->ev_prev[i]; ctx->ev_prev[i] = t; }
    const int par = had ? ctx->pend_par ^ 1 : 0;
    ctx->retries = 0;
    ctx->fit_par = par;            // (also for the fits a chained sweep launches itself)
    int st = run_knn(ctx, k, eps, algo, true);
    if (st == PCT_OK) {
        st = [&]() -> int {
            // (no event of its own for the start of the fit: it starts where the sweep's last event, ev[4], was recorded --
            // every event record is a marker packet between two kernels, ~3 us of dispatch gap)
            ctx->stats_mirrored = false;
            ctx->stats_mirror_req = true;
            const int fs = pct_launch_fit_table(ctx);
            ctx->stats_mirror_req = false;
            PCT_TRY(fs);
            PCT_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
            if (!ctx->stats_mirrored)     // (no fit kernel ran: the chained sweep fitted its passes itself, or there are no rows)
                PCT_HIP(ctx, hipMemcpyAsync(stat_slot(ctx, par), ctx->counters.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
            ctx->stats_mirrored = false;
            return PCT_OK;
        }();
    }
    ctx->fit_par = 0;
    if (st != PCT_OK) {            // leave the handle in the plain state: nothing pending, events where they were
        (void)hipStreamSynchronize(ctx->stream);
        if (had) {
            for (int i = 2; i <= 7; ++i) { hipEvent_t t = ctx->ev[i]; ctx->ev[i] = ctx->ev_prev[i]; ctx->ev_prev[i] = t; }
            finish_pending(ctx, ctx->ev);
        }
        return st;
    }
    if (had) {
        // its kernels lie in front of this call's mid-build wait on the stream: normally long finished (checked, not assumed)
        if (hipEventQuery(ctx->ev_prev[6]) != hipSuccess) (void)hipEventSynchronize(ctx->ev_prev[6]);
        (void)hipGetLastError();
        finish_pending(ctx, ctx->ev_prev);
    }
    ctx->tm.limit_retries = 0;
    ctx->tm_snap = ctx->tm;
    ctx->pend_par = par;
    ctx->pend_sorted = ctx->knn_sorted_space;
    ctx->pend_levels = ctx->last_levels;
    ctx->pend_owned = ctx->q_end - ctx->q_begin;
    ctx->pending = true;
    ctx->fit_rows = ctx->q_end - ctx->q_begin;
    ctx->fit_valid = true;
    ctx->fit_cloud_aligned = true;
    return PCT_OK;
}

int pct_curvature(pct_ctx* ctx, int32_t k, double eps, int32_t algo) {
    // (asynchronous only where nothing of the sweep's verdict is needed before returning: a whole-cloud handle never
    // repeats a pass for points it left out, and statistics are off)
    const bool async = ctx && ctx->async_mode && ctx->n > 0 && ctx->q_begin == 0 && ctx->q_end == ctx->n && ctx->slab_parts == 0 &&
                       !ctx->collect_stats;
    PCT_TRY(begin_call(ctx, !async));
    if (async) return curvature_async(ctx, k, eps, algo);
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool again = false;
        PCT_TRY(run_knn(ctx, k, eps, algo, true));
        ctx->stats_mirrored = false;
        ctx->stats_mirror_req = true;
        PCT_TRY(pct_launch_fit_table(ctx));
        ctx->stats_mirror_req = false;
        PCT_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
        PCT_TRY(finish_knn_stats(ctx, &again));
        if (!again) break;
        ctx->no_cull = true;          // rare: redo with every point in the grid
        ctx->retries = 1;
        ctx->cull_box_valid = false;   // the cached box was too small for this cloud: measure it again next time
    }
    ctx->tm.fit_ms = ev_ms(ctx, 4, 6);            // (the fit starts where the sweep's last event was recorded)
    ctx->tm.fit_svd_rows = *svd_slot(ctx, 0);
    ctx->tm.total_ms = ev_ms(ctx, 2, 6);
    ctx->tm_done = ctx->tm;
    ctx->fit_rows = ctx->q_end - ctx->q_begin;
    ctx->fit_valid = true;
    ctx->fit_cloud_aligned = true;
    return PCT_OK;
}

int pct_get_neighbors(pct_ctx* ctx, int64_t begin, int64_t end, int32_t* idx, float* dist, int32_t* count) {
    PCT_TRY(begin_call(ctx));
    PCT_TRY(refuse_in_slab_mode(ctx, "pct_get_neighbors"));
    if (!ctx->knn_valid) return pct_fail(ctx, PCT_ERR_NO_NEIGHBORS, "no neighbour table");
    if (begin < ctx->q_begin || end > ctx->q_end || begin > end)
        return pct_fail(ctx, PCT_ERR_INVALID, "rows [%lld,%lld) outside the owned range [%lld,%lld)", (long long)begin,
                        (long long)end, (long long)ctx->q_begin, (long long)ctx->q_end);
    const int64_t rows = end - begin;
    if (rows == 0) return PCT_OK;
    const size_t cells = (size_t)rows * ctx->k;
    if (idx) PCT_TRY(pct_reserve(ctx, &ctx->stage_a, cells * sizeof(int)));
    if (dist) PCT_TRY(pct_reserve(ctx, &ctx->stage_b, cells * sizeof(float)));
    if (count) PCT_TRY(pct_reserve(ctx, &ctx->stage_c, (size_t)rows * sizeof(int)));
    PCT_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    PCT_TRY(pct_launch_export_neighbors(ctx, begin, end, idx ? (int*)ctx->stage_a.p : nullptr,
                                        dist ? (float*)ctx->stage_b.p : nullptr, count ? (int*)ctx->stage_c.p : nullptr));
    PCT_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    if (idx) PCT_HIP(ctx, hipMemcpyAsync(idx, ctx->stage_a.p, cells * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (dist) PCT_HIP(ctx, hipMemcpyAsync(dist, ctx->stage_b.p, cells * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (count) PCT_HIP(ctx, hipMemcpyAsync(count, ctx->stage_c.p, (size_t)rows * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tm.export_ms = ev_ms(ctx, 0, 1);
    return PCT_OK;
}

int pct_get_neighbor_rows(pct_ctx* ctx, const int64_t* rows, int64_t n_rows, int32_t* idx, float* dist, int32_t* count) {
    PCT_TRY(begin_call(ctx));
    PCT_TRY(refuse_in_slab_mode(ctx, "pct_get_neighbor_rows"));
    if (!ctx->knn_valid) return pct_fail(ctx, PCT_ERR_NO_NEIGHBORS, "no neighbour table");
    if (!rows || n_rows <= 0) return pct_fail(ctx, PCT_ERR_INVALID, "bad row list");
    for (int64_t i = 0; i < n_rows; ++i)
        if (rows[i] < ctx->q_begin || rows[i] >= ctx->q_end)
            return pct_fail(ctx, PCT_ERR_INVALID, "row %lld outside the owned range", (long long)rows[i]);
    const size_t cells = (size_t)n_rows * ctx->k;
    PCT_TRY(pct_reserve(ctx, &ctx->stage_d, (size_t)n_rows * sizeof(int64_t)));
    if (idx) PCT_TRY(pct_reserve(ctx, &ctx->stage_a, cells * sizeof(int)));
    if (dist) PCT_TRY(pct_reserve(ctx, &ctx->stage_b, cells * sizeof(float)));
    if (count) PCT_TRY(pct_reserve(ctx, &ctx->stage_c, (size_t)n_rows * sizeof(int)));
    PCT_HIP(ctx, hipMemcpyAsync(ctx->stage_d.p, rows, (size_t)n_rows * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    PCT_TRY(pct_launch_export_rows(ctx, (const int64_t*)ctx->stage_d.p, n_rows, idx ? (int*)ctx->stage_a.p : nullptr,
                                   dist ? (float*)ctx->stage_b.p : nullptr, count ? (int*)ctx->stage_c.p : nullptr));
    if (idx) PCT_HIP(ctx, hipMemcpyAsync(idx, ctx->stage_a.p, cells * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (dist) PCT_HIP(ctx, hipMemcpyAsync(dist, ctx->stage_b.p, cells * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (count) PCT_HIP(ctx, hipMemcpyAsync(count, ctx->stage_c.p, (size_t)n_rows * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;
}

// validates host-supplied neighbour rows and stages them on the device (stage_a: ids with a 16-byte row pitch,
// stage_c: counts, stage_d: query ids); the public-order records are packed if they are not resident
static int stage_neighbour_rows(pct_ctx* ctx, const int32_t* idx, const int32_t* count, const int64_t* query, int64_t rows, int32_t k, int32_t* pitch_out, bool need_pts4) {
    if (ctx->n <= 0) return pct_fail(ctx, PCT_ERR_INVALID, "no cloud loaded");
    if (!idx || rows <= 0 || k < 1 || k > 4096) return pct_fail(ctx, PCT_ERR_INVALID, "bad neighbour rows");
    // host-side validation: the reference raises IndexError on such rows (pct:640).  PCT_TRUST_ROWS=1 skips it (test
    // hook for the kernel's own guard: an entry outside the cloud is clamped there and the row reads NaN)
    const bool trust = pct_getenv("PCT_TRUST_ROWS") != nullptr;
    for (int64_t r = 0; r < rows && !trust; ++r) {
        const int m = count ? count[r] : k;
        if (m < 0 || m > k) return pct_fail(ctx, PCT_ERR_INVALID, "row %lld: count %d outside [0,%d]", (long long)r, m, k);
        if (query && (query[r] < 0 || query[r] >= ctx->n)) return pct_fail(ctx, PCT_ERR_INVALID, "row %lld: query index out of range", (long long)r);
        if (!query && r >= ctx->n) return pct_fail(ctx, PCT_ERR_INVALID, "more rows than points and no query list");
        const int32_t* p = idx + r * k;
        for (int j = 0; j < m; ++j)
            if (p[j] < 0 || p[j] >= ctx->n)
                return pct_fail(ctx, PCT_ERR_INVALID, "row %lld: neighbour index %d out of range (IndexError in the reference)", (long long)r, p[j]);
    }
    if (need_pts4 && !ctx->pts4_valid) {
        float bbox[6];
        PCT_TRY(pct_pack_points(ctx, bbox));
    }
    // device rows are 16-byte aligned: pitch = k rounded up to a multiple of 4
    const int32_t pitch = (k + 3) & ~3;
    const size_t cells = (size_t)rows * pitch;
    PCT_TRY(pct_reserve(ctx, &ctx->stage_a, cells * sizeof(int)));
    PCT_HIP(ctx, hipMemcpy2DAsync(ctx->stage_a.p, (size_t)pitch * sizeof(int), idx, (size_t)k * sizeof(int),
                                  (size_t)k * sizeof(int), (size_t)rows, hipMemcpyHostToDevice, ctx->stream));
    if (count) {
        PCT_TRY(pct_reserve(ctx, &ctx->stage_c, (size_t)rows * sizeof(int)));
        PCT_HIP(ctx, hipMemcpyAsync(ctx->stage_c.p, count, (size_t)rows * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    }
    if (query) {
        PCT_TRY(pct_reserve(ctx, &ctx->stage_d, (size_t)rows * sizeof(int64_t)));
        PCT_HIP(ctx, hipMemcpyAsync(ctx->stage_d.p, query, (size_t)rows * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    }
    *pitch_out = pitch;
    return PCT_OK;
}

int pct_fit_indices(pct_ctx* ctx, const int32_t* idx, const int32_t* count, const int64_t* query, int64_t rows, int32_t k) {
    PCT_TRY(begin_call(ctx));
    int32_t pitch = 0;
    PCT_TRY(stage_neighbour_rows(ctx, idx, count, query, rows, k, &pitch, true));
    PCT_TRY(pct_reserve(ctx, &ctx->coefs, (size_t)rows * 6 * sizeof(float)));
    PCT_TRY(pct_reserve(ctx, &ctx->K, (size_t)rows * sizeof(float)));
    PCT_TRY(pct_reserve(ctx, &ctx->H, (size_t)rows * sizeof(float)));
    PCT_TRY(pct_reserve(ctx, &ctx->H2, (size_t)rows * sizeof(float)));
    PCT_HIP(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    PCT_TRY(pct_launch_fit_rows(ctx, (const int*)ctx->stage_a.p, count ? (const int*)ctx->stage_c.p : nullptr,
                                query ? (const int64_t*)ctx->stage_d.p : nullptr, rows, k, pitch, (float*)ctx->coefs.p,
                                (float*)ctx->K.p, (float*)ctx->H.p, (float*)ctx->H2.p, false));
    PCT_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tm.fit_ms = ev_ms(ctx, 5, 6);
    ctx->tm.fit_svd_rows = *(const long long*)(ctx->pin + 2048);
    ctx->fit_rows = rows;
    ctx->fit_row_order = false;
    ctx->fit_valid = true;
    ctx->fit_cloud_aligned = false;   // (a resident neighbour table is untouched and stays valid)
    return PCT_OK;
}

int pct_fit_indices_f64(pct_ctx* ctx, const int32_t* idx, const int32_t* count, const int64_t* query, int64_t rows, int32_t k,
                        double* coefs, double* K, double* H) {
    PCT_TRY(begin_call(ctx));
    if (!coefs || !K || !H) return pct_fail(ctx, PCT_ERR_INVALID, "output arrays missing");
    int32_t pitch = 0;
    PCT_TRY(stage_neighbour_rows(ctx, idx, count, query, rows, k, &pitch, false));
    PCT_TRY(pct_reserve(ctx, &ctx->stage_b, (size_t)rows * 8 * sizeof(double)));
    double* d_c = (double*)ctx->stage_b.p;
    double* d_K = d_c + rows * 6;
    double* d_H = d_K + rows;
    PCT_TRY(pct_launch_fit_rows_f64(ctx, (const int*)ctx->stage_a.p, count ? (const int*)ctx->stage_c.p : nullptr,
                                    query ? (const int64_t*)ctx->stage_d.p : nullptr, rows, k, pitch, d_c, d_K, d_H));
    PCT_HIP(ctx, hipMemcpyAsync(coefs, d_c, (size_t)rows * 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipMemcpyAsync(K, d_K, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipMemcpyAsync(H, d_H, (size_t)rows * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;       // the resident float32 results and the neighbour table are untouched
}

int pct_query_points(pct_ctx* ctx, const double* q_xyz, int64_t m, int32_t k, double eps, int32_t* idx, double* dist) {
    PCT_TRY(begin_call(ctx));
    PCT_TRY(refuse_in_slab_mode(ctx, "pct_query_points"));
    if (ctx->n <= 0 || !ctx->xyz_view) return pct_fail(ctx, PCT_ERR_INVALID, "no cloud loaded");
    if (m < 0 || (m > 0 && (!q_xyz || !idx || !dist))) return pct_fail(ctx, PCT_ERR_INVALID, "bad query arrays");
    if (k < 1 || k > 128) return pct_fail(ctx, PCT_ERR_INVALID, "k=%d outside [1,128]", k);
    if (m == 0) return PCT_OK;
    for (int64_t i = 0; i < 3 * m; ++i)
        if (!isfinite(q_xyz[i])) return pct_fail(ctx, PCT_ERR_NONFINITE, "query point %lld is not finite", (long long)(i / 3));
    if (!(eps >= 0) || isinf(eps)) eps = 0;
    PCT_TRY(pct_reserve(ctx, &ctx->stage_a, (size_t)m * 3 * sizeof(double)));
    PCT_TRY(pct_reserve(ctx, &ctx->stage_b, (size_t)m * k * sizeof(int32_t)));
    PCT_TRY(pct_reserve(ctx, &ctx->stage_c, (size_t)m * k * sizeof(double)));
    PCT_HIP(ctx, hipMemcpyAsync(ctx->stage_a.p, q_xyz, (size_t)m * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PCT_TRY(pct_launch_query_points(ctx, (const double*)ctx->stage_a.p, m, k, eps, (int32_t*)ctx->stage_b.p, (double*)ctx->stage_c.p));
    PCT_HIP(ctx, hipMemcpyAsync(idx, ctx->stage_b.p, (size_t)m * k * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipMemcpyAsync(dist, ctx->stage_c.p, (size_t)m * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;
}

int pct_get_fit(pct_ctx* ctx, int64_t begin, int64_t end, float* coefs, float* K, float* H, float* H2) {
    PCT_TRY(begin_call(ctx));
    PCT_TRY(refuse_in_slab_mode(ctx, "pct_get_fit"));
    if (!ctx->fit_valid) return pct_fail(ctx, PCT_ERR_INVALID, "no fit results");
    const int64_t base = ctx->fit_cloud_aligned ? ctx->q_begin : 0;   // cloud-aligned vs row-aligned results
    if (begin < base || end > base + ctx->fit_rows || begin > end)
        return pct_fail(ctx, PCT_ERR_INVALID, "rows [%lld,%lld) outside the fitted range", (long long)begin, (long long)end);
    const int64_t rows = end - begin, off = begin - base;
    if (rows == 0) return PCT_OK;
    if (ctx->fit_row_order) {          // results live in table-row order: gather the requested public rows first
        float *d_c = nullptr, *d_s = nullptr;
        if (coefs) { PCT_TRY(pct_reserve(ctx, &ctx->stage_a, (size_t)rows * 6 * sizeof(float))); d_c = (float*)ctx->stage_a.p; }
        PCT_TRY(pct_reserve(ctx, &ctx->stage_b, (size_t)rows * 3 * sizeof(float)));
        d_s = (float*)ctx->stage_b.p;
        PCT_TRY(pct_launch_gather_fit(ctx, off, rows, d_c, K ? d_s : nullptr, H ? d_s + rows : nullptr, H2 ? d_s + 2 * rows : nullptr));
        if (coefs) PCT_HIP(ctx, hipMemcpyAsync(coefs, d_c, (size_t)rows * 6 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        // the three arrays lie back to back on the device; where the caller's do too, one copy moves the run
        float* host[3] = {K, H, H2};
        for (int i = 0; i < 3;) {
            if (!host[i]) { ++i; continue; }
            int j = i + 1;
            while (j < 3 && host[j] == host[j - 1] + rows) ++j;
            PCT_HIP(ctx, hipMemcpyAsync(host[i], d_s + (size_t)i * rows, (size_t)(j - i) * rows * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
            i = j;
        }
        PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return PCT_OK;
    }
    if (coefs) PCT_HIP(ctx, hipMemcpyAsync(coefs, (float*)ctx->coefs.p + off * 6, (size_t)rows * 6 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (K) PCT_HIP(ctx, hipMemcpyAsync(K, (float*)ctx->K.p + off, (size_t)rows * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (H) PCT_HIP(ctx, hipMemcpyAsync(H, (float*)ctx->H.p + off, (size_t)rows * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (H2) PCT_HIP(ctx, hipMemcpyAsync(H2, (float*)ctx->H2.p + off, (size_t)rows * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;
}

int pct_curvatures_from_coefficients(pct_ctx* ctx, const float* coefs, int64_t rows, float* K, float* H, float* H2) {
    PCT_TRY(begin_call(ctx));
    if (!coefs || rows <= 0) return pct_fail(ctx, PCT_ERR_INVALID, "bad coefficient rows");
    PCT_TRY(pct_reserve(ctx, &ctx->stage_a, (size_t)rows * 6 * sizeof(float)));
    PCT_TRY(pct_reserve(ctx, &ctx->stage_b, (size_t)rows * 3 * sizeof(float)));
    float* d_out = (float*)ctx->stage_b.p;
    PCT_HIP(ctx, hipMemcpyAsync(ctx->stage_a.p, coefs, (size_t)rows * 6 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PCT_TRY(pct_launch_curvatures(ctx, (const float*)ctx->stage_a.p, rows, d_out, d_out + rows, d_out + 2 * rows));
    if (K) PCT_HIP(ctx, hipMemcpyAsync(K, d_out, (size_t)rows * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (H) PCT_HIP(ctx, hipMemcpyAsync(H, d_out + rows, (size_t)rows * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (H2) PCT_HIP(ctx, hipMemcpyAsync(H2, d_out + 2 * rows, (size_t)rows * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;
}

int pct_plane_rotate(pct_ctx* ctx, const void* nbrs, int32_t is_f64, int64_t batch, int32_t m, double* rotated) {
    PCT_TRY(begin_call(ctx));
    if (!nbrs || !rotated || batch <= 0 || m < 2 || (double)batch * m > 2.0e9) return pct_fail(ctx, PCT_ERR_INVALID, "bad neighbourhood block");
    const size_t count = (size_t)batch * m * 3, esz = is_f64 ? sizeof(double) : sizeof(float);
    if (is_f64) { const double* p = (const double*)nbrs; for (size_t i = 0; i < count; ++i) if (!isfinite(p[i])) return pct_fail(ctx, PCT_ERR_NONFINITE, "Non-finite values in input points"); }
    else { const float* p = (const float*)nbrs; for (size_t i = 0; i < count; ++i) if (!isfinite(p[i])) return pct_fail(ctx, PCT_ERR_NONFINITE, "Non-finite values in input points"); }
    PCT_TRY(pct_reserve(ctx, &ctx->stage_a, count * esz));
    PCT_TRY(pct_reserve(ctx, &ctx->stage_b, count * sizeof(double)));
    PCT_HIP(ctx, hipMemcpyAsync(ctx->stage_a.p, nbrs, count * esz, hipMemcpyHostToDevice, ctx->stream));
    PCT_TRY(pct_launch_plane_rotate(ctx, ctx->stage_a.p, is_f64 != 0, batch, m, (double*)ctx->stage_b.p));
    PCT_HIP(ctx, hipMemcpyAsync(rotated, ctx->stage_b.p, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;
}

int pct_fit_quadric(pct_ctx* ctx, const float* pts, int64_t batch, int32_t m, float* coefs) {
    PCT_TRY(begin_call(ctx));
    if (!pts || !coefs || batch <= 0 || m < 1 || (double)batch * m > 2.0e9) return pct_fail(ctx, PCT_ERR_INVALID, "bad point block");
    const size_t count = (size_t)batch * m * 3;
    for (size_t i = 0; i < count; ++i)
        if (!isfinite(pts[i])) return pct_fail(ctx, PCT_ERR_NONFINITE, "Input contains non-finite values.");
    PCT_TRY(pct_reserve(ctx, &ctx->stage_a, count * sizeof(float)));
    PCT_TRY(pct_reserve(ctx, &ctx->stage_b, (size_t)batch * 6 * sizeof(float)));
    PCT_HIP(ctx, hipMemcpyAsync(ctx->stage_a.p, pts, count * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PCT_TRY(pct_launch_quadric_rows(ctx, (const float*)ctx->stage_a.p, batch, m, (float*)ctx->stage_b.p));
    PCT_HIP(ctx, hipMemcpyAsync(coefs, ctx->stage_b.p, (size_t)batch * 6 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;
}

int pct_neighbor_study_curvatures(pct_ctx* ctx, const int64_t* sample_rows, int64_t n_samples, int32_t n_lo, int32_t n_hi,
                                   float* K_out) {
    PCT_TRY(begin_call(ctx));
    PCT_TRY(refuse_in_slab_mode(ctx, "pct_neighbor_study_curvatures"));
    if (!ctx->knn_valid) return pct_fail(ctx, PCT_ERR_NO_NEIGHBORS, "plant the neighbour table first (pct_knn)");
    if (!sample_rows || !K_out || n_samples <= 0 || n_lo < 1 || n_hi < n_lo) return pct_fail(ctx, PCT_ERR_INVALID, "bad study arguments");
    if (n_hi > ctx->k) return pct_fail(ctx, PCT_ERR_INVALID, "the study needs %d neighbours per point, the table holds %d", n_hi, ctx->k);
    if (ctx->eps > 0) return pct_fail(ctx, PCT_ERR_INVALID, "the study needs a plain k-NN table (no eps bound)");
    const int nn = n_hi - n_lo + 1;
    const int64_t rows = n_samples * nn;
    if (rows > (int64_t)1 << 30) return pct_fail(ctx, PCT_ERR_INVALID, "study too large");
    // sample rows -> positions in the order the table is indexed by
    int* h_pos = (int*)malloc((size_t)n_samples * sizeof(int));
    if (!h_pos) return pct_fail(ctx, PCT_ERR_OOM, "host allocation failed");
    for (int64_t i = 0; i < n_samples; ++i) {
        if (sample_rows[i] < ctx->q_begin || sample_rows[i] >= ctx->q_end) {
            free(h_pos);
            return pct_fail(ctx, PCT_ERR_INVALID, "sample row %lld outside the owned range", (long long)sample_rows[i]);
        }
        h_pos[i] = (int)(sample_rows[i] - ctx->q_begin);    // exhaustive table: that is the row; grid table: index into row_of
    }
    const int32_t pitch = (n_hi + 1 + 3) & ~3;
    int st = pct_reserve(ctx, &ctx->stage_c, (size_t)n_samples * sizeof(int));
    if (st == PCT_OK) st = pct_reserve(ctx, &ctx->stage_a, (size_t)rows * pitch * sizeof(int));
    if (st == PCT_OK) st = pct_reserve(ctx, &ctx->stage_b, (size_t)rows * (sizeof(int) + 9 * sizeof(float)));
    if (st == PCT_OK) st = pct_reserve(ctx, &ctx->stage_d, (size_t)rows * sizeof(int64_t));
    if (st != PCT_OK) { free(h_pos); return st; }
    hipError_t e = hipMemcpyAsync(ctx->stage_c.p, h_pos, (size_t)n_samples * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    free(h_pos);
    if (e != hipSuccess) return pct_fail(ctx, PCT_ERR_HIP, "sample upload failed: %s", hipGetErrorString(e));
    int* d_spos = (int*)ctx->stage_c.p;
    if (ctx->knn_sorted_space) {      // public index -> neighbour-table row, on the device
        PCT_TRY(pct_launch_gather_int(ctx, (const int*)ctx->row_of.p, d_spos, n_samples));
    }
    int* d_cnt = (int*)ctx->stage_b.p;
    float* d_out = (float*)(d_cnt + rows);            // coefs (rows,6), K, H, H2
    PCT_TRY(pct_launch_prefix_rows(ctx, d_spos, n_samples, n_lo, n_hi, (int*)ctx->stage_a.p, pitch, d_cnt,
                                   (int64_t*)ctx->stage_d.p));
    PCT_TRY(pct_launch_fit_rows(ctx, (const int*)ctx->stage_a.p, d_cnt, (const int64_t*)ctx->stage_d.p, rows, n_hi + 1, pitch,
                                d_out, d_out + rows * 6, d_out + rows * 7, d_out + rows * 8, ctx->knn_sorted_space));
    PCT_HIP(ctx, hipMemcpyAsync(K_out, d_out + rows * 6, (size_t)rows * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;
}

int pct_mesh_energies(pct_ctx* ctx, const double* vertices, int64_t n_vertices, const int32_t* triangles, int64_t n_triangles,
                      const void* gaussian, const void* mean, int32_t curvature_is_f64, double* out3) {
    PCT_TRY(begin_call(ctx));
    if (!vertices || !triangles || !gaussian || !mean || !out3 || n_vertices <= 0 || n_triangles < 0)
        return pct_fail(ctx, PCT_ERR_INVALID, "bad mesh arguments");
    out3[0] = out3[1] = out3[2] = 0.0;
    if (n_triangles == 0) return PCT_OK;                               // utils.py:719-721: zeros
    for (int64_t i = 0; i < 3 * n_triangles; ++i)
        if (triangles[i] < 0 || triangles[i] >= n_vertices)
            return pct_fail(ctx, PCT_ERR_INVALID, "triangle %lld refers to vertex %d outside [0,%lld)", (long long)(i / 3), triangles[i], (long long)n_vertices);
    const size_t esz = curvature_is_f64 ? sizeof(double) : sizeof(float);
    const int nblk = (int)((n_triangles + 255) / 256 < 1024 ? (n_triangles + 255) / 256 : 1024);
    PCT_TRY(pct_reserve(ctx, &ctx->stage_a, (size_t)n_vertices * 3 * sizeof(double)));
    PCT_TRY(pct_reserve(ctx, &ctx->stage_b, (size_t)n_triangles * 3 * sizeof(int)));
    PCT_TRY(pct_reserve(ctx, &ctx->stage_c, (size_t)n_vertices * 2 * esz));
    PCT_TRY(pct_reserve(ctx, &ctx->stage_d, ((size_t)nblk * 3 + 3) * sizeof(double)));
    char* d_kh = (char*)ctx->stage_c.p;
    PCT_HIP(ctx, hipMemcpyAsync(ctx->stage_a.p, vertices, (size_t)n_vertices * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PCT_HIP(ctx, hipMemcpyAsync(ctx->stage_b.p, triangles, (size_t)n_triangles * 3 * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    PCT_HIP(ctx, hipMemcpyAsync(d_kh, gaussian, (size_t)n_vertices * esz, hipMemcpyHostToDevice, ctx->stream));
    PCT_HIP(ctx, hipMemcpyAsync(d_kh + (size_t)n_vertices * esz, mean, (size_t)n_vertices * esz, hipMemcpyHostToDevice, ctx->stream));
    double* d_part = (double*)ctx->stage_d.p;
    PCT_TRY(pct_launch_mesh_energies(ctx, (const double*)ctx->stage_a.p, (const int*)ctx->stage_b.p, n_triangles, d_kh,
                                     d_kh + (size_t)n_vertices * esz, curvature_is_f64 != 0, d_part, nblk, d_part + (size_t)nblk * 3));
    PCT_HIP(ctx, hipMemcpyAsync(out3, d_part + (size_t)nblk * 3, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;
}

static int voxel_downsample(pct_ctx* ctx, const void* xyz, bool f64, int64_t n, double voxel_size, int64_t* indices, int64_t* count) {
    PCT_TRY(begin_call(ctx));
    if (!xyz || !indices || !count || n <= 0 || n >= 0x7F000000 || !(voxel_size > 0) || (!f64 && !((float)voxel_size > 0.f)))
        return pct_fail(ctx, PCT_ERR_INVALID, "bad down-sampling arguments");
    for (int64_t i = 0; i < 3 * n; ++i) {
        const double v = f64 ? ((const double*)xyz)[i] : (double)((const float*)xyz)[i];
        if (!isfinite(v) || fabs(v / voxel_size) >= 2147483000.0)
            return pct_fail(ctx, PCT_ERR_INVALID, "coordinate %lld does not map to an int32 voxel index", (long long)(i / 3));
    }
    const size_t bytes = (size_t)n * 3 * (f64 ? sizeof(double) : sizeof(float));
    PCT_TRY(pct_reserve(ctx, &ctx->stage_a, bytes));
    PCT_TRY(pct_reserve(ctx, &ctx->stage_b, (size_t)n * sizeof(int64_t)));
    PCT_HIP(ctx, hipMemcpyAsync(ctx->stage_a.p, xyz, bytes, hipMemcpyHostToDevice, ctx->stream));
    PCT_TRY(pct_voxel_downsample_device(ctx, ctx->stage_a.p, f64, n, voxel_size, (int64_t*)ctx->stage_b.p, count));
    if (*count > 0) PCT_HIP(ctx, hipMemcpyAsync(indices, ctx->stage_b.p, (size_t)*count * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;
}

int pct_voxel_downsample(pct_ctx* ctx, const double* xyz, int64_t n, double voxel_size, int64_t* indices, int64_t* count) {
    return voxel_downsample(ctx, xyz, true, n, voxel_size, indices, count);
}

int pct_voxel_downsample_f32(pct_ctx* ctx, const float* xyz, int64_t n, double voxel_size, int64_t* indices, int64_t* count) {
    return voxel_downsample(ctx, xyz, false, n, voxel_size, indices, count);
}

int pct_surface_variation(pct_ctx* ctx, int32_t k_total, float* out) {
    PCT_TRY(begin_call(ctx));
    PCT_TRY(refuse_in_slab_mode(ctx, "pct_surface_variation"));
    if (!out || k_total < 2) return pct_fail(ctx, PCT_ERR_INVALID, "bad surface-variation arguments");
    const int64_t rows = ctx->q_end - ctx->q_begin;
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool again = false;
        PCT_TRY(run_knn(ctx, k_total - 1, 0.0, PCT_KNN_AUTO));        // k-1 neighbours + the point itself (utils.py:812-814)
        PCT_TRY(pct_reserve(ctx, &ctx->K, (size_t)rows * sizeof(float)));
        PCT_TRY(pct_launch_surface_variation(ctx, (float*)ctx->K.p));
        PCT_HIP(ctx, hipMemcpyAsync(out, ctx->K.p, (size_t)rows * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        PCT_TRY(finish_knn_stats(ctx, &again));
        if (!again) break;
        ctx->no_cull = true;
        ctx->retries = 1;
        ctx->cull_box_valid = false;   // the cached box was too small for this cloud: measure it again next time
    }
    ctx->fit_valid = false;
    return PCT_OK;
}

int pct_timings_size(void) { return (int)sizeof(pct_timings); }

int pct_get_timings(const pct_ctx* ctx, pct_timings* out) {
    if (!ctx || !out) return PCT_ERR_INVALID;
    if (ctx->pending) {                    // (an asynchronous call: its times exist once its kernels have finished)
        pct_ctx* c = const_cast<pct_ctx*>(ctx);
        PCT_TRY(begin_call(c));
    }
    *out = ctx->tm;
    return PCT_OK;
}

int pct_get_timings_done(const pct_ctx* ctx, pct_timings* out) {
    if (!ctx || !out) return PCT_ERR_INVALID;
    *out = ctx->pending ? ctx->tm_done : ctx->tm;
    return PCT_OK;
}

int pct_device_alloc(pct_ctx* ctx, int64_t bytes, void** dev_ptr) {
    PCT_TRY(begin_call(ctx));
    if (!dev_ptr || bytes <= 0) return pct_fail(ctx, PCT_ERR_INVALID, "bad allocation request");
    hipError_t e = hipMalloc(dev_ptr, (size_t)bytes);
    if (e != hipSuccess) return pct_fail(ctx, PCT_ERR_OOM, "hipMalloc(%lld) failed: %s", (long long)bytes, hipGetErrorString(e));
    return PCT_OK;
}

int pct_device_free(pct_ctx* ctx, void* dev_ptr) {
    PCT_TRY(begin_call(ctx));
    if (dev_ptr) PCT_HIP(ctx, hipFree(dev_ptr));
    return PCT_OK;
}

int pct_device_upload(pct_ctx* ctx, void* dev_dst, const void* host_src, int64_t bytes) {
    PCT_TRY(begin_call(ctx));
    PCT_HIP(ctx, hipMemcpyAsync(dev_dst, host_src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;
}

int pct_device_download(pct_ctx* ctx, void* host_dst, const void* dev_src, int64_t bytes) {
    PCT_TRY(begin_call(ctx));
    PCT_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;
}

int pct_synchronize(pct_ctx* ctx) {
    PCT_TRY(begin_call(ctx));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;
}

int pct_selftest(pct_ctx* ctx, int32_t* failures) {
    PCT_TRY(begin_call(ctx));
    if (!failures) return pct_fail(ctx, PCT_ERR_INVALID, "null output");
    PCT_TRY(pct_reserve(ctx, &ctx->red, 64));
    PCT_HIP(ctx, hipMemsetAsync(ctx->red.p, 0, 64, ctx->stream));
    PCT_TRY(pct_launch_selftest(ctx, (int*)ctx->red.p));
    PCT_HIP(ctx, hipMemcpyAsync(failures, ctx->red.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PCT_OK;
}

}  // extern "C"
