// Engine lifetime, error text, IQ ring in HBM, per-kernel hipEvent timing.
#include "engine_internal.h"
#include <sched.h>
#include <cstdint>
#include <cerrno>
#include <cctype>
#include "build_id.h"

#include <cstring>

static thread_local std::string g_last_error = "";

void sdr_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int sdr_fail(int status, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}

int sdr_set_device_keep(sdr_engine* e) {
    if (!e) return sdr_fail(SDR_ERR_INVALID, "engine is NULL");
    SDR_HIP(hipSetDevice(e->device));
    return SDR_OK;
}

// Every entry point comes through here: whatever it is about to do -- another launch, an allocation, a channel's state
// written from the host -- a resident tick server (track.hip) must not be in its way, nor keep its copy of what changes.
int sdr_set_device(sdr_engine* e) {
    if (int rc = sdr_set_device_keep(e)) return rc;
    e->srv_steady_ticks = 0;
    if (e->srv_running)
        if (int rc = sdr_tick_server_stop(e)) return rc;
    // (a slab staged for the next tick's launch to pull in -- "ingest_with_tick" -- goes into the ring now: whatever this call
    // is, it may read the ring or write to it)
    e->last_tick_took_slab = false;
    if (e->srv_slab_pending)
        if (int rc = sdr_iq_flush_server_slab(e)) return rc;
    return SDR_OK;
}

int sdr_devbuf_reserve(sdr_engine* e, DevBuf* b, size_t bytes) { return sdr_devbuf_reserve_on(e, e->stream, b, bytes); }

int sdr_pinned_reserve(sdr_engine* e, StreamCtx* ctx, size_t bytes) {
    if (bytes <= ctx->pinned_bytes && ctx->pinned) return SDR_OK;
    if (ctx->pinned) {
        SDR_HIP(hipStreamSynchronize(ctx->stream));
        SDR_HIP(hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
    }
    const size_t want = bytes < 65536 ? 65536 : bytes * 2;
    hipError_t err = hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault);
    if (err != hipSuccess) {
        ctx->pinned = nullptr;
        return sdr_fail(SDR_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(err));
    }
    ctx->pinned_bytes = want;
    (void)e;
    return SDR_OK;
}

StreamCtx* sdr_stream_ctx(sdr_engine* e, int stream_id) {
    if (stream_id == 0) return &e->ctx0;
    if (stream_id < 0 || stream_id > (int)e->streams.size()) return nullptr;
    return e->streams[(size_t)stream_id - 1];
}

int sdr_devbuf_reserve_on(sdr_engine* e, hipStream_t stream, DevBuf* b, size_t bytes) {
    (void)e;
    if (bytes <= b->bytes && b->ptr) return SDR_OK;
    if (b->ptr) {
        SDR_HIP(hipStreamSynchronize(stream));
        SDR_HIP(hipFree(b->ptr));
        b->ptr = nullptr;
        b->bytes = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t err = hipMalloc(&b->ptr, want);
    if (err != hipSuccess) {
        b->ptr = nullptr;
        return sdr_fail(SDR_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(err));
    }
    b->bytes = want;
    return SDR_OK;
}

ProfScope::ProfScope(sdr_engine* eng, const char* name, hipStream_t on)
    : e(eng), active(eng->prof), stream(on ? on : eng->stream) {
    rec.name = name;
    rec.start = rec.stop = nullptr;
    // sdr_prof_enable(e, 2): whole calls only ("call_*" scopes: one event pair around everything a call launches) -- the
    // per-stage scopes inside a call each cost the stream a few microseconds, which a sum over them would count as kernel time
    if (active && e->prof_calls_only != (strncmp(name, "call_", 5) == 0)) active = false;
    if (!active) return;
    for (int i = 0; i < 2; ++i) {
        hipEvent_t ev = nullptr;
        if (!e->prof_pool.empty()) {
            ev = e->prof_pool.back();
            e->prof_pool.pop_back();
        } else if (hipEventCreate(&ev) != hipSuccess) {
            active = false;
            return;
        }
        (i == 0 ? rec.start : rec.stop) = ev;
    }
    (void)hipEventRecord(rec.start, stream);
}

ProfScope::~ProfScope() {
    if (!active) return;
    (void)hipEventRecord(rec.stop, stream);
    e->prof_records.push_back(rec);
}

// Stream copy: 16 bytes per lane, four loads in flight per lane, grid-stride.
__global__ __launch_bounds__(256) void hbm_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a;
        dst[i + stride] = b;
        dst[i + 2 * stride] = c;
        dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

void sdr_iq_mark_written(sdr_engine* e, int64_t ring_offset, int64_t n_samples) {
    // (round 5 kept a sign-flipped IMAGE of a ci8 ring and refreshed it from the range marked here -- the refresh's event was
    // also what ordered readers on other streams behind an upload; the ring itself holds that form now, the ordering stays)
    (void)ring_offset, (void)n_samples;
    ++e->iq_write_seq;
}

int sdr_iq_order_reader(sdr_engine* e, StreamCtx* ctx) {
    if (!ctx || ctx->stream == e->stream || ctx->iq_waited_seq == e->iq_write_seq) return SDR_OK;
    if (e->iq_recorded_seq != e->iq_write_seq) {
        if (!e->iq_written) SDR_HIP(hipEventCreateWithFlags(&e->iq_written, hipEventDisableTiming));
        SDR_HIP(hipEventRecord(e->iq_written, e->stream));
        e->iq_recorded_seq = e->iq_write_seq;
    }
    SDR_HIP(hipStreamWaitEvent(ctx->stream, e->iq_written, 0));
    ctx->iq_waited_seq = e->iq_recorded_seq;
    return SDR_OK;
}

// A ci8 ring holds its samples with the sign bit of every byte flipped (correlator.h kCi8Flip): bytes [lo, hi) of the ring,
// just copied in as the host has them, flipped in place -- whole 16-byte granules inside the range, byte by byte at its ends.
__global__ __launch_bounds__(256) void flip_range_kernel(uint4* __restrict__ ring, size_t lo, size_t hi) {
    const size_t g0 = lo / 16, g1 = (hi + 15) / 16;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = g0 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < g1; g += stride) {
        uint4 v = ring[g];
        const size_t b = g * 16;
        if (b >= lo && b + 16 <= hi) {
            v.x ^= 0x80808080u, v.y ^= 0x80808080u, v.z ^= 0x80808080u, v.w ^= 0x80808080u;
        } else {
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (int i = 0; i < 16; ++i)
                if (b + (size_t)i >= lo && b + (size_t)i < hi) w[i >> 2] ^= 0x80u << (8 * (i & 3));
            v.x = w[0], v.y = w[1], v.z = w[2], v.w = w[3];
        }
        ring[g] = v;
    }
}

// sdr_iq_upload_async: 16-byte granules of a page-locked slab into the ring at granule `first` (modulo the ring); flip:
// what every dword is xor-ed with on the way (ci8: the sign bits).
__global__ __launch_bounds__(256) void ingest_kernel(const uint4* __restrict__ src, uint4* __restrict__ ring, size_t n16, size_t first,
                                                     size_t ring16, uint32_t flip) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        size_t d = first + i;
        if (d >= ring16) d -= ring16;
        uint4 v = src[i];
        v.x ^= flip, v.y ^= flip, v.z ^= flip, v.w ^= flip;
        ring[d] = v;
    }
}

static inline uint32_t ring_flip_mask(const sdr_engine* e) { return e->iq_fmt == SDR_FMT_CI8 ? 0x80808080u : 0u; }

int sdr_iq_flipped(sdr_engine* e, hipStream_t stream, const void** out) {
    // (round 6: a ci8 ring IS the sign-flipped form -- flipped where the samples enter; no second image, no refresh pass)
    (void)stream;
    if (e->iq_fmt != SDR_FMT_CI8 || !e->iq) return sdr_fail(SDR_ERR_STATE, "the sign-flipped sample form exists for ci8 rings only");
    *out = e->iq;
    return SDR_OK;
}


// The calling thread onto the CPUs next to the engine's GPU (its PCI function's local_cpulist in sysfs).  A receiver tick
// is a few round trips over the link through page-locked words: from the other socket each of them crosses the sockets'
// interconnect as well -- measured on a two-socket host, examples/receiver_loop.c with the resident tick server: 21.0-21.6 us per
// tick from the GPU's own node, 26.3-27.5 from the other.  Page-locked memory the engine allocates afterwards lands on that
// node too (first touch by this thread).  value = 0 restores the mask the thread had.  The option is a property of the calling
// THREAD (one saved mask per thread, whichever engine it was set through); processes the thread starts afterwards inherit
// the narrowed mask.
static int bind_thread_to_device(sdr_engine* e, int value) {
    static thread_local cpu_set_t saved;
    static thread_local bool have_saved = false;
    if (!value) {
        if (have_saved && sched_setaffinity(0, sizeof(saved), &saved) != 0)
            return sdr_fail(SDR_ERR_UNSUPPORTED, "sched_setaffinity: %s", strerror(errno));
        have_saved = false;
        return SDR_OK;
    }
    char bdf[64] = {0};
    SDR_HIP(hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf) - 1, e->device));
    for (char* c = bdf; *c; ++c) *c = (char)tolower((unsigned char)*c);
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
    FILE* f = fopen(path, "r");
    if (!f) return sdr_fail(SDR_ERR_UNSUPPORTED, "%s: %s", path, strerror(errno));
    char list[1024] = {0};
    const bool got = fgets(list, sizeof(list), f) != nullptr;
    fclose(f);
    cpu_set_t now, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(now), &now) != 0) return sdr_fail(SDR_ERR_UNSUPPORTED, "sched_getaffinity: %s", strerror(errno));
    // (per THREAD, not per engine: a thread that drives several engines and binds to a second GPU is moved from the mask it
    // had before the first binding, not from the first GPU's CPUs -- those of another socket would leave nothing)
    if (have_saved) now = saved;
    int n = 0;
    for (char* p = list; got && *p && *p != '\n';) {            // "0-63,128-191"
        char* end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (c >= 0 && CPU_ISSET((int)c, &now)) CPU_SET((int)c, &want), ++n;   // (never outside what the thread was allowed)
        if (*p == ',') ++p;
    }
    if (!n) return sdr_fail(SDR_ERR_UNSUPPORTED, "%s names no CPU this thread may run on ('%s')", path, list);
    if (!have_saved) saved = now, have_saved = true;
    if (sched_setaffinity(0, sizeof(want), &want) != 0) return sdr_fail(SDR_ERR_UNSUPPORTED, "sched_setaffinity: %s", strerror(errno));
    return SDR_OK;
}

extern "C" {

int sdr_set_option(sdr_engine* e, const char* name, int value) {
    if (!e || !name) return sdr_fail(SDR_ERR_INVALID, "NULL engine or option name");
    if (!strcmp(name, "pcps_materialise_map")) e->pcps_force_map = value != 0;
    else if (!strcmp(name, "pcps_radix_passes")) e->pcps_force_passes = value != 0;
    else if (!strcmp(name, "pcps_general_kernels")) e->pcps_no_fast = value != 0;
    else if (!strcmp(name, "pcps_one_stream")) e->pcps_no_overlap = value != 0;
    else if (!strcmp(name, "pcps_fused")) e->pcps_fused = value != 0;
    else if (!strcmp(name, "pcps_general_second_sweep")) e->pcps_slow_second = value != 0;
    else if (!strcmp(name, "pcps_no_spectra_cache")) e->pcps_no_spec_cache = value != 0;
    else if (!strcmp(name, "ingest_by_copy_command")) e->ingest_by_copy = value != 0;
    else if (!strcmp(name, "track_one_launch_tick")) e->track_one_launch_tick = value != 0;
    else if (!strcmp(name, "track_two_launch_tick")) e->track_two_launch_tick = value != 0;
    else if (!strcmp(name, "ingest_with_tick")) e->ingest_with_tick = value != 0;
    else if (!strcmp(name, "pcps_no_shared_spectra")) e->pcps_no_shared_spectra = value != 0;
    else if (!strcmp(name, "tick_server")) {
        // (stopping a server may queue its unpulled slab on this engine's stream: with several engines in one process the
        // current device may be another engine's)
        if (int rc = sdr_set_device_keep(e)) return rc;
        if (e->srv_running) (void)sdr_tick_server_stop(e);
        e->tick_server_opt = value != 0;
    }
    else if (!strcmp(name, "bind_thread_to_device")) return bind_thread_to_device(e, value);
    else if (!strcmp(name, "pcps_prn_chunk")) e->pcps_prn_chunk = value;
    else if (!strcmp(name, "epl_no_chip_variant")) e->epl_no_chip = value != 0;
    else if (!strcmp(name, "epl_no_split_variant")) e->epl_no_split = value != 0;
    else if (!strcmp(name, "epl_no_half_chip_view")) e->epl_no_double = value != 0;
    else if (!strcmp(name, "epl_no_two_chip_variant")) e->epl_no_chip2 = value != 0;
    else return sdr_fail(SDR_ERR_INVALID, "unknown option '%s'", name);
    return SDR_OK;
}

int sdr_hbm_copy_rate(sdr_engine* e, int64_t n_bytes, int reps, double* gbps) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!gbps || n_bytes < (1 << 20) || reps < 1) return sdr_fail(SDR_ERR_INVALID, "bad copy-rate request");
    const size_t n16 = (size_t)n_bytes / 16;
    void *a = nullptr, *b = nullptr;
    hipError_t err = hipMalloc(&a, n16 * 16);
    if (err == hipSuccess) err = hipMalloc(&b, n16 * 16);
    if (err != hipSuccess) {
        if (a) (void)hipFree(a);
        return sdr_fail(SDR_ERR_NOMEM, "hipMalloc for the copy test failed: %s", hipGetErrorString(err));
    }
    hipEvent_t t0, t1;
    SDR_HIP(hipEventCreate(&t0));
    SDR_HIP(hipEventCreate(&t1));
    SDR_HIP(hipMemsetAsync(a, 1, n16 * 16, e->stream));
    const int blocks = e->n_cus * 8;
    hipLaunchKernelGGL(hbm_copy_kernel, dim3(blocks), dim3(256), 0, e->stream, (const uint4*)a, (uint4*)b, n16);  // warm
    SDR_HIP(hipEventRecord(t0, e->stream));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(hbm_copy_kernel, dim3(blocks), dim3(256), 0, e->stream, (const uint4*)a, (uint4*)b, n16);
    SDR_HIP(hipEventRecord(t1, e->stream));
    SDR_HIP(hipStreamSynchronize(e->stream));
    float ms = 0.f;
    SDR_HIP(hipEventElapsedTime(&ms, t0, t1));
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    (void)hipFree(a);
    (void)hipFree(b);
    *gbps = 2.0 * (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9;
    return SDR_OK;
}

const char* sdr_last_error(void) { return g_last_error.c_str(); }

int sdr_abi_version(void) { return SDR_ABI_VERSION; }
const char* sdr_build_id(void) { return SDR_BUILD_ID; }

int sdr_device_count(int* n) {
    if (!n) return sdr_fail(SDR_ERR_INVALID, "n is NULL");
    int c = 0;
    hipError_t err = hipGetDeviceCount(&c);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        c = 0;  // CPU-only host: not an error for this query
    }
    *n = c;
    return SDR_OK;
}

int sdr_engine_create(int device_id, sdr_engine** out) {
    if (!out) return sdr_fail(SDR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int c = 0;
    hipError_t err = hipGetDeviceCount(&c);
    if (err != hipSuccess || c == 0) {
        (void)hipGetLastError();
        return sdr_fail(SDR_ERR_HIP, "no HIP device visible (%s): this engine has no CPU fallback",
                        err == hipSuccess ? "count=0" : hipGetErrorString(err));
    }
    if (device_id < 0 || device_id >= c)
        return sdr_fail(SDR_ERR_INVALID, "device_id %d outside [0,%d)", device_id, c);
    SDR_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    SDR_HIP(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return sdr_fail(SDR_ERR_UNSUPPORTED, "device %d is %s; this library only carries gfx950 code",
                        device_id, prop.gcnArchName);
    sdr_engine* e = new (std::nothrow) sdr_engine();
    if (!e) return sdr_fail(SDR_ERR_NOMEM, "host allocation failed");
    e->device = device_id;
    e->n_cus = prop.multiProcessorCount;
    err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (err != hipSuccess) {
        delete e;
        return sdr_fail(SDR_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(err));
    }
    e->ctx0.stream = e->stream;
    *out = e;
    return SDR_OK;
}

static void free_ctx(StreamCtx* c) {
    for (DevBuf* b : {&c->traj, &c->bits, &c->xchg})
        if (b->ptr) (void)hipFree(b->ptr);
    if (c->pinned) (void)hipHostFree(c->pinned);
}

void sdr_engine_destroy(sdr_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    sdr_tick_server_free(e);            // (a resident tick server leaves first: the synchronisation below would wait for it)
    (void)hipDeviceSynchronize();
    free_ctx(&e->ctx0);
    for (StreamCtx* c : e->streams) {
        free_ctx(c);
        (void)hipStreamDestroy(c->stream);
        delete c;
    }
    for (auto& r : e->prof_records) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    for (auto ev : e->prof_pool) (void)hipEventDestroy(ev);
    DevBuf* bufs[] = {&e->ws_items,  &e->ws_out,   &e->ws_spacing, &e->ws_setups, &e->ws_stats, &e->pcps_fwd,   &e->pcps_a,
                      &e->pcps_b,    &e->pcps_code, &e->pcps_code2, &e->pcps_tickets, &e->pcps_spec_off, &e->pcps_tw,   &e->pcps_map,   &e->pcps_csum,
                      &e->pcps_part, &e->pcps_res,  &e->track_state, &e->track_cfg,
                      &e->pcps_blu,  &e->pcps_blu_x, &e->pcps_blu_a, &e->pcps_blu_b, &e->pcps_work, &e->pcps_theta};
    for (DevBuf* b : bufs)
        if (b->ptr) (void)hipFree(b->ptr);
    for (DevBuf& b : e->plan_pool)
        if (b.ptr) (void)hipFree(b.ptr);
    if (e->slab_pinned) (void)hipHostFree(e->slab_pinned);
    for (int h = 0; h < 2; ++h)
        if (e->slab_done[h]) (void)hipEventDestroy(e->slab_done[h]);
    if (e->iq_written) (void)hipEventDestroy(e->iq_written);
    if (e->iq) (void)hipFree(e->iq);
    if (e->codes) (void)hipFree(e->codes);
    if (e->luts) (void)hipFree(e->luts);
    if (e->luts2) (void)hipFree(e->luts2);
    if (e->pcps_aux) (void)hipStreamDestroy(e->pcps_aux);
    for (hipEvent_t ev : e->pcps_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (e->code_len) (void)hipFree(e->code_len);
    (void)hipStreamDestroy(e->stream);
    delete e;
}

int sdr_engine_sync(sdr_engine* e) {
    if (int rc = sdr_set_device(e)) return rc;
    SDR_HIP(hipStreamSynchronize(e->stream));
    return SDR_OK;
}

int sdr_stream_create(sdr_engine* e, int* stream_id) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!stream_id) return sdr_fail(SDR_ERR_INVALID, "stream_id is NULL");
    StreamCtx* c = new (std::nothrow) StreamCtx();
    if (!c) return sdr_fail(SDR_ERR_NOMEM, "host allocation failed");
    hipError_t err = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (err != hipSuccess) {
        delete c;
        return sdr_fail(SDR_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(err));
    }
    e->streams.push_back(c);
    *stream_id = (int)e->streams.size();
    return SDR_OK;
}

int sdr_stream_sync(sdr_engine* e, int stream_id) {
    if (int rc = sdr_set_device(e)) return rc;
    StreamCtx* c = sdr_stream_ctx(e, stream_id);
    if (!c) return sdr_fail(SDR_ERR_INVALID, "stream id %d does not exist", stream_id);
    SDR_HIP(hipStreamSynchronize(c->stream));
    return SDR_OK;
}

int sdr_prof_enable(sdr_engine* e, int enable) {
    if (!e) return sdr_fail(SDR_ERR_INVALID, "engine is NULL");
    e->prof = enable != 0;
    e->prof_calls_only = enable == 2;
    return SDR_OK;
}

int sdr_prof_reset(sdr_engine* e) {
    if (int rc = sdr_set_device(e)) return rc;
    SDR_HIP(hipStreamSynchronize(e->stream));
    for (auto& r : e->prof_records) {
        e->prof_pool.push_back(r.start);
        e->prof_pool.push_back(r.stop);
    }
    e->prof_records.clear();
    return SDR_OK;
}

int sdr_prof_read(sdr_engine* e, const char* prefix, double* total_ms, int64_t* launches) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!total_ms || !launches) return sdr_fail(SDR_ERR_INVALID, "NULL output");
    SDR_HIP(hipStreamSynchronize(e->stream));
    size_t plen = prefix ? strlen(prefix) : 0;
    double tot = 0.0;
    int64_t cnt = 0;
    for (auto& r : e->prof_records) {
        if (plen && strncmp(r.name, prefix, plen) != 0) continue;
        float ms = 0.f;
        SDR_HIP(hipEventElapsedTime(&ms, r.start, r.stop));
        tot += ms;
        ++cnt;
    }
    *total_ms = tot;
    *launches = cnt;
    return SDR_OK;
}

/* ------------------------------------------------------------------ IQ ring */

int sdr_iq_alloc(sdr_engine* e, int64_t capacity_samples, int fmt) {
    if (int rc = sdr_set_device(e)) return rc;
    if (fmt < SDR_FMT_CI8 || fmt > SDR_FMT_CF64) return sdr_fail(SDR_ERR_INVALID, "bad IQ format %d", fmt);
    if (capacity_samples <= 0 || capacity_samples % 8 != 0)
        return sdr_fail(SDR_ERR_INVALID, "ring capacity %lld must be a positive multiple of 8 samples",
                        (long long)capacity_samples);
    SDR_HIP(hipStreamSynchronize(e->stream));
    for (DevBuf& b : e->plan_pool)      // (plans of the old ring are stale: what they left behind makes room for the new one)
        if (b.ptr) (void)hipFree(b.ptr);
    e->plan_pool.clear();
    if (e->iq) {
        SDR_HIP(hipFree(e->iq));
        e->iq = nullptr;
        e->iq_capacity = 0;
    }
    // (256 bytes of slack behind the ring: the chip-aligned correlator loads whole 56-byte windows, whose tail
    // may reach past the last sample of the last epoch)
    size_t bytes = (size_t)capacity_samples * sdr_fmt_bytes(fmt) + 256;
    hipError_t err = hipMalloc(&e->iq, bytes);
    if (err != hipSuccess) {
        e->iq = nullptr;
        return sdr_fail(SDR_ERR_NOMEM, "hipMalloc(%zu) for the IQ ring failed: %s", bytes,
                        hipGetErrorString(err));
    }
    SDR_HIP(hipMemsetAsync(e->iq, fmt == SDR_FMT_CI8 ? 0x80 : 0, bytes, e->stream));     // (zero samples: ci8 bytes are kept sign-flipped)
    e->iq_capacity = capacity_samples;
    e->iq_fmt = fmt;
    return SDR_OK;
}

static int iq_copy(sdr_engine* e, void* host, int64_t n, int64_t off, bool upload, bool wait = true) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!e->iq) return sdr_fail(SDR_ERR_STATE, "IQ ring not allocated");
    if (!host && n > 0) return sdr_fail(SDR_ERR_INVALID, "host pointer is NULL");
    if (n < 0 || n > e->iq_capacity)
        return sdr_fail(SDR_ERR_RANGE, "n_samples %lld exceeds ring capacity %lld", (long long)n,
                        (long long)e->iq_capacity);
    if (off < 0) return sdr_fail(SDR_ERR_RANGE, "negative ring offset");
    off %= e->iq_capacity;
    size_t sb = sdr_fmt_bytes(e->iq_fmt);
    int64_t first = n < e->iq_capacity - off ? n : e->iq_capacity - off;
    char* dev = (char*)e->iq;
    char* h = (char*)host;
    const bool ci8 = e->iq_fmt == SDR_FMT_CI8;
    auto flip = [&](size_t lo, size_t hi) {      // (ci8: the bytes just copied in, into the ring's sign-flipped form)
        const size_t n16 = (hi + 15) / 16 - lo / 16;
        const unsigned blocks = (unsigned)(n16 / 256 / 4 + 1 < 8192 ? n16 / 256 / 4 + 1 : 8192);
        hipLaunchKernelGGL(flip_range_kernel, dim3(blocks), dim3(256), 0, e->stream, (uint4*)e->iq, lo, hi);
    };
    if (upload) {
        sdr_iq_mark_written(e, off, n);
        if (first) {
            SDR_HIP(hipMemcpyAsync(dev + off * sb, h, first * sb, hipMemcpyHostToDevice, e->stream));
            if (ci8) flip((size_t)off * sb, (size_t)(off + first) * sb);
        }
        if (n > first) {
            SDR_HIP(hipMemcpyAsync(dev, h + first * sb, (n - first) * sb, hipMemcpyHostToDevice, e->stream));
            if (ci8) flip(0, (size_t)(n - first) * sb);
        }
        if (ci8 && n > 0) SDR_HIP(hipGetLastError());
    } else {
        if (first) SDR_HIP(hipMemcpyAsync(h, dev + off * sb, first * sb, hipMemcpyDeviceToHost, e->stream));
        if (n > first)
            SDR_HIP(hipMemcpyAsync(h + first * sb, dev, (n - first) * sb, hipMemcpyDeviceToHost, e->stream));
    }
    // The caller owns `host`; do not keep it past return (wait == false: the caller of this file synchronises).
    if (wait || (!upload && ci8)) SDR_HIP(hipStreamSynchronize(e->stream));
    if (!upload && ci8) {                        // (what leaves the ring leaves it as the host wrote it)
        unsigned char* b = (unsigned char*)host;
        const size_t nb = (size_t)n * sb;
        size_t i = 0;
        for (; i + 8 <= nb; i += 8) {
            uint64_t w;
            memcpy(&w, b + i, 8);
            w ^= 0x8080808080808080ull;
            memcpy(b + i, &w, 8);
        }
        for (; i < nb; ++i) b[i] ^= 0x80;
    }
    return SDR_OK;
}


int sdr_iq_upload(sdr_engine* e, const void* iq, int64_t n_samples, int64_t ring_offset) {
    return iq_copy(e, const_cast<void*>(iq), n_samples, ring_offset, true);
}

int sdr_iq_download(sdr_engine* e, void* iq, int64_t n_samples, int64_t ring_offset) {
    return iq_copy(e, iq, n_samples, ring_offset, false);
}

int sdr_iq_upload_queue(sdr_engine* e, const void* iq, int64_t n_samples, int64_t ring_offset) {
    if (!e) return sdr_fail(SDR_ERR_INVALID, "null engine");
    return iq_copy(e, const_cast<void*>(iq), n_samples, ring_offset, true, false);
}

int sdr_host_alloc(sdr_engine* e, size_t bytes, void** out) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!out || bytes == 0) return sdr_fail(SDR_ERR_INVALID, "sdr_host_alloc: NULL result pointer or zero bytes");
    *out = nullptr;
    hipError_t err = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (err != hipSuccess) {
        *out = nullptr;
        return sdr_fail(SDR_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
    }
    e->host_blocks.emplace_back((const char*)*out, bytes);
    return SDR_OK;
}

int sdr_host_free(sdr_engine* e, void* block) {
    if (int rc = sdr_set_device(e)) return rc;
    if (!block) return SDR_OK;
    SDR_HIP(hipStreamSynchronize(e->stream));     // (a queued upload may still read it)
    for (size_t k = 0; k < e->host_blocks.size(); ++k)
        if (e->host_blocks[k].first == (const char*)block) {
            e->host_blocks.erase(e->host_blocks.begin() + (long)k);
            break;
        }
    SDR_HIP(hipHostFree(block));
    return SDR_OK;
}

}  // extern "C"

int sdr_iq_flush_server_slab(sdr_engine* e) {
    if (!e->srv_slab_pending) return SDR_OK;
    e->srv_slab_pending = false;
    const size_t sb = sdr_fmt_bytes(e->iq_fmt);
    const size_t bytes = (size_t)e->srv_slab_n * sb, off_b = (size_t)e->srv_slab_off * sb, cap_b = (size_t)e->iq_capacity * sb;
    const int half = e->srv_slab_half;
    const char* stage = (const char*)e->srv_slab_src;
    const size_t n16 = bytes / 16;
    const unsigned blocks = (unsigned)((n16 + 255) / 256 < 64 ? (n16 + 255) / 256 : 64);
    hipLaunchKernelGGL(ingest_kernel, dim3(blocks), dim3(256), 0, e->stream, (const uint4*)stage, (uint4*)e->iq, n16, off_b / 16, cap_b / 16, ring_flip_mask(e));
    SDR_HIP(hipGetLastError());
    if (half >= 0) {      // (a staging half: busy until the kernel has read it; the caller's own block is the caller's to keep)
        if (!e->slab_done[half]) SDR_HIP(hipEventCreateWithFlags(&e->slab_done[half], hipEventDisableTiming));
        SDR_HIP(hipEventRecord(e->slab_done[half], e->stream));
        e->slab_busy[half] = true;
    } else {
        // the caller's own page-locked block, read in place: the header promises it back when the next tick has returned, and
        // a tick in which no channel is ready ends without a synchronisation of its own -- sdr_bank_tick_mirrored_end waits
        // for this flag's kernel (tick_end_wait_inplace_slab)
        e->inplace_slab_in_flight = true;
    }
    return SDR_OK;
}

// Asynchronous upload of a caller-owned (pageable) slab: small slabs -- a receiver tick brings 1 ms, 50 KB at 25 MHz --
// go through a page-locked staging buffer of the engine (one memcpy here, then a DMA the stream does not wait for the
// host on); the caller's pointer is not kept past return either way.
int sdr_iq_upload_async(sdr_engine* e, const void* iq, int64_t n_samples, int64_t ring_offset) {
    const size_t bytes = n_samples > 0 ? (size_t)n_samples * sdr_fmt_bytes(e->iq_fmt) : 0;
    if (!iq || bytes == 0 || bytes > (1u << 20)) return iq_copy(e, const_cast<void*>(iq), n_samples, ring_offset, true, false);
    if (int rc = sdr_set_device_keep(e)) return rc;           // (a resident tick server stays: it may be the one to pull this slab)
    if (e->srv_slab_pending)                                   // (a slab still waiting for a request: into the ring the ordinary way first)
        if (int rc = sdr_iq_flush_server_slab(e)) return rc;
    // A slab that already lies in page-locked memory this engine handed out (sdr_host_alloc), on a 16-byte boundary, is read IN
    // PLACE by whoever pulls it into the ring -- the ingest workgroups of the next tick's launch, the tick server's doormen, an
    // ingest kernel queued here -- instead of being copied into a staging half first (2 us of a 50 KB slab): the caller leaves
    // it unchanged until the next tick of the engine (or sdr_engine_sync) has returned.
    if (e->iq && ring_offset >= 0 && n_samples <= e->iq_capacity && !e->ingest_by_copy && (uintptr_t)iq % 16 == 0 && bytes % 16 == 0) {
        bool ours = false;
        for (const auto& blk : e->host_blocks)
            ours = ours || ((const char*)iq >= blk.first && (const char*)iq + bytes <= blk.first + blk.second);
        const size_t sb = sdr_fmt_bytes(e->iq_fmt);
        const int64_t off = ring_offset % e->iq_capacity;
        const size_t off_b = (size_t)off * sb, cap_b = (size_t)e->iq_capacity * sb;
        if (ours && off_b % 16 == 0 && cap_b % 16 == 0) {
            sdr_iq_mark_written(e, off, n_samples);
            if (e->srv_running || (e->ingest_with_tick && e->last_tick_took_slab)) {
                e->srv_slab_pending = true;
                e->srv_slab_half = -1;
                e->srv_slab_src = iq;
                e->srv_slab_off = off;
                e->srv_slab_n = n_samples;
                return SDR_OK;
            }
            const size_t n16 = bytes / 16;
            const unsigned blocks = (unsigned)((n16 + 255) / 256 < 64 ? (n16 + 255) / 256 : 64);
            hipLaunchKernelGGL(ingest_kernel, dim3(blocks), dim3(256), 0, e->stream, (const uint4*)iq, (uint4*)e->iq, n16, off_b / 16, cap_b / 16, ring_flip_mask(e));
            SDR_HIP(hipGetLastError());
            e->inplace_slab_in_flight = true;      // (sdr_bank_tick_mirrored_end waits for it where nothing else would)
            return SDR_OK;
        }
    }
    if (bytes > e->slab_bytes) {
        SDR_HIP(hipStreamSynchronize(e->stream));   // (an earlier slab's DMA may still read the old buffer)
        if (e->slab_pinned) SDR_HIP(hipHostFree(e->slab_pinned));
        e->slab_pinned = nullptr;
        e->slab_bytes = 0;
        const size_t want = bytes < 131072 ? 131072 : bytes * 2;
        hipError_t err = hipHostMalloc(&e->slab_pinned, 2 * want, hipHostMallocDefault);   // two halves, used alternately
        if (err != hipSuccess) {
            e->slab_pinned = nullptr;
            return sdr_fail(SDR_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", 2 * want, hipGetErrorString(err));
        }
        e->slab_bytes = want;
        e->slab_busy[0] = e->slab_busy[1] = false;   // (the synchronisation above completed whatever read the old halves)
    }
    // Two halves, used alternately, each guarded by an event recorded behind the transfer that reads it: a tick normally
    // ends with a synchronisation of this stream, but a tick in which no channel is ready does not, and a C caller may
    // queue any number of slabs -- a half is written again only when the transfer out of it has completed (the wait is
    // free in the common case: the event completed with the tick before last).
    e->slab_flip ^= 1;
    const int half = e->slab_flip;
    if (e->slab_busy[half]) {
        SDR_HIP(hipEventSynchronize(e->slab_done[half]));
        e->slab_busy[half] = false;
    }
    if (!e->slab_done[half]) SDR_HIP(hipEventCreateWithFlags(&e->slab_done[half], hipEventDisableTiming));
    char* stage = (char*)e->slab_pinned + (half ? e->slab_bytes : 0);
    memcpy(stage, iq, bytes);
    // A KERNEL pulls the slab out of the page-locked buffer (16 bytes per lane over PCIe) instead of a copy command: the
    // tick's launch follows it on the same queue with nothing but the queue's own ordering in between, where a DMA
    // engine's copy puts a cross-queue signal in front of the kernel that needs the samples (measured on the tick: the
    // host's wait 28.3 -> see DESIGN.md).  Needs 16-byte granules; anything else takes the copy command.
    const size_t sb = sdr_fmt_bytes(e->iq_fmt);
    const int64_t cap = e->iq_capacity;
    if (e->iq && ring_offset >= 0 && n_samples <= cap && !e->ingest_by_copy) {
        const int64_t off = ring_offset % cap;
        const size_t off_b = (size_t)off * sb, cap_b = (size_t)cap * sb;
        if (off_b % 16 == 0 && bytes % 16 == 0 && cap_b % 16 == 0) {
            sdr_iq_mark_written(e, off, n_samples);
            if (e->srv_running || (e->ingest_with_tick && e->last_tick_took_slab)) {
                // the resident tick server's doormen pull it out of the staging half (track.hip): no launch here.  The half is
                // busy until the request that needs it has been answered (or the slab flushed the ordinary way).
                e->srv_slab_pending = true;
                e->srv_slab_half = half;
                e->srv_slab_src = stage;
                e->srv_slab_off = off;
                e->srv_slab_n = n_samples;
                return SDR_OK;
            }
            const size_t n16 = bytes / 16;
            const unsigned blocks = (unsigned)((n16 + 255) / 256 < 64 ? (n16 + 255) / 256 : 64);
            hipLaunchKernelGGL(ingest_kernel, dim3(blocks), dim3(256), 0, e->stream, (const uint4*)stage, (uint4*)e->iq, n16,
                               off_b / 16, cap_b / 16, ring_flip_mask(e));
            SDR_HIP(hipGetLastError());
            SDR_HIP(hipEventRecord(e->slab_done[half], e->stream));
            e->slab_busy[half] = true;
            return SDR_OK;
        }
    }
    if (int rc = iq_copy(e, stage, n_samples, ring_offset, true, false)) return rc;
    SDR_HIP(hipEventRecord(e->slab_done[half], e->stream));
    e->slab_busy[half] = true;
    return SDR_OK;
}
