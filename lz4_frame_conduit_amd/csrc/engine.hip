// engine.hip -- kernels (via the .cuh headers) + their launch plumbing + the device-pointer C ABI.
// Target: gfx950 only (MI355X).  No CPU fallback: without a usable device every entry fails loudly.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <thread>
#include <vector>

#include "engine.hpp"
#include "frame_dev.cuh"

using namespace lz4f;

static_assert(sizeof(BlockOut) == sizeof(lz4f_mi355x_block), "block table layout");
static_assert(sizeof(ResultRec) == sizeof(lz4f_mi355x_result), "result layout");
static_assert(sizeof(ChunkInfo) == 32, "chunk info layout");

namespace lz4f {

static thread_local char t_err[512] = "";
static thread_local int t_device = -1;

void set_last_error(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(t_err, sizeof(t_err), fmt, ap); va_end(ap);
}
const char* last_error() { return t_err; }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            set_last_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
            return make_err(LZ4F_ERROR_GENERIC);                                                        \
        }                                                                                               \
    } while (0)

int DevBuf::ensure(size_t n)
{
    if (n <= cap) return 0;
    size_t want = n + n / 8 + 4096;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { set_last_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); p = nullptr; return 1; }
    cap = want;
    return 0;
}
void DevBuf::release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
int PinBuf::ensure(size_t n)
{
    if (n <= cap) return 0;
    size_t want = n + n / 8 + 4096;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) { set_last_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); p = nullptr; return 1; }
    cap = want;
    return 0;
}
void PinBuf::release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }

int selected_device()
{
    if (t_device < 0) {
        const char* s = getenv("LZ4F_MI355X_DEVICE");
        t_device = s ? atoi(s) : 0;
    }
    return t_device;
}

uint32_t pick_chunk_size(uint32_t block_size)
{
    const uint32_t c = 64u << 10;    // pass E1's tile (E1_TILE): what one workgroup searches at a time, and the unit of passes S and E2
    return block_size < c ? block_size : c;
}

}  // namespace lz4f
void lz4f_mi355x_engine::Switches::read()
{
    auto on = [](const char* n) { return getenv(n) != nullptr; };
    no_index = on("LZ4F_MI355X_NO_INDEX"); no_selfindex = on("LZ4F_MI355X_NO_SELFINDEX"); no_resolve = on("LZ4F_MI355X_NO_RESOLVE");
    no_trace = on("LZ4F_MI355X_NO_TRACE"); no_doubling = on("LZ4F_MI355X_NO_DOUBLING"); trace_always = on("LZ4F_MI355X_TRACE_ALWAYS");
    no_groups = on("LZ4F_MI355X_NO_GROUPS"); no_window = on("LZ4F_MI355X_NO_WINDOW"); serial_walk = on("LZ4F_MI355X_SERIAL_WALK");
    no_trailer = on("LZ4F_MI355X_NO_TRAILER"); no_density_probe = on("LZ4F_MI355X_NO_DENSITY_PROBE"); no_spx = on("LZ4F_MI355X_NO_SPX"); no_overlap = on("LZ4F_MI355X_NO_OVERLAP"); no_content_check = on("LZ4F_MI355X_NO_CONTENT_CHECK"); prof = on("LZ4F_MI355X_PROF"); e1_sync = on("LZ4F_MI355X_E1_SYNC"); no_selffeed = on("LZ4F_MI355X_NO_SELFFEED"); dense_mode = 0; if (const char* v = getenv("LZ4F_MI355X_DENSE_MODE")) { const int g = atoi(v); if (g >= 0 && g <= 2) dense_mode = (unsigned)g; }
    group_kib = 0; if (const char* v = getenv("LZ4F_MI355X_GROUP_KIB")) { const int k = atoi(v); if (k >= 64 && k <= 4096 && (k & (k - 1)) == 0) group_kib = (unsigned)k; }
    feed_round = 0; if (const char* v = getenv("LZ4F_MI355X_FEED_ROUND")) { const int k = atoi(v); if (k >= 17 && k <= 4096) feed_round = (unsigned)k; }
    chain_gate = 0; if (const char* v = getenv("LZ4F_MI355X_CHAIN_GATE")) { const int g = atoi(v); if (g > 0 && g < (1 << 20)) chain_gate = g; }
    decode_mode = 0; if (const char* v = getenv("LZ4F_MI355X_DECODE")) decode_mode = v[0];
    e1_run = 0; if (const char* v = getenv("LZ4F_MI355X_E1_RUN")) { const int g = atoi(v); if (g >= 1 && g <= 4096) e1_run = (unsigned)g; }
    e1_solo = 0; if (const char* v = getenv("LZ4F_MI355X_E1_SOLO")) e1_solo = (unsigned)atoi(v);
    if (on("LZ4F_MI355X_DETERMINISTIC")) e1_solo |= 1u;              // equal input -> equal bytes: one wave per workgroup parses, in order (see lz4f_mi355x_engine_set_deterministic)
    wait_ticks = 0; if (const char* v = getenv("LZ4F_MI355X_WAIT_TICKS")) { unsigned long long a = 0; if (sscanf(v, "%llu", &a) == 1) wait_ticks = a; }
    dblk_lds = 0; if (const char* v = getenv("LZ4F_MI355X_DBLK_LDS")) { const int k = atoi(v); if (k > 0 && k <= 150) dblk_lds = (unsigned)k << 10; }      // (development: fewer wave-per-block decoders per CU)
    recs_per_tile = 0; if (const char* v = getenv("LZ4F_MI355X_RECS_PER_TILE")) { const int k = atoi(v); if (k >= 1 && k <= 16385) recs_per_tile = (unsigned)k; }
    seed = 2; if (const char* v = getenv("LZ4F_MI355X_SEED")) { unsigned a = 0; if (sscanf(v, "%u", &a) == 1 && a >= 1 && a <= 64) seed = a; }
}
namespace lz4f {
size_t new_engine(lz4f_mi355x_engine** out, int device, void* stream, bool borrow)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_last_error("no usable HIP device (hipGetDeviceCount: %s, count %d): liblz4f_mi355x has no CPU fallback", hipGetErrorString(e), n);
        return make_err(LZ4F_ERROR_GENERIC);
    }
    if (device < 0 || device >= n) { set_last_error("device %d out of range (%d devices)", device, n); return make_err(LZ4F_ERROR_GENERIC); }
    HIP_TRY(hipSetDevice(device));
    lz4f_mi355x_engine* en = new lz4f_mi355x_engine();
    en->device = device;
    en->sw.read();
    if (borrow) { en->stream = stream; en->own_stream = false; }
    else {
        hipStream_t s;
        hipError_t e2 = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e2 != hipSuccess) { delete en; set_last_error("hipStreamCreate failed: %s", hipGetErrorString(e2)); return make_err(LZ4F_ERROR_GENERIC); }
        en->stream = s; en->own_stream = true;
    }
    *out = en;
    return 0;
}

// ---- engines for the host-pointer entry points: a pool per process, not an engine per thread ----
// (A Haskell host's safe FFI calls land on arbitrary OS threads: engines owned by threads would be leaked with them - stream,
// pinned staging, device workspace.  A call borrows an idle engine of the wanted device, or makes one, and gives it back;
// lz4f_mi355x_release_engines() frees the idle ones, and so does the library's unload.)
namespace {
struct EnginePool {
    std::mutex mu;
    std::vector<lz4f_mi355x_engine*> idle;
    ~EnginePool() { drain(); }
    void drain()
    {
        std::vector<lz4f_mi355x_engine*> v;
        { std::lock_guard<std::mutex> g(mu); v.swap(idle); }
        for (auto* e : v) delete e;
    }
};
EnginePool& pool() { static EnginePool p; return p; }
}  // namespace

size_t acquire_engine(lz4f_mi355x_engine** out, int device)
{
    if (device < 0) device = selected_device();
    {
        EnginePool& p = pool();
        std::lock_guard<std::mutex> g(p.mu);
        for (size_t i = 0; i < p.idle.size(); i++)
            if (p.idle[i]->device == device) { *out = p.idle[i]; p.idle.erase(p.idle.begin() + i); return 0; }
    }
    return new_engine(out, device, nullptr, false);
}
void release_engine(lz4f_mi355x_engine* e)
{
    if (!e) return;
    EnginePool& p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    p.idle.push_back(e);
}
void release_idle_engines() { pool().drain(); }

}  // namespace lz4f

lz4f_mi355x_engine::~lz4f_mi355x_engine()
{
    (void)hipSetDevice(device);
    (void)hipStreamSynchronize((hipStream_t)stream);
    desc.release(); seqcnt.release(); spx.release(); selfix.release(); selfcnt.release(); postab.release(); pdbuf.release();
    info.release(); recs.release(); e1_scratch.release(); walkbuf.release(); density.release(); ixtmp.release(); table.release(); blk_bytes.release(); res.release(); bad.release();
    d_in.release(); d_out.release();
    h_in.release(); h_out.release(); h_small.release();
    for (int i = 0; i < 24; i++) if (ev[i]) (void)hipEventDestroy((hipEvent_t)ev[i]);
    if (aux_stream) { (void)hipStreamSynchronize((hipStream_t)aux_stream); (void)hipStreamDestroy((hipStream_t)aux_stream); }
    if (ev_fork) (void)hipEventDestroy((hipEvent_t)ev_fork);
    if (ev_join) (void)hipEventDestroy((hipEvent_t)ev_join);
    if (own_stream && stream) (void)hipStreamDestroy((hipStream_t)stream);
}

void lz4f_mi355x_engine::tick(int slot, bool end, void* on_stream)
{
    if (!timing) return;
    const int i = slot * 2 + (end ? 1 : 0);
    if (!ev[i]) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; ev[i] = e; }
    (void)hipEventRecord((hipEvent_t)ev[i], (hipStream_t)(on_stream ? on_stream : stream));
    if (end) ev_used[slot] = true;
}
bool lz4f_mi355x_engine::aux_ready()
{
    if (aux_stream && ev_fork && ev_join) return true;
    hipStream_t s; hipEvent_t a, b;
    if (!aux_stream) { if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return false; } aux_stream = s; }
    if (!ev_fork) { if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; } ev_fork = a; }
    if (!ev_join) { if (hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; } ev_join = b; }
    return true;
}

size_t lz4f_mi355x_engine::sync()
{
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    // (work forked onto the second stream that the main stream never joined - a call that returned an error behind the fork: a caller that
    // frees its buffers after sync() must not race it)
    if (aux_pending && aux_stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)aux_stream)); aux_pending = false; }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// developer aid: cycle counters of workgroup 0 (LZ4F_MI355X_PROF=1), read back with lz4f_mi355x_debug_prof
static unsigned long long* g_prof = nullptr;
static unsigned long long* prof_buf()
{
    if (!g_prof) { if (hipMalloc(&g_prof, 1024) != hipSuccess) return nullptr; (void)hipMemset(g_prof, 0, 1024); (void)hipMemset(g_prof + 70, 0xFF, 8); (void)hipMemset(g_prof + 75, 0xFF, 8); }
    return g_prof;
}
extern "C" __attribute__((visibility("default"))) int lz4f_mi355x_debug_prof(unsigned long long* out128)
{
    if (!prof_buf()) return 1;
    (void)hipDeviceSynchronize();
    const int rc = hipMemcpy(out128, g_prof, 1024, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
#ifdef DB_PROF
    { unsigned long long z[32] = {0}; (void)hipMemcpyFromSymbol(out128 + 96, HIP_SYMBOL(lz4f::g_dbprof), sizeof(z)); (void)hipMemcpyToSymbol(HIP_SYMBOL(lz4f::g_dbprof), z, sizeof(z)); }
#endif
    (void)hipMemset(g_prof, 0, 1024); (void)hipMemset(g_prof + 70, 0xFF, 8); (void)hipMemset(g_prof + 75, 0xFF, 8);      // (the grid-wide words accumulate: start again)
    return rc;
}

// The record pool of one compress call, in records: `per_tile` a 64 KiB tile on average (0 = the default, 12288: a sequence per 5.3 input
// bytes - the densest input of the tests, Zipf text, needs more than 8192 per tile; the bench input one per 2000), never less than 64 tiles' worst case
// (small calls are sized for the worst case outright) and never more than the worst case.  LZ4F_MI355X_RECS_PER_TILE=16385 is the worst
// case for every tile; a caller that knows its data is sparse sets it low (1024: 0.13 bytes of workspace per input byte).
static uint32_t device_cus(int device)
{
    static int cus[64];                                                  // (0: not asked yet; a benign race: every thread stores the same number)
    const unsigned d = (unsigned)device % 64;
    if (!cus[d]) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v < 1) v = 256; cus[d] = v; }
    return (uint32_t)cus[d];
}

static uint64_t rec_pool_records(uint32_t n_chunks, uint32_t max_rec_per_chunk, unsigned per_tile)
{
    const uint64_t worst = (uint64_t)(n_chunks + 1) * max_rec_per_chunk;
    uint64_t want = (uint64_t)(n_chunks + 1) * (per_tile ? per_tile : 12288u);
    if (want < 64ull * max_rec_per_chunk) want = 64ull * max_rec_per_chunk;
    if (want > worst) want = worst;
    if (want > 0xFFFFFFF0ull) want = 0xFFFFFFF0ull;                      // (a list's place is a 32-bit record number)
    return want;
}

size_t lz4f_mi355x_engine::launch_compress(const CompressJob& j, uint8_t* d_dst, uint64_t dst_cap,
                                           lz4f_mi355x_result* d_res, lz4f_mi355x_block* d_table, void* d_index, size_t index_cap)
{    // in-band: the index is made in the engine's own buffer and copied, with the block list, into a skippable frame behind the
    // LZ4 frame (frame_dev.cuh: the trailer)
    const bool inband = d_index == nullptr && index_cap == LZ4F_MI355X_INBAND;
    HIP_TRY(hipSetDevice(device));                     // (before anything is allocated: the caller's current device may be another one)
    if (inband) {
        const size_t bsz = j.block_size, nb_ = (size_t)((j.src_size - j.first_off + bsz - 1) / bsz);
        const uint32_t ch_ = pick_chunk_size(j.block_size);
        index_cap = ix_entries_at((uint32_t)nb_, j.block_size / ch_) + ix_typical_entries(j.src_size - j.first_off, (uint32_t)(nb_ * (j.block_size / ch_))) * sizeof(IxEntry) + 64;
        if (ixtmp.ensure(index_cap + 64) || res.ensure(sizeof(ResultRec) + sizeof(TrailerPlan) + 64)) return make_err(LZ4F_ERROR_allocation_failed);
        d_index = ixtmp.p;
        if (((uintptr_t)d_dst & 15) != 0) { set_last_error("in-band index: the frame buffer must be 16-byte aligned"); return make_err(LZ4F_ERROR_GENERIC); }
    }

    hipStream_t st = (hipStream_t)stream;
    EncGeom g;
    memset(&g, 0, sizeof(g));
    g.src_size = j.src_size; g.first_off = j.first_off; g.write_endmark = j.endmark ? (j.content_checksum ? 2 : 1) : 0;
    g.block_size = j.block_size;
    g.chunk_size = pick_chunk_size(j.block_size);
    g.chunks_per_block = j.block_size / g.chunk_size;
    const uint64_t payload = j.src_size - j.first_off;
    const uint64_t nb = (payload + j.block_size - 1) / j.block_size;
    if (nb > 0x7FFFFFFFull / g.chunks_per_block) { set_last_error("input too large for one call"); return make_err(LZ4F_ERROR_srcSize_tooLarge); }
    g.n_blocks = (uint32_t)nb; g.n_chunks = g.n_blocks * g.chunks_per_block;
    g.linked = j.linked; g.block_checksum = j.block_checksum;
    g.header_size = j.header_size; memcpy(g.header, j.header, j.header_size);
    g.max_rec_per_chunk = g.chunk_size / 4 + 1;
    g.seed_stride = sw.seed;

    if (info.ensure((size_t)(g.n_chunks + 1) * sizeof(ChunkInfo))) return make_err(LZ4F_ERROR_allocation_failed);
    // (deterministic mode: the worst case for every tile - which tiles a short pool turns away is a matter of which workgroup's merge
    // gets to the bump pointer first, and "equal input, equal bytes" must not hang on that: 2 bytes of workspace per input byte)
    g.rec_pool = rec_pool_records(g.n_chunks, g.max_rec_per_chunk, (sw.e1_solo & 1u) ? 16385u : sw.recs_per_tile);
    if (recs.ensure((size_t)(rec_pool_at(g.n_chunks) + g.rec_pool) * 8)) return make_err(LZ4F_ERROR_allocation_failed);
    if (blk_bytes.ensure((size_t)(g.n_blocks + 1) * 4)) return make_err(LZ4F_ERROR_allocation_failed);
    if (!d_table) { if (table.ensure((size_t)(g.n_blocks + 1) * sizeof(BlockOut))) return make_err(LZ4F_ERROR_allocation_failed); d_table = (lz4f_mi355x_block*)table.p; }
    if (res.ensure(sizeof(ResultRec) + sizeof(TrailerPlan) + 64)) return make_err(LZ4F_ERROR_allocation_failed);
    if (!d_res) d_res = (lz4f_mi355x_result*)res.p;

    constexpr int W = 4;
    for (int i = 0; i < 4; i++) ev_used[i] = false;
    ev_used[10] = false;
    tick(10, false);
    if (g.n_chunks) {
        tick(0, false);
        {
            // a workgroup (one per CU: ~150 KiB of LDS) takes a run of consecutive 64 KiB tiles.  At most 1024 workgroups (each
            // has its slice lists in `e1_scratch`).  Round 4: as many workgroups as there are CUs where the input has fewer than 64 tiles
            // for each, runs of 64 tiles from there on - a run's first tile pays for the 64 KiB of history in front of it, for seeding the
            // table with them and for not knowing the data's density yet, so fewer, longer runs win until the CUs run out of work:
            // tools/e1_run_sweep.py, tiles per workgroup 1024-wide rule -> this one: 64 MiB 0.132 -> 0.065 ms (ratio 1.9154 -> 1.9425),
            // 256 MiB 0.256 -> 0.187, 1 GiB 0.722 -> 0.647, 2 GiB 1.342 -> 1.228; 4 GiB and beyond as before (64 tiles, 1024 workgroups).
            uint32_t run = g.n_chunks / device_cus(device); run = run < 1 ? 1 : run > 64 ? 64 : run;
            if ((g.n_chunks + run - 1) / run > 1024) run = (g.n_chunks + 1023) / 1024;
            if (sw.e1_run) run = sw.e1_run;
            g.tiles_per_wg = run;
            g.e1_solo = sw.e1_solo;
            const uint32_t n_wg = (g.n_chunks + run - 1) / run;
            if (e1_scratch.ensure((size_t)n_wg * 2 * E1_NSLICE * E1_REC_PER_SLICE * 8 + 2048)) return make_err(LZ4F_ERROR_allocation_failed);
#ifdef E1_DEBUG
            (void)hipMemsetAsync((uint8_t*)e1_scratch.p + (size_t)n_wg * 2 * E1_NSLICE * E1_REC_PER_SLICE * 8, 0, 2048, st);
#endif
            // (the pool's bump pointer and its count of tiles turned away: every call's scan leaves them at zero for the next; zeroed here when the
            // workspace is new, or when a call before this one may not have got as far as its scan)
            if (recs_ctl_clean != recs.p) { HIP_TRY(hipMemsetAsync(recs.p, 0, 64, st)); }
            recs_ctl_clean = nullptr;
            // deterministic mode: a wave per chunk with a table of its own (encode_solo.cuh) - nothing shared, nothing that depends on timing.
            // (e1_solo bit 2: the shared kernel with one wave per workgroup parsing, the mode's form until round 4 - kept for comparison)
            if ((sw.e1_solo & 1u) && !(sw.e1_solo & 4u))
                hipLaunchKernelGGL((k_find_matches_solo<1>), dim3(g.n_chunks), dim3(64), 0, st, j.d_src, g, (ChunkInfo*)info.p, (uint64_t*)recs.p);
            else
            hipLaunchKernelGGL(k_find_matches, dim3(n_wg), dim3(64 * E1_WAVES), 0, st, j.d_src, g, (ChunkInfo*)info.p, (uint64_t*)recs.p, (uint64_t*)e1_scratch.p);
            if (sw.e1_sync) (void)hipStreamSynchronize(st);
#ifdef E1_DEBUG
            { unsigned long long d[256]; if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(d, (uint8_t*)e1_scratch.p + (size_t)n_wg * 2 * E1_NSLICE * E1_REC_PER_SLICE * 8, 2048, hipMemcpyDeviceToHost) == hipSuccess) {
                for (int w = 0; w < 16; w += 5) { unsigned long long* x = d + 16 + w * 8; if (x[6]) fprintf(stderr, "E1 wave %d: per tile cycles: merge %llu parse %llu waitB1 %llu dma-issue %llu dma-wait %llu waitB2 %llu (%llu tiles)\n", w, x[0]/x[6], x[1]/x[6], x[2]/x[6], x[3]/x[6], x[4]/x[6], x[5]/x[6], x[6]); unsigned long long* f = d + 160 + w * 6; fprintf(stderr, "   parse: dequeue %llu cycles x %llu, probe step %llu cycles x %llu, hit %llu cycles x %llu (per tile)\n", f[3] ? f[0]/f[3] : 0, f[3]/x[6], f[4] ? f[1]/f[4] : 0, f[4]/x[6], f[5] ? f[2]/f[5] : 0, f[5]/x[6]); }
                fprintf(stderr, "E1 dense passes (workgroup 0): %llu, matches taken %llu, positions advanced %llu; one-match steps because: hit in B %llu, mode not dense %llu, step != 1 %llu, first match long %llu\n", d[13], d[14], d[15], d[4], d[5], d[6], d[7]);
                fprintf(stderr, "E1 debug: bounds hit: dequeue %llu, probe %llu, backward %llu, forward %llu; probe ip/last %llx step/slice %llx; back room/nb %llx; fwd mp/fw %llx end_lim/d %llx\n", d[0], d[1], d[2], d[3], d[8], d[9], d[10], d[11], d[12]); } }
#endif
        }
        tick(0, true);
    }
    tick(1, false);
    {
        if (g.n_blocks <= LAYOUT_SMALL_BLOCKS && g.n_chunks <= LAYOUT_SMALL_CHUNKS)       // (a few blocks - the streaming API's one per call: one launch instead of three)
            hipLaunchKernelGGL(k_layout_small, dim3(1), dim3(1024), 0, st, g, (ChunkInfo*)info.p, (BlockOut*)d_table, (uint32_t*)blk_bytes.p, d_dst, dst_cap, (ResultRec*)d_res, (const uint64_t*)recs.p);
        else {
        if (g.n_blocks) hipLaunchKernelGGL((k_layout_blocks<W>), dim3((g.n_blocks + W - 1) / W), dim3(64 * W), 0, st, g, (ChunkInfo*)info.p, (BlockOut*)d_table, (uint32_t*)blk_bytes.p);
        hipLaunchKernelGGL(k_layout_scan, dim3(1), dim3(1024), 0, st, g, (BlockOut*)d_table, (const uint32_t*)blk_bytes.p, d_dst, dst_cap, (ResultRec*)d_res, (const uint64_t*)recs.p);
        if (g.n_chunks) hipLaunchKernelGGL(k_layout_chunks, dim3((g.n_chunks + 255) / 256), dim3(256), 0, st, g, (ChunkInfo*)info.p, (const BlockOut*)d_table, d_dst, (const ResultRec*)d_res);
        }
        if (g.n_chunks) recs_ctl_clean = recs.p;                                  // (the scan is enqueued: it leaves the pool's control words at zero)
        if (d_index) {                                                            // sequence index for the indexed decoder
            if (g.n_blocks) hipLaunchKernelGGL((k_index_blocks<W>), dim3((g.n_blocks + W - 1) / W), dim3(64 * W), 0, st, g, (const ChunkInfo*)info.p, (const BlockOut*)d_table, (const ResultRec*)d_res, d_index, (uint64_t)index_cap);
            hipLaunchKernelGGL(k_build_index, dim3(1), dim3(1024), 0, st, g, (const ChunkInfo*)info.p, (const BlockOut*)d_table, (const ResultRec*)d_res, d_index, (uint64_t)index_cap, 1u);
        }
    }
    tick(1, true);
    if (g.n_chunks) {
        tick(2, false);
        // (few chunks - the streaming API's one block per call: the four waves of a workgroup share a chunk instead of taking one each)
        const bool e2_split = g.n_chunks <= 512;
        if (e2_split)
            hipLaunchKernelGGL((k_emit_gather<W, true>), dim3(g.n_chunks), dim3(64 * W), 0, st, j.d_src, g, (const ChunkInfo*)info.p,
                               (const uint64_t*)recs.p, d_dst, (const BlockOut*)d_table, d_index);
        else
        hipLaunchKernelGGL((k_emit_gather<W, false>), dim3((g.n_chunks + W - 1) / W), dim3(64 * W), 0, st, j.d_src, g, (const ChunkInfo*)info.p,
                           (const uint64_t*)recs.p, d_dst, (const BlockOut*)d_table, d_index);
        tick(2, true);
        if (j.block_checksum) {
            tick(3, false);
            if (g.n_blocks < XXH_LANE4_BELOW)                          // few big blocks: the four accumulators as four lanes (lane4_xxh32)
                hipLaunchKernelGGL((k_xxh32_blocks4<1>), dim3(g.n_blocks), dim3(64), XXH_SPREAD_LDS, st, d_dst, (BlockOut*)d_table, (const ResultRec*)d_res, g.n_blocks, 0u, (uint32_t*)nullptr);
            else
            hipLaunchKernelGGL((k_xxh32_blocks<W>), dim3((g.n_blocks + W - 1) / W), dim3(64 * W), 0, st, d_dst, (BlockOut*)d_table,
                               (const ResultRec*)d_res, g.n_blocks, 0u, (uint32_t*)nullptr);
            tick(3, true);
        }
    }
    if (j.endmark && j.content_checksum)                              // (one chain over the whole input: see k_xxh32_content for what that costs)
        hipLaunchKernelGGL(k_xxh32_content, dim3(1), dim3(64), 0, st, j.d_src + j.first_off, (uint64_t)(j.src_size - j.first_off), d_dst, (ResultRec*)d_res, 0u);
    if (inband && g.n_blocks) {
        TrailerPlan* plan = (TrailerPlan*)((uint8_t*)res.p + sizeof(ResultRec) + 32);
        hipLaunchKernelGGL(k_trailer_plan, dim3(1), dim3(64), 0, st, d_dst, dst_cap, (ResultRec*)d_res, g.n_blocks, (const void*)d_index,
                           (uint64_t)ix_entries_at(g.n_blocks, g.chunks_per_block), plan);
        hipLaunchKernelGGL(k_trailer_copy, dim3(256), dim3(256), 0, st, d_dst, (const TrailerPlan*)plan, (const BlockOut*)d_table, g.n_blocks, (const void*)d_index);
    }
    tick(10, true);
    HIP_TRY(hipGetLastError());
    return 0;
}

size_t lz4f_mi355x_engine::launch_decompress(const DecompressJob& j, lz4f_mi355x_result* d_res)
{
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    if (!d_res) { if (res.ensure(sizeof(ResultRec))) return make_err(LZ4F_ERROR_allocation_failed); d_res = (lz4f_mi355x_result*)res.p; }
    if (bad.ensure(64)) return make_err(LZ4F_ERROR_allocation_failed);
    BlockOut* tbl;
    uint32_t n_max;
    bool bad_set = false;                                            // the finishing kernels' verdict words are initialised by a kernel already launched
    uint32_t plan = 0;                                               // LZ4F_MI355X_PATH_*: reported in result.flags
    for (int i = 4; i < 10; i++) ev_used[i] = false;
    ev_used[11] = false;
    if (aux_pending) { HIP_TRY(hipStreamWaitEvent(st, (hipEvent_t)ev_join, 0)); aux_pending = false; }      // (a call that left early: its forked work first)
    // every exit behind the fork joins: an error return (allocation, a HIP call) leaves the main stream waiting for the forked verification, so
    // that whatever the caller enqueues or frees behind a synchronisation of its stream comes after the kernel that still reads the frame
    struct AuxJoin {
        lz4f_mi355x_engine* e; hipStream_t st;
        ~AuxJoin() { if (e->aux_pending && hipStreamWaitEvent(st, (hipEvent_t)e->ev_join, 0) == hipSuccess) e->aux_pending = false; }
    } aux_join{this, st};
    tick(11, false);
    if (j.d_table || j.table_in_place || j.table_direct) {
        // caller-supplied table: work on a copy (decode overwrites dst_size); table_direct: the engine's own staging copy, used where it lies
        n_max = j.n_blocks;
        if (j.table_direct) tbl = (BlockOut*)j.table_direct;
        else {
            if (table.ensure((size_t)(n_max + 1) * sizeof(BlockOut))) return make_err(LZ4F_ERROR_allocation_failed);
            tbl = (BlockOut*)table.p;
        }
        if (!j.table_in_place && !j.table_direct)
            HIP_TRY(hipMemcpyAsync(tbl, j.d_table, (size_t)n_max * sizeof(BlockOut), hipMemcpyDeviceToDevice, st));
        if (n_max <= 256) {                                          // (a few blocks - the streaming API's one per call: one launch for the record, the verdict words and the table check)
            hipLaunchKernelGGL(k_begin_table_small, dim3(1), dim3(256), 0, st, (const BlockOut*)tbl, n_max, (uint64_t)j.frame_cap, (uint64_t)j.dst_cap,
                               j.block_size, j.block_checksum ? 1u : 0u, j.linked ? 1u : 0u, (ResultRec*)d_res, (uint32_t*)bad.p);
            bad_set = true;
        } else {
        hipLaunchKernelGGL(k_init_result, dim3(1), dim3(64), 0, st, (ResultRec*)d_res, n_max, 0u);
        if (n_max) hipLaunchKernelGGL(k_check_table, dim3(std::min<uint32_t>((n_max + 255) / 256, 1024u)), dim3(256), 0, st, (const BlockOut*)tbl, n_max, (uint64_t)j.frame_cap, (uint64_t)j.dst_cap,
                                      j.block_size, j.block_checksum ? 1u : 0u, j.linked ? 1u : 0u, (ResultRec*)d_res);
        }
        plan |= LZ4F_MI355X_PATH_TABLE_GIVEN;
    } else {
        n_max = j.max_blocks;
        if (table.ensure((size_t)(n_max + 1) * sizeof(BlockOut))) return make_err(LZ4F_ERROR_allocation_failed);
        tbl = (BlockOut*)table.p;
        tick(4, false);
        // frames of many small blocks: the size words are found in parallel (frame_dev.cuh); k_walk_frame behind it returns at
        // once when that has delivered, and walks the list itself otherwise (big blocks: a few hundred hops, and one in 2^9
        // byte positions would be a candidate)
        const uint32_t* walked = nullptr;
        if (j.hint_list && j.hint_n <= n_max) {
            // the frame's own trailer says where the size words are: checked link by link like the parallel walk's candidates
            if (walkbuf.ensure(256)) return make_err(LZ4F_ERROR_allocation_failed);
            WalkState* ws = (WalkState*)walkbuf.p;
            hipLaunchKernelGGL(k_walk_head, dim3(1), dim3(64), 0, st, j.d_frame, j.frame_cap, ws, j.hint_n);
            hipLaunchKernelGGL(k_walk_link, dim3(std::min<uint32_t>((j.hint_n + 255) / 256, 4096u)), dim3(256), 0, st, j.d_frame, j.frame_cap, j.dst_cap, ws, j.hint_list, tbl, n_max);
            hipLaunchKernelGGL(k_walk_verdict, dim3(1), dim3(64), 0, st, j.d_frame, j.frame_cap, j.dst_cap, ws, j.hint_list, n_max, (ResultRec*)d_res);
            walked = &ws->done;
            plan |= LZ4F_MI355X_PATH_TRAILER;
        } else
        if (j.block_size <= (256u << 10) && j.frame_cap >= (1u << 20) && !sw.serial_walk) {
            const uint32_t n_chunks = (uint32_t)((j.frame_cap + WK_CHUNK - 1) / WK_CHUNK);
            const size_t list_cap = (size_t)n_max + 1024;
            const size_t at_chunks = 256, at_list = at_chunks + (size_t)n_chunks * sizeof(WalkChunk), at_list2 = at_list + list_cap * 8, at_mark = at_list2 + list_cap * 8;
            if (walkbuf.ensure(at_mark + list_cap * 4)) return make_err(LZ4F_ERROR_allocation_failed);
            WalkState* ws = (WalkState*)walkbuf.p;
            WalkChunk* ch = (WalkChunk*)((uint8_t*)walkbuf.p + at_chunks);
            uint64_t* list = (uint64_t*)((uint8_t*)walkbuf.p + at_list);
            uint64_t* list2 = (uint64_t*)((uint8_t*)walkbuf.p + at_list2);
            uint32_t* mark = (uint32_t*)((uint8_t*)walkbuf.p + at_mark);
            const uint32_t lgrid = std::min<uint32_t>((uint32_t)((list_cap + 255) / 256), 4096u);
            HIP_TRY(hipMemsetAsync(mark, 0, list_cap * 4, st));
            hipLaunchKernelGGL(k_walk_head, dim3(1), dim3(64), 0, st, j.d_frame, j.frame_cap, ws);
            hipLaunchKernelGGL(k_walk_cand, dim3(n_chunks), dim3(256), 0, st, j.d_frame, j.frame_cap, (const WalkState*)ws, ch);
            hipLaunchKernelGGL(k_walk_order, dim3(1), dim3(1024), 0, st, ch, n_chunks, ws, list, (uint32_t)list_cap);
            hipLaunchKernelGGL(k_walk_mark, dim3(lgrid), dim3(256), 0, st, j.d_frame, j.frame_cap, (const WalkState*)ws, (const uint64_t*)list, mark);
            hipLaunchKernelGGL(k_walk_filter, dim3(1), dim3(1024), 0, st, ws, (const uint64_t*)list, (const uint32_t*)mark, list2);
            hipLaunchKernelGGL(k_walk_link, dim3(lgrid), dim3(256), 0, st, j.d_frame, j.frame_cap, j.dst_cap, ws, (const uint64_t*)list2, tbl, n_max);
            hipLaunchKernelGGL(k_walk_verdict, dim3(1), dim3(64), 0, st, j.d_frame, j.frame_cap, j.dst_cap, ws, (const uint64_t*)list2, n_max, (ResultRec*)d_res);
            walked = &ws->done;
            plan |= LZ4F_MI355X_PATH_PARALLEL_WALK;
            if (sw.prof) { WalkState h; if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(&h, ws, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) fprintf(stderr, "parallel walk: done %u overflow %u candidates %u first_end %u first_break %u (header ok %u, hsize %u, block %u)\n", h.done, h.overflow, h.total, h.first_end, h.first_break, h.head_ok, h.hsize, h.bs); }
        }
        else if (j.block_size > (256u << 10) && j.frame_cap >= (size_t)192 * j.block_size && !sw.serial_walk) {      // (~0.15 ms whatever the frame: pays from ~330 blocks of half their size on)
            // big blocks: seeds found in parallel, a lane per seed walking to the next one (frame_dev.cuh); the list is checked link by
            // link like the small blocks' candidates, and k_walk_frame behind walks the frame itself if it is not the chain
            const size_t list_cap = (size_t)n_max + 1024;
            const size_t at_seeds = 256, at_list = at_seeds + (WK_SEEDS + 1) * 8;
            if (walkbuf.ensure(at_list + list_cap * 8)) return make_err(LZ4F_ERROR_allocation_failed);
            WalkState* ws = (WalkState*)walkbuf.p;
            unsigned long long* seeds = (unsigned long long*)((uint8_t*)walkbuf.p + at_seeds);
            HIP_TRY(hipMemsetAsync(seeds, 0xFF, (WK_SEEDS + 1) * 8, st));
            uint64_t* list = (uint64_t*)((uint8_t*)walkbuf.p + at_list);
            const uint32_t lgrid = std::min<uint32_t>((uint32_t)((list_cap + 255) / 256), 4096u);
            hipLaunchKernelGGL(k_walk_head, dim3(1), dim3(64), 0, st, j.d_frame, j.frame_cap, ws);
            hipLaunchKernelGGL(k_walk_seeds, dim3(WK_SEEDS * wk_seed_pieces(j.block_size)), dim3(256), 0, st, j.d_frame, j.frame_cap, (const WalkState*)ws, seeds);
            hipLaunchKernelGGL(k_walk_chains, dim3(1), dim3(WK_SEEDS), 0, st, j.d_frame, j.frame_cap, ws, (const unsigned long long*)seeds, list, (uint32_t)std::min<size_t>(list_cap, 0xFFFFFFFFu));
            hipLaunchKernelGGL(k_walk_link, dim3(lgrid), dim3(256), 0, st, j.d_frame, j.frame_cap, j.dst_cap, ws, (const uint64_t*)list, tbl, n_max);
            hipLaunchKernelGGL(k_walk_verdict, dim3(1), dim3(64), 0, st, j.d_frame, j.frame_cap, j.dst_cap, ws, (const uint64_t*)list, n_max, (ResultRec*)d_res);
            walked = &ws->done;
            plan |= LZ4F_MI355X_PATH_PARALLEL_WALK;
            if (sw.prof) { WalkState h; if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(&h, ws, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) fprintf(stderr, "seeded walk: done %u overflow %u entries %u first_end %u first_break %u\n", h.done, h.overflow, h.total, h.first_end, h.first_break); }
        }
        hipLaunchKernelGGL(k_walk_frame, dim3(1), dim3(64), 0, st, j.d_frame, j.frame_cap, j.dst_cap, tbl, n_max, (ResultRec*)d_res, walked);
        tick(4, true);
    }
    if (!bad_set) {
        HIP_TRY(hipMemsetAsync(bad.p, 0xFF, 8, st));                 // [0] block-checksum verdict, [1] first failed block
        HIP_TRY(hipMemsetAsync((uint8_t*)bad.p + 8, 0, 24, st));    // [2] "something has to move", [4..5] sum of sizes (k_finish_check)
    }
    constexpr int W = 4;
    const uint32_t grid = j.linked ? 1u : (n_max + W - 1) / W;
    const uint32_t* ix_flags = nullptr;                              // the indexed kernels' "gave up" word, if they were launched
    if (n_max) {
        if (j.block_checksum) {
            // The verification only reads the payloads, and so do the decode kernels: it runs beside them on the engine's second stream
            // (forked here, joined in front of k_finish_check, which reads its verdict).  A 4 MiB block is one serial chain for one wave
            // (~2 ms) whatever else the GPU does, so side by side the two cost max(2.0, decode) instead of the sum.  The one-wave
            // workgroups then ask for 12 KiB of LDS instead of 36 (they would not fit beside four decode workgroups per CU).
            const bool beside = !sw.no_overlap && aux_ready();
            hipStream_t xs = beside ? (hipStream_t)aux_stream : st;
            if (beside) { HIP_TRY(hipEventRecord((hipEvent_t)ev_fork, st)); HIP_TRY(hipStreamWaitEvent(xs, (hipEvent_t)ev_fork, 0)); }
            tick(5, false, xs);
            if (n_max < XXH_LANE4_BELOW)
                hipLaunchKernelGGL((k_xxh32_blocks4<1>), dim3(n_max), dim3(64), beside ? (12u << 10) : XXH_SPREAD_LDS, xs, (uint8_t*)j.d_frame, tbl, (const ResultRec*)d_res, n_max, 1u, (uint32_t*)bad.p);
            else
            hipLaunchKernelGGL((k_xxh32_blocks<W>), dim3((n_max + W - 1) / W), dim3(64 * W), 0, xs, (uint8_t*)j.d_frame, tbl,
                               (const ResultRec*)d_res, n_max, 1u, (uint32_t*)bad.p);
            tick(5, true, xs);
            if (beside) { HIP_TRY(hipEventRecord((hipEvent_t)ev_join, xs)); aux_pending = true; }
        }
        tick(6, false);
        // large blocks / linked frames: fused parse+copy workgroups ('f'); small independent blocks: one wave per block ('1')
        char mode = (j.linked || j.block_size >= (256u << 10)) ? 'f' : '1';
        if (sw.decode_mode) mode = sw.decode_mode;
        bool indexed = false;
        void* d_index = j.d_index; size_t index_size = j.index_size;
        bool self_indexed = false;
        uint32_t self_seqs = 0, self_entries = 0;
        // A linked frame is one match chain without a usable index (seconds instead of milliseconds on dense data), so for
        // those the header of the index that came along is read NOW (a host synchronisation, ~30 us): the compressor marks an
        // index unusable when the stream had more sequences than it had room for, and then one is made here instead.
        IxHeader hd_now; memset(&hd_now, 0, sizeof(hd_now));
        bool have_now = false;
        // (an index out of the frame's trailer brings its counts in the footer - no read - and its frame is one call's work: every block but the last is full,
        // so the table the list check writes has every block's place in the output, which is what a linked frame's indexed decode needs; if a trailer lies about
        // that the descriptors do not tile the blocks and the generic kernels take the frame)
        const bool ix_by_trailer = j.d_index && j.ix_seqs && j.hint_list && j.hint_n <= n_max;
        if (mode == 'f' && j.linked && j.d_index && j.index_size >= sizeof(IxHeader) && !sw.no_index && !ix_by_trailer) {
            HIP_TRY(hipMemcpyAsync(&hd_now, j.d_index, sizeof(hd_now), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            have_now = true;
            if (hd_now.magic != IX_MAGIC || hd_now.stride != IX_STRIDE) { d_index = nullptr; index_size = 0; have_now = false; }
        }
        if (mode == 'f' && j.linked && !d_index && j.hist0 <= 65536 && n_max >= 2 && !sw.no_index && !sw.no_selfindex) {
            // A linked frame without an index (a foreign one: the reference's default output): make the index here - a lane per block
            // walks the payload (parsing needs no history), a scan places the blocks - and take the same kernels as with the
            // compressor's index.  Two host synchronisations (the totals size the buffers); anything odd leaves the frame to the
            // window kernel, as before.
            const uint32_t cpb = j.block_size / pick_chunk_size(j.block_size);
            const size_t fixed = ix_entries_at(n_max, cpb);
            if (selfcnt.ensure((size_t)n_max * 8 + 64) || seqcnt.ensure(256 + (size_t)n_max * (8 + 8 * IXL_PUB)) || selfix.ensure(fixed + 64))
                return make_err(LZ4F_ERROR_allocation_failed);
            uint32_t* cnt = (uint32_t*)selfcnt.p; uint32_t* osz = cnt + n_max;
            HIP_TRY(hipMemsetAsync(seqcnt.p, 0, 64, st));
            if (density.ensure(64)) return make_err(LZ4F_ERROR_allocation_failed);
            hipLaunchKernelGGL(k_density_probe, dim3(1), dim3(64), 0, st, j.d_frame, (uint64_t)j.frame_cap, (const BlockOut*)tbl, (const ResultRec*)d_res, n_max, (uint32_t*)density.p);
            hipLaunchKernelGGL((k_selfindex_walk_wave<0, 4>), dim3((n_max + 3) / 4), dim3(256), 0, st, j.d_frame, (uint64_t)j.frame_cap, (const BlockOut*)tbl,
                               (const ResultRec*)d_res, n_max, cnt, osz, (void*)nullptr, (uint32_t*)seqcnt.p, (const uint32_t*)density.p);
            hipLaunchKernelGGL(k_selfindex_walk<0>, dim3((n_max + 255) / 256), dim3(256), 0, st, j.d_frame, (uint64_t)j.frame_cap, (const BlockOut*)tbl,
                               (const ResultRec*)d_res, n_max, cnt, osz, (void*)nullptr, (uint32_t*)seqcnt.p, (const uint32_t*)density.p + 1);
            uint32_t tot[10];
            for (int pass = 0; pass < 2; pass++) {
                hipLaunchKernelGGL(k_selfindex_scan, dim3(1), dim3(1024), 0, st, tbl, (const ResultRec*)d_res, n_max, (const uint32_t*)cnt, (const uint32_t*)osz,
                                   selfix.p, cpb, (uint64_t)j.dst_cap, j.block_size, (uint32_t*)seqcnt.p);
                if (pass == 1) break;
                HIP_TRY(hipMemcpyAsync(tot, seqcnt.p, sizeof(tot), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                if (tot[0] != 0 || tot[9] == 0) break;
                const void* before = selfix.p;
                if (selfix.ensure(fixed + (size_t)tot[8] * sizeof(IxEntry) + 64)) return make_err(LZ4F_ERROR_allocation_failed);
                if (selfix.p == before) break;                                   // (same buffer: the block table is already in it)
            }
            if (tot[0] == 0 && tot[9] != 0) {
                hipLaunchKernelGGL((k_selfindex_walk_wave<1, 4>), dim3((n_max + 3) / 4), dim3(256), 0, st, j.d_frame, (uint64_t)j.frame_cap, (const BlockOut*)tbl,
                                   (const ResultRec*)d_res, n_max, cnt, osz, selfix.p, (uint32_t*)seqcnt.p, (const uint32_t*)density.p);
                hipLaunchKernelGGL(k_selfindex_walk<1>, dim3((n_max + 255) / 256), dim3(256), 0, st, j.d_frame, (uint64_t)j.frame_cap, (const BlockOut*)tbl,
                                   (const ResultRec*)d_res, n_max, cnt, osz, selfix.p, (uint32_t*)seqcnt.p, (const uint32_t*)density.p + 1);
                d_index = selfix.p; index_size = fixed + (size_t)tot[8] * sizeof(IxEntry);
                self_indexed = true;
                plan |= LZ4F_MI355X_PATH_SELF_INDEX;
                self_seqs = tot[9]; self_entries = tot[8];
            }
        }
        bool spx_mode = false;
        if (mode == 'f' && !j.linked && !d_index && j.block_size >= (256u << 10) && j.block_size <= (4u << 20) && !sw.no_index && !sw.no_selfindex && !sw.no_spx) {
            // A frame of big independent blocks without an index (a foreign one: `lz4 -c`, LZ4F_compressFrame): the blocks are cut into
            // stretches by lanes that start at guessed tokens and are stitched where they meet (decode_spx.cuh); the stretches are
            // parsed in parallel, then the indexed kernels.  One host synchronisation (the totals size the descriptor workspace);
            // dense payloads (k_density_probe) and anything odd stay with the generic decoders.
            const uint32_t cpb = j.block_size / pick_chunk_size(j.block_size);
            const size_t fixed = ix_entries_at(n_max, cpb);
            if (selfcnt.ensure((size_t)n_max * 8 + 64) || seqcnt.ensure(256 + (size_t)n_max * (8 + 8 * IXL_PUB)) || selfix.ensure(fixed + 64) || density.ensure(64) ||
                spx.ensure((size_t)n_max * ((SPX_MAXPT + 1) * sizeof(SpxPoint) + 4) + 64))
                return make_err(LZ4F_ERROR_allocation_failed);
            uint32_t* cnt = (uint32_t*)selfcnt.p; uint32_t* osz = cnt + n_max;
            SpxPoint* spt = (SpxPoint*)spx.p; uint32_t* snr = (uint32_t*)((uint8_t*)spx.p + (size_t)n_max * (SPX_MAXPT + 1) * sizeof(SpxPoint));
            HIP_TRY(hipMemsetAsync(seqcnt.p, 0, 256, st));
            hipLaunchKernelGGL(k_density_probe, dim3(1), dim3(64), 0, st, j.d_frame, (uint64_t)j.frame_cap, (const BlockOut*)tbl, (const ResultRec*)d_res, n_max, (uint32_t*)density.p);
            hipLaunchKernelGGL(k_spx_index, dim3(n_max), dim3(SPX_MAXSEG), 0, st, j.d_frame, (uint64_t)j.frame_cap, (const BlockOut*)tbl, (const ResultRec*)d_res, n_max, cnt, osz,
                               spt, snr, (uint32_t*)seqcnt.p, sw.no_density_probe ? (const uint32_t*)nullptr : (const uint32_t*)density.p + 1);
            uint32_t tot[10];
            for (int pass = 0; pass < 2; pass++) {
                hipLaunchKernelGGL(k_selfindex_scan, dim3(1), dim3(1024), 0, st, tbl, (const ResultRec*)d_res, n_max, (const uint32_t*)cnt, (const uint32_t*)osz,
                                   selfix.p, cpb, (uint64_t)j.dst_cap, j.block_size, (uint32_t*)seqcnt.p, 1u);
                if (pass == 1) break;
                HIP_TRY(hipMemcpyAsync(tot, seqcnt.p, sizeof(tot), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                if (sw.prof) fprintf(stderr, "spx: flags %u, %u sequences in %u blocks, %u stretches walked by the stitching thread\n", tot[0], tot[9], n_max, tot[2]);
#ifdef SPX_PROF
                { uint32_t y[8]; if (hipMemcpy(y, (uint32_t*)seqcnt.p + 48, 32, hipMemcpyDeviceToHost) == hipSuccess) fprintf(stderr, "spx lanes: hit the hop cap %u, without a start %u, started at their segment's first byte %u, ran into something %u; most sequences in one lane %u\n", y[0], y[1], y[2], y[3], y[4]); }
                { uint32_t z[8]; if (hipMemcpy(z, (uint32_t*)seqcnt.p + 40, 32, hipMemcpyDeviceToHost) == hipSuccess) fprintf(stderr, "spx cycles: guess max %u avg %u, walk max %u avg %u, stitch max %u avg %u, workgroup max %u (waves %u)\n", z[0], (unsigned)(((unsigned long long)z[4] << 8) / (2 * n_max)), z[1], (unsigned)(((unsigned long long)z[5] << 8) / (2 * n_max)), z[2], (unsigned)(((unsigned long long)z[6] << 8) / n_max), z[3], 2 * n_max); }
#endif
                if (tot[0] != 0 || tot[9] == 0) break;
                const void* before = selfix.p;
                if (selfix.ensure(fixed + (size_t)tot[8] * sizeof(IxEntry) + 64)) return make_err(LZ4F_ERROR_allocation_failed);
                if (selfix.p == before) break;                                   // (same buffer: the block table is already in it)
            }
            if (tot[0] == 0 && tot[9] != 0) {
                d_index = selfix.p; index_size = fixed + (size_t)tot[8] * sizeof(IxEntry);
                self_indexed = true; spx_mode = true;
                plan |= LZ4F_MI355X_PATH_SELF_INDEX;
                self_seqs = tot[9]; self_entries = tot[8];
            }
        }
        // (linked frames: only with a table that has every block's output position - the compressor's, or the one just made)
        if (mode == 'f' && d_index && index_size >= sizeof(IxHeader) && (!j.linked || self_indexed || ((j.d_table || j.table_in_place || j.table_direct || ix_by_trailer) && j.hist0 <= 65536)) &&
            !sw.no_index) {
            // Descriptors from the compressor's sequence index: a lane per entry parses, a lane per sequence resolves direct
            // matches, a workgroup per block copies.
            const uint32_t chunk = pick_chunk_size(j.block_size), cpb = j.block_size / chunk;
            // (an index out of the frame's trailer is laid out for the block count the trailer names; the table and the generic
            // kernels keep the caller's upper bound - if the walk finds another count, the index is dropped on the device)
            const uint32_t n_ix = (j.hint_list && j.hint_n <= n_max) ? j.hint_n : n_max;
            // How many sequences and entries the index holds comes with THIS call (the trailer's footer, the self-index scan, or one
            // read of the index header for the explicit-index entry point): it sizes the descriptor workspace and decides whether a
            // dense frame's scratch is worth having.  Nothing is carried over from earlier calls; the device checks the real header
            // against the workspace (k_check_index) and hands the call to the generic decoder if it does not fit.
            uint32_t ix_seqs = j.ix_seqs, ix_entries = j.ix_entries;
            if (self_indexed) { ix_seqs = self_seqs; ix_entries = self_entries; }
            else if (have_now) { ix_seqs = hd_now.total_seqs; ix_entries = hd_now.total_entries; }
            else if (!ix_seqs) {
                IxHeader hd; memset(&hd, 0, sizeof(hd));
                HIP_TRY(hipMemcpyAsync(&hd, d_index, sizeof(hd), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                if (hd.magic == IX_MAGIC && hd.stride == IX_STRIDE) { ix_seqs = hd.total_seqs; ix_entries = hd.total_entries; }
            }
            if ((uint64_t)ix_seqs > (uint64_t)ix_entries * (IX_STRIDE + 1) || ix_entries > (index_size - sizeof(IxHeader)) / sizeof(IxEntry)) ix_seqs = 0;      // (not a count this index can hold)
            const bool ix_dense = (uint64_t)ix_seqs * 48 > (uint64_t)n_ix * j.block_size;      // short sequences: worth the tracer's scratch, see below
            const uint32_t ix_entries_hint = ix_entries;
            if ((size_t)ix_seqs + 4096 > ix_seq_cap) ix_seq_cap = (size_t)ix_seqs + ix_seqs / 4 + 4096;
            if (ix_seqs) {
                const size_t dsrc_at = (ix_seq_cap + 64) * sizeof(SeqDesc);
                if (desc.ensure(dsrc_at + (ix_seq_cap + 64) * 4) || seqcnt.ensure(256 + (size_t)n_max * (8 + 8 * IXL_PUB))) return make_err(LZ4F_ERROR_allocation_failed);
                HIP_TRY(hipMemsetAsync(seqcnt.p, 0, 256 + (j.linked ? (size_t)n_max * 8 : 0), st));      // flags (+ per block of a linked frame: the "done" word and the count of published ranges)
                uint32_t* done = (uint32_t*)seqcnt.p + 64;
                uint32_t lk = j.linked ? 1u : 0u;                                  // (bits 1..: chain gate, see k_copy_indexed)
                if (j.linked && sw.chain_gate) lk |= (uint32_t)sw.chain_gate << 1;
                unsigned long long* iprof = (unsigned long long*)(sw.prof ? prof_buf() : nullptr);
                tick(8, false);
                hipLaunchKernelGGL(k_check_index, dim3(1), dim3(64), 0, st, (const void*)d_index, (uint64_t)index_size, n_ix, cpb, chunk,
                                   (uint64_t)ix_seq_cap, (uint32_t*)seqcnt.p, (const ResultRec*)d_res);
                uint32_t n_lanes = ix_entries_hint > n_ix ? ix_entries_hint : n_ix;           // (grid-stride inside: a hint is enough)
                // Independent blocks that are not dense (no tracer on offer): ONE kernel - the copy workgroup's first wave parses its block's
                // runs and resolves direct matches while the copiers move bytes (decode_indexed.cuh: k_copy_selffed)
                const bool can_double0 = ((uint64_t)ix_seqs * 48 > (uint64_t)n_ix * j.block_size || sw.trace_always) && (uint64_t)n_ix * j.block_size <= IXP_MAX_SPAN && !sw.no_doubling;
                const bool selffeed = !j.linked && !sw.no_selffeed && !sw.no_resolve && !sw.trace_always && !can_double0;
                if (selffeed) {
                    tick(8, true);
                    tick(9, false);
                    if (spx_mode) {
                        const FzSrcSpx src{(const SpxPoint*)spx.p, (const uint32_t*)((const uint8_t*)spx.p + (size_t)n_max * (SPX_MAXPT + 1) * sizeof(SpxPoint))};
                        if (j.block_size <= (1u << 20))
                            hipLaunchKernelGGL((k_copy_selffed<FzCfgS4, FzSrcSpx>), dim3(n_ix), dim3(64 * 4), 0, st, j.d_frame, (uint64_t)j.frame_cap, j.d_dst, tbl, (const ResultRec*)d_res, n_ix,
                                               (const void*)d_index, (SeqDesc*)desc.p, (uint32_t*)seqcnt.p, iprof, src, sw.feed_round);
                        else
                            hipLaunchKernelGGL((k_copy_selffed<FzCfgS8, FzSrcSpx>), dim3(n_ix), dim3(64 * 8), 0, st, j.d_frame, (uint64_t)j.frame_cap, j.d_dst, tbl, (const ResultRec*)d_res, n_ix,
                                               (const void*)d_index, (SeqDesc*)desc.p, (uint32_t*)seqcnt.p, iprof, src, sw.feed_round);
                    } else {
                        const FzSrcIx src{(const void*)d_index, n_ix};
                        if (j.block_size <= (1u << 20))
                            hipLaunchKernelGGL((k_copy_selffed<FzCfgS4, FzSrcIx>), dim3(n_ix), dim3(64 * 4), 0, st, j.d_frame, (uint64_t)j.frame_cap, j.d_dst, tbl, (const ResultRec*)d_res, n_ix,
                                               (const void*)d_index, (SeqDesc*)desc.p, (uint32_t*)seqcnt.p, iprof, src, sw.feed_round);
                        else
                            hipLaunchKernelGGL((k_copy_selffed<FzCfgS8, FzSrcIx>), dim3(n_ix), dim3(64 * 8), 0, st, j.d_frame, (uint64_t)j.frame_cap, j.d_dst, tbl, (const ResultRec*)d_res, n_ix,
                                               (const void*)d_index, (SeqDesc*)desc.p, (uint32_t*)seqcnt.p, iprof, src, sw.feed_round);
                    }
                    tick(9, true);
                    indexed = true;
                    ix_flags = (const uint32_t*)seqcnt.p;
                    plan |= LZ4F_MI355X_PATH_INDEXED;
                } else {
                if (spx_mode)                                                                 // (no entries: the stretches between the blocks' check lines)
                    hipLaunchKernelGGL(k_spx_parse, dim3(n_ix), dim3(128), 0, st, j.d_frame, (uint64_t)j.frame_cap, (const BlockOut*)tbl, (const ResultRec*)d_res, n_ix,
                                       (const void*)d_index, (const SpxPoint*)spx.p, (const uint32_t*)((const uint8_t*)spx.p + (size_t)n_max * (SPX_MAXPT + 1) * sizeof(SpxPoint)),
                                       (SeqDesc*)desc.p, (uint32_t*)seqcnt.p);
                else
                hipLaunchKernelGGL(k_parse_indexed, dim3((n_lanes + 255) / 256), dim3(256), 0, st, j.d_frame, (uint64_t)j.frame_cap,
                                   (const BlockOut*)tbl, (const void*)d_index, n_ix, (SeqDesc*)desc.p, (uint32_t*)seqcnt.p, lk, (uint64_t)j.hist0);
                const uint64_t trace_span = (uint64_t)n_ix * j.block_size;            // (the last block may be short)
                const bool trace_can = !sw.no_trace && !sw.no_resolve && (j.block_size & 63u) == 0;
                // (pointer doubling does ~12 GiB/s on text whatever the framing; hop by hop it is 1.3 GiB/s, which only pays where
                // there is no block-level parallelism - linked frames; independent blocks then stay with the copier workgroups, 4.8 GiB/s)
                const bool can_double = (ix_dense || sw.trace_always) && trace_span <= IXP_MAX_SPAN && !sw.no_doubling;
                uint32_t gate = !trace_can ? 0u : sw.trace_always ? 2u : (j.linked || can_double) ? 1u : 0u;
                if (gate && postab.ensure((size_t)(trace_span >> 6) * 4 + 512 + ((size_t)(trace_span >> IXT_REGION_LOG) + 4) * 4)) gate = 0;      // (no memory for the position table: the copiers do it)
                if (gate && can_double)                                        // (dense by the sequence density: no need to resolve anything)
                    hipLaunchKernelGGL(k_dense_gate, dim3(1), dim3(64), 0, st, (uint32_t*)seqcnt.p, gate, (const BlockOut*)tbl, (const ResultRec*)d_res, n_ix, 1u, 1u);
                uint32_t* dsrc = (uint32_t*)((uint8_t*)desc.p + dsrc_at);
                if (sw.no_resolve) dsrc = nullptr;
                else
                    hipLaunchKernelGGL(k_resolve_direct, dim3(n_ix, j.block_size >= (1u << 19) ? j.block_size >> 18 : 1u), dim3(256), 0, st, d_index, (const BlockOut*)tbl, (const ResultRec*)d_res, n_ix, (const SeqDesc*)desc.p,
                                       dsrc, (uint32_t*)seqcnt.p, (iprof ? 1u : 0u) | (gate ? 2u : 0u), lk);
                if (iprof) {                                                   // developer aid: how many matches are direct
                    uint32_t c[8];
                    if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(c, seqcnt.p, 32, hipMemcpyDeviceToHost) == hipSuccess)
                        fprintf(stderr, "indexed: flags %u, matches direct after parse %u, resolved %u, left to the chain %u\n", c[0], c[4], c[5], c[6]);
                    uint32_t x[6] = {0, 0, 0, 0, 0, 0}; if (hipMemcpy(x, (uint32_t*)seqcnt.p + 10, 24, hipMemcpyDeviceToHost) == hipSuccess) fprintf(stderr, "indexed (linked): %u matches stay on the chain, %u of them reach into the block in front (%u blocks); not resolved because: beyond one block %u, source straddles two runs %u, run-length source %u, hop limit %u\n", x[0], x[1], n_ix, x[2], x[3], x[4], x[5]);
                }
                // dense frames (text): no chain at all, every output byte traced to its literal (see k_trace_copy)
                {
                    hipLaunchKernelGGL(k_dense_gate, dim3(1), dim3(64), 0, st, (uint32_t*)seqcnt.p, gate, (const BlockOut*)tbl, (const ResultRec*)d_res, n_ix, 0u, j.block_size >= (1u << 19) ? j.block_size >> 18 : 1u);
                    if (gate) {
                        hipLaunchKernelGGL(k_build_postab, dim3(n_ix, j.block_size >= (1u << 19) ? j.block_size >> 18 : 1u), dim3(256), 0, st, d_index, (const BlockOut*)tbl, (const ResultRec*)d_res, n_ix,
                                           (const SeqDesc*)desc.p, (uint32_t*)postab.p, (uint32_t*)seqcnt.p);
                        const uint64_t n_thr = (trace_span + IXT_TB - 1) / IXT_TB;
                        // if the last index seen here was of a dense stream (the device decides about THIS one, but the
                        // scratch - 4 bytes per output byte - and 18 launches are the host's to spend): one hop per byte, then pointer doubling
                        const uint32_t pd_grid = (uint32_t)((trace_span / 4 + 255) / 256), pd_ngrp = pd_grid * 4u;      // (k_pd_round: a wave per 256 bytes)
                        const size_t pd_grp_at = (((size_t)trace_span * 4 + 255) & ~(size_t)255) + (((IXP_ROUNDS + 1) * IXP_STRIPES * 4 + 255) & ~(size_t)255);
                        const bool doubling = can_double && !pdbuf.ensure(pd_grp_at + 2 * (size_t)pd_ngrp + 256);
                        if (doubling) {
                            plan |= LZ4F_MI355X_PATH_DOUBLING;
                            uint32_t* remaining = (uint32_t*)((uint8_t*)pdbuf.p + (((size_t)trace_span * 4 + 255) & ~(size_t)255));
                            if (hipMemsetAsync(remaining, 0, (IXP_ROUNDS + 1) * IXP_STRIPES * 4, st) != hipSuccess) return make_err(LZ4F_ERROR_GENERIC);
                            hipLaunchKernelGGL(k_pd_init, dim3((uint32_t)((n_thr + 255) / 256)), dim3(256), 0, st, j.d_frame, (uint64_t)j.frame_cap, j.d_dst, (const BlockOut*)tbl,
                                               (const ResultRec*)d_res, n_ix, d_index, (const SeqDesc*)desc.p, (const uint32_t*)dsrc, (const uint32_t*)postab.p, (uint32_t*)seqcnt.p,
                                               lk & 1u, (uint32_t)j.block_size, (uint64_t)j.hist0, (uint32_t*)pdbuf.p, remaining);
                            for (uint32_t r = 1; r <= IXP_ROUNDS; r++)
                                hipLaunchKernelGGL(k_pd_round, dim3(pd_grid), dim3(256), 0, st, j.d_dst, (uint32_t*)pdbuf.p, (const BlockOut*)tbl,
                                                   (const ResultRec*)d_res, n_ix, r, remaining, (uint32_t*)seqcnt.p, (uint8_t*)pdbuf.p + pd_grp_at, pd_ngrp);
                            hipLaunchKernelGGL(k_pd_verdict, dim3(1), dim3(64), 0, st, (uint32_t*)seqcnt.p, (const uint32_t*)remaining);
                            if (iprof) { static uint32_t t[(IXP_ROUNDS + 1) * IXP_STRIPES]; if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(t, remaining, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess) { fprintf(stderr, "doubling: bytes open after each round:"); for (uint32_t r = 0; r <= IXP_ROUNDS; r++) { uint64_t sum = 0; for (uint32_t q = 0; q < IXP_STRIPES; q++) sum += t[r * IXP_STRIPES + q]; fprintf(stderr, " %llu", (unsigned long long)sum); } fprintf(stderr, "\n"); } }
                        } else {
                        plan |= LZ4F_MI355X_PATH_TRACE_HOPS;
                        uint32_t* region_cnt = (uint32_t*)((uint8_t*)postab.p + (((size_t)(trace_span >> 6) * 4 + 255) & ~(size_t)255));
                        if (hipMemsetAsync(region_cnt, 0, ((size_t)(trace_span >> IXT_REGION_LOG) + 2) * 4, st) != hipSuccess) return make_err(LZ4F_ERROR_GENERIC);
                        const uint32_t tc_wg = (uint32_t)((n_thr + 255) / 256);
                        hipLaunchKernelGGL(k_trace_copy, dim3(std::min<uint32_t>(tc_wg, 8192u)), dim3(256), 0, st, j.d_frame, (uint64_t)j.frame_cap, j.d_dst, (const BlockOut*)tbl,
                                           (const ResultRec*)d_res, n_ix, d_index, (const SeqDesc*)desc.p, (const uint32_t*)dsrc, (const uint32_t*)postab.p, (uint32_t*)seqcnt.p,
                                           lk & 1u, (uint32_t)j.block_size, (uint64_t)j.hist0, region_cnt, iprof ? 1u : 0u, tc_wg);
                        if (iprof) { uint32_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(t, (uint32_t*)seqcnt.p + 24, 32, hipMemcpyDeviceToHost) == hipSuccess && t[0]) fprintf(stderr, "traced: %llu turns for %u pieces (%u read from the output), deepest thread %u turns\n", (unsigned long long)t[2] | ((unsigned long long)t[3] << 32), t[4], t[5], t[6]); }
                        }
                    }
                }
                tick(8, true);
                tick(9, false);
                // (linked frames of small blocks: a workgroup takes a group of consecutive blocks - see k_copy_indexed)
                // (a group is 1 MiB of blocks where that fills the machine - 1024 workgroups of the 4-wave shape are half of its wave slots - and less for
                // smaller frames: LZ4F_MI355X_GROUP_KIB sets it)
                const uint32_t group_bytes = sw.group_kib ? sw.group_kib << 10 : ((uint64_t)n_ix * j.block_size <= (2ull << 30) ? (512u << 10) : (1u << 20));      // (1 GiB: 512 KiB 0.509 ms, 1 MiB 0.541, 256 KiB 0.788, 2 MiB 0.778)
                const uint32_t group = (j.linked && j.block_size < group_bytes && !sw.no_groups) ? group_bytes / j.block_size : 1u;
                const uint32_t n_wg = (n_ix + group - 1) / group;
                // (how long a group of a linked frame waits for the one in front: half a second plus 20 ticks of the 100 MHz clock per
                // output byte - 5 MB/s, a fifth of the slowest chain measured (text, linked, 27 MB/s); LZ4F_MI355X_WAIT_TICKS overrides)
                const uint64_t wait_ticks = sw.wait_ticks ? sw.wait_ticks : 50000000ull + 20ull * n_ix * j.block_size;
                if (j.block_size <= (1u << 20))
                    hipLaunchKernelGGL(k_copy_indexed<FzCfg<4>>, dim3(n_wg), dim3(64 * 4), 0, st, j.d_frame, j.d_dst, tbl, (const ResultRec*)d_res, n_ix,
                                       d_index, (const SeqDesc*)desc.p, (const uint32_t*)dsrc, (uint32_t*)seqcnt.p, iprof, lk, done, group, (uint64_t)j.hist0, wait_ticks);
                else
                    hipLaunchKernelGGL(k_copy_indexed<FzCfg<8>>, dim3(n_wg), dim3(64 * 8), 0, st, j.d_frame, j.d_dst, tbl, (const ResultRec*)d_res, n_ix,
                                       d_index, (const SeqDesc*)desc.p, (const uint32_t*)dsrc, (uint32_t*)seqcnt.p, iprof, lk, done, group, (uint64_t)j.hist0, wait_ticks);
                tick(9, true);
                if (iprof && j.linked) { uint32_t y[4] = {0, 0, 0, 0}; if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(y, (uint32_t*)seqcnt.p + 20, 16, hipMemcpyDeviceToHost) == hipSuccess) fprintf(stderr, "indexed (linked): blocks that found the block in front at state 3: %u (of those, had to wait for all of it: %u); blocks with set-aside matches %u (block in front already done: %u)\n", y[0], y[1], y[2], y[3]); }
                indexed = true;
                ix_flags = (const uint32_t*)seqcnt.p;
                plan |= LZ4F_MI355X_PATH_INDEXED;
                }
            }
        }
        if (mode == 'f') {
            unsigned long long* prof = (unsigned long long*)(sw.prof ? prof_buf() : nullptr);
            // more blocks than the machine has 8-wave workgroup slots: the 4-wave shape keeps twice as many in flight
            const bool small = !j.linked && j.block_size <= (1u << 20);
            // a linked frame is one chain: one workgroup with the 64 KiB window in LDS; frames with short (flushed) blocks
            // set the flag and are decoded by the generic kernel launched right behind (it returns at once otherwise)
            // (one block of a linked frame - what the streaming functions hand over per call: the window kernel is built for whole frames
            // and takes 180-210 us for a single 64 KiB block; the fused workgroup takes it directly)
            const bool windowed = j.linked && j.dst_cap < 0xFFF00000ull && !sw.no_window && !(n_max == 1 && (j.d_table || j.table_in_place || j.table_direct));
            const uint32_t* only_if = indexed ? (const uint32_t*)seqcnt.p : nullptr;      // behind the indexed kernels the generic ones only run if they gave up
            if (windowed) {
                plan |= LZ4F_MI355X_PATH_WINDOW;
                if (!indexed && seqcnt.ensure(256)) return make_err(LZ4F_ERROR_allocation_failed);
                uint32_t* fb = (uint32_t*)seqcnt.p + (indexed ? 16 : 0);                  // (word 0 is the indexed kernels' flag)
                hipLaunchKernelGGL(k_decode_linked, dim3(1), dim3(64 * LK_WAVES), 0, st, j.d_frame, j.d_dst, j.dst_cap, tbl,
                                   (const ResultRec*)d_res, n_max, j.block_size, j.hist0, fb, only_if);
                only_if = fb;
                if (sw.prof) {                              // developer aid: why the windowed kernel stopped, if it did
                    uint32_t dbg[3] = {0, 0, 0};
                    if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(dbg, fb, 12, hipMemcpyDeviceToHost) == hipSuccess)
                        fprintf(stderr, "k_decode_linked: fallback %u why %u block %u\n", dbg[0], dbg[1], dbg[2]);
                }
            }
            plan |= LZ4F_MI355X_PATH_FUSED;
            // big independent blocks and no index to go by: a look at the payload decides between the fused workgroups and - dense data -
            // the wave-per-block decoder (k_density_probe); both are launched, one of them returns at once
            if (!indexed && !j.linked && !sw.no_density_probe) {
                if (density.ensure(64)) return make_err(LZ4F_ERROR_allocation_failed);
                hipLaunchKernelGGL(k_density_probe, dim3(1), dim3(64), 0, st, j.d_frame, (uint64_t)j.frame_cap, (const BlockOut*)tbl, (const ResultRec*)d_res, n_max, (uint32_t*)density.p);
                // Few big blocks: a wave per block leaves the machine idle and waits out every trip to memory (13-15 GiB/s for a GiB in 4 MiB blocks);
                // a workgroup per block with the block's window in LDS and its waves taking the payload in turns (decode_relay.cuh) is three times
                // as fast per block - but it has a CU to itself, so from ~3 blocks per CU on the waves win again (8 GiB in 4 MiB blocks: 90 GiB/s).
                if (j.block_size > 65536u && sw.dense_mode != 2 && (sw.dense_mode == 1 || n_max <= 3u * device_cus(device))) {
                    plan |= LZ4F_MI355X_PATH_WORKGROUP_PER_BLOCK;
                    hipLaunchKernelGGL((k_decode_blocks_relay<RELAY_W, RELAY_S>), dim3(n_max), dim3(64 * (RELAY_W + RELAY_S + 3)), 0, st, j.d_frame, j.d_dst, tbl, (const ResultRec*)d_res, n_max, (uint64_t)j.frame_cap,
                                       (const uint32_t*)density.p);
                } else
                hipLaunchKernelGGL((k_decode_blocks<W>), dim3((n_max + W - 1) / W), dim3(64 * W), 0, st, j.d_frame, j.d_dst, j.dst_cap, tbl, (const ResultRec*)d_res,
                                   n_max, 0u, j.block_size, j.hist0, (uint64_t)j.frame_cap, (const uint32_t*)density.p);
                only_if = (const uint32_t*)density.p + 1;
                plan |= LZ4F_MI355X_PATH_WAVE_PER_BLOCK;
            }
            if (small)
                hipLaunchKernelGGL(k_decode_blocks_fused<FzCfg<4>>, dim3(n_max), dim3(64 * 4), 0, st, j.d_frame, j.d_dst, j.dst_cap, tbl,
                                   (const ResultRec*)d_res, n_max, 0u, j.block_size, j.hist0, prof, only_if);
            else
                hipLaunchKernelGGL(k_decode_blocks_fused<FzCfg<8>>, dim3(j.linked ? 1u : n_max), dim3(64 * 8), 0, st, j.d_frame, j.d_dst, j.dst_cap, tbl,
                                   (const ResultRec*)d_res, n_max, j.linked ? 1u : 0u, j.block_size, j.hist0, prof, only_if);
        } else {
            plan |= LZ4F_MI355X_PATH_WAVE_PER_BLOCK;
            hipLaunchKernelGGL((k_decode_blocks<W>), dim3(grid), dim3(64 * W), sw.dblk_lds, st, j.d_frame, j.d_dst, j.dst_cap, tbl, (const ResultRec*)d_res,
                               n_max, j.linked ? 1u : 0u, j.block_size, j.hist0, (uint64_t)j.frame_cap);
        }
        tick(6, true);
    }
    if (aux_pending) { HIP_TRY(hipStreamWaitEvent(st, (hipEvent_t)ev_join, 0)); aux_pending = false; }
    tick(7, false);
    const bool check_here = n_max <= 64;                              // (the finishing wave looks at a few blocks itself: a launch less)
    if (n_max && !check_here) hipLaunchKernelGGL(k_finish_check, dim3((n_max + 255) / 256), dim3(256), 0, st, (const BlockOut*)tbl, (const ResultRec*)d_res, n_max, j.linked ? 1u : 0u, j.block_size, (uint32_t*)bad.p);
    hipLaunchKernelGGL(k_finish_decode, dim3(1), dim3(64), 0, st, j.d_dst, tbl, (ResultRec*)d_res, n_max, j.linked ? 1u : 0u, j.block_size,
                       (const uint32_t*)bad.p, j.block_checksum ? 1u : 0u, plan, ix_flags, check_here ? 1u : 0u, (uint64_t)j.dst_cap);
    if (j.content_checksum && !j.d_table && !j.table_in_place && !j.table_direct && !sw.no_content_check)      // (a whole frame was walked: res->consumed is behind its checksum word)
        hipLaunchKernelGGL(k_xxh32_content, dim3(1), dim3(64), 0, st, (const uint8_t*)j.d_dst, 0ull, (uint8_t*)j.d_frame, (ResultRec*)d_res, 1u);
    tick(7, true);
    tick(11, true);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// host-pointer helpers
// The host-pointer calls are bounded by how fast bytes move between the caller's (pageable) buffers and the pinned staging
// buffers: one thread's memcpy is ~10 GB/s, a fifth of what the PCIe link takes.  Large copies are split over a few threads.
static void big_memcpy(void* dst, const void* src, size_t n)
{
    const size_t MIN_PER_THREAD = (size_t)8 << 20;
    unsigned hw = std::thread::hardware_concurrency();
    size_t t = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 8), n / MIN_PER_THREAD);
    if (t <= 1) { memcpy(dst, src, n); return; }
    const size_t per = ((n / t) + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (size_t i = 1; i < t; i++) {
        const size_t a = i * per;
        if (a >= n) break;
        const size_t len = std::min(per, n - a);
        th.emplace_back([=] { memcpy((uint8_t*)dst + a, (const uint8_t*)src + a, len); });
    }
    memcpy(dst, src, std::min(per, n));
    for (auto& x : th) x.join();
}

// Host buffer <-> device through the pinned staging buffer, in pieces: the copy between the caller's pageable memory and the
// staging buffer (CPU threads) of one piece runs while the DMA of the piece before is in flight, instead of one after the other.
// Is `p` page-locked memory the DMA engines can reach directly (hipHostMalloc / hipHostRegister: lz4f_mi355x_host_alloc,
// the conduits' batch buffers)?  Then no staging copy is needed.
// One upload and one download at a time per device.  Engines that share a device share its host link: two uploads side by
// side each take twice as long, and - symmetric as they are - the engines then also download side by side, so the link is never
// busy in both directions (measured: 32 GiB/s).  With a token per direction they fall out of step by themselves: one engine's
// upload runs beside the other's kernels and download.
namespace { std::mutex g_up_mu[16], g_down_mu[16]; }
static std::mutex& up_token(int device) { return g_up_mu[(unsigned)device % 16]; }
static std::mutex& down_token(int device) { return g_down_mu[(unsigned)device % 16]; }

bool lz4f::is_pinned_host(const void* p)
{
    hipPointerAttribute_t a; memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
static const size_t XFER_PIECE = (size_t)32 << 20;
static hipError_t staged_h2d(void* d_dst, void* pinned, const void* src, size_t n, hipStream_t st)
{
    for (size_t a = 0; a < n; a += XFER_PIECE) {
        const size_t len = std::min(XFER_PIECE, n - a);
        big_memcpy((uint8_t*)pinned + a, (const uint8_t*)src + a, len);
        hipError_t e = hipMemcpyAsync((uint8_t*)d_dst + a, (uint8_t*)pinned + a, len, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
static hipError_t staged_d2h(void* dst, void* pinned, const void* d_src, size_t n, hipStream_t st)
{
    const size_t np = (n + XFER_PIECE - 1) / XFER_PIECE;
    std::vector<hipEvent_t> ev(np);
    hipError_t e = hipSuccess;
    size_t made = 0;
    for (size_t i = 0; i < np && e == hipSuccess; i++) {
        const size_t a = i * XFER_PIECE, len = std::min(XFER_PIECE, n - a);
        e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
        if (e != hipSuccess) break;
        made++;
        e = hipMemcpyAsync((uint8_t*)pinned + a, (const uint8_t*)d_src + a, len, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipEventRecord(ev[i], st);
    }
    for (size_t i = 0; i < made; i++) {
        if (e == hipSuccess) e = hipEventSynchronize(ev[i]);
        if (e == hipSuccess) { const size_t a = i * XFER_PIECE, len = std::min(XFER_PIECE, n - a); big_memcpy((uint8_t*)dst + a, (uint8_t*)pinned + a, len); }
        (void)hipEventDestroy(ev[i]);
    }
    if (e != hipSuccess) (void)hipStreamSynchronize(st);
    return e;
}

size_t lz4f_mi355x_engine::slab_compress(const uint8_t* src, size_t n, const uint8_t* hist, size_t hist_len, uint32_t block_size, bool linked,
                                         bool block_checksum, bool src_pinned, size_t* size)
{
    *size = 0;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    if (!linked) hist_len = 0;
    if (hist_len > 65536) { hist += hist_len - 65536; hist_len = 65536; }
    const size_t total = hist_len + n;
    const size_t nblocks = (n + block_size - 1) / block_size;
    const size_t out_cap = n + nblocks * 8 + 64;
    if ((!src_pinned && h_in.ensure(total)) || h_small.ensure(65536 + 256) || d_in.ensure(total + 64) || d_out.ensure(out_cap) || res.ensure(sizeof(ResultRec)))
        return make_err(LZ4F_ERROR_allocation_failed);
    {
        std::lock_guard<std::mutex> up(up_token(device));
        if (hist_len) {
            if (src_pinned && hist + hist_len == src) HIP_TRY(hipMemcpyAsync(d_in.p, hist, hist_len, hipMemcpyHostToDevice, st));      // (the history sits in front of the input, in the same pinned buffer)
            else { memcpy(h_small.p, hist, hist_len); HIP_TRY(hipMemcpyAsync(d_in.p, h_small.p, hist_len, hipMemcpyHostToDevice, st)); }
        }
        if (src_pinned) HIP_TRY(hipMemcpyAsync((uint8_t*)d_in.p + hist_len, src, n, hipMemcpyHostToDevice, st));
        else HIP_TRY(staged_h2d((uint8_t*)d_in.p + hist_len, (uint8_t*)h_in.p + hist_len, src, n, st));
        if (n >= ((size_t)8 << 20)) HIP_TRY(hipStreamSynchronize(st));      // (bulk slabs: hold the token until the bytes are over)
    }
    CompressJob j; memset(&j, 0, sizeof(j));
    j.d_src = (const uint8_t*)d_in.p; j.src_size = total; j.first_off = hist_len; j.block_size = block_size;
    j.linked = linked; j.block_checksum = block_checksum; j.endmark = false; j.header_size = 0;
    size_t r = launch_compress(j, (uint8_t*)d_out.p, out_cap, (lz4f_mi355x_result*)res.p, nullptr);
    if (is_err(r)) return r;
    ResultRec* hr = (ResultRec*)((uint8_t*)h_small.p + 65536 + 64);
    HIP_TRY(hipMemcpyAsync(hr, res.p, sizeof(ResultRec), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (hr->status != ST_OK) { set_last_error("device compress status %u", hr->status); return make_err((int)hr->status); }
    *size = hr->size;
    return 0;
}

size_t lz4f_mi355x_engine::compress_block_pinned(const uint8_t* pin_src, size_t hist_len, size_t n, uint32_t block_size, bool linked, bool block_checksum,
                                                 uint8_t* pin_dst, size_t dst_cap, void* pin_res, size_t* size)
{
    // (Kernels reading the staging buffer through the link themselves - no copies at all - were tried first: 510 us per 64 KiB block
    // against 190 us with copies.  A kernel's scattered 16-byte reads over PCIe are not what a DMA engine's are.)
    *size = 0;
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    if (!linked) hist_len = 0;
    const size_t total = hist_len + n;
    const size_t out_cap = n + ((n + block_size - 1) / block_size) * 8 + 64;
    const size_t res_at = (out_cap + 63) & ~(size_t)63;                  // the result record rides behind the blocks: one copy back
    if (out_cap > dst_cap) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
    if (d_in.ensure(total + 64) || d_out.ensure(res_at + sizeof(ResultRec) + 64)) return make_err(LZ4F_ERROR_allocation_failed);
    HIP_TRY(hipMemcpyAsync(d_in.p, pin_src, total, hipMemcpyHostToDevice, st));
    CompressJob j; memset(&j, 0, sizeof(j));
    j.d_src = (const uint8_t*)d_in.p; j.src_size = total; j.first_off = hist_len; j.block_size = block_size;
    j.linked = linked; j.block_checksum = block_checksum; j.endmark = false; j.header_size = 0;
    lz4f_mi355x_result* d_res = (lz4f_mi355x_result*)((uint8_t*)d_out.p + res_at);
    size_t r = launch_compress(j, (uint8_t*)d_out.p, out_cap, d_res, nullptr);
    if (is_err(r)) return r;
    // (the blocks' size is not known on the host yet: everything up to the record comes back - at most a block and a few bytes)
    HIP_TRY(hipMemcpyAsync(pin_dst, d_out.p, res_at + sizeof(ResultRec), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const ResultRec* hr = (const ResultRec*)(pin_dst + res_at);
    if (hr->status != ST_OK) { set_last_error("device compress status %u", hr->status); return make_err((int)hr->status); }
    *size = hr->size;
    (void)pin_res;
    return 0;
}

size_t lz4f_mi355x_engine::slab_fetch(uint8_t* dst, size_t size, size_t d_off, bool dst_pinned)
{
    if (size == 0) return 0;
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    if (!dst_pinned && h_out.ensure(size + 64)) return make_err(LZ4F_ERROR_allocation_failed);
    std::lock_guard<std::mutex> down(down_token(device));
    if (dst_pinned) { HIP_TRY(hipMemcpyAsync(dst, (const uint8_t*)d_out.p + d_off, size, hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st)); return 0; }
    HIP_TRY(staged_d2h(dst, h_out.p, (const uint8_t*)d_out.p + d_off, size, st));
    return 0;
}

size_t lz4f_mi355x_engine::compress_blocks_host(const uint8_t* src, size_t n, const uint8_t* hist, size_t hist_len,
                                                uint32_t block_size, bool linked, bool block_checksum, uint8_t* dst, size_t dst_cap, size_t* written)
{
    *written = 0;
    size_t size = 0;
    size_t r = slab_compress(src, n, hist, hist_len, block_size, linked, block_checksum, false, &size);
    if (is_err(r)) return r;
    if (size > dst_cap) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
    r = slab_fetch(dst, size, 0, false);
    if (is_err(r)) return r;
    *written = size;
    return 0;
}

static size_t status_to_err(uint32_t st)
{
    return st == ST_OK ? 0 : make_err((int)st);
}

size_t lz4f_mi355x_engine::slab_decode(const uint8_t* frame_part, size_t part_len, const std::vector<lz4f_mi355x_block>& entries,
                                       const ParsedHeader& ph, const uint8_t* hist, size_t hist_len, bool src_pinned, size_t* got,
                                       uint8_t* fetch_to, size_t fetch_room)
{   // fetch_to (one block of at most 256 KiB - the streaming API's call): the output comes back with the result record, before the ONE
    // synchronisation of the call - a block's worth is copied whatever the block decodes to, and what it did decode to goes to fetch_to
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const bool linked = ph.info.blockMode == LZ4F_blockLinked;
    if (!linked) hist_len = 0;
    const size_t nb = entries.size();
    // the device buffer always has room for every block at full size: blocks are decoded at provisional positions and
    // compacted when some are short (frames written with LZ4F_flush); only what is actually produced must fit the caller's buffer
    const size_t out_room = nb * ph.max_block;
    const size_t tbytes = nb * sizeof(BlockOut);
    // One block of at most 256 KiB out of pageable memory (the streaming API's call): table, payload and history go up in ONE copy, laid out
    // [table | payload | history] in front of the output - three copies of a few KiB each cost more in launches than in bytes
    const bool one_up = fetch_to && nb == 1 && !src_pinned && ph.max_block <= (256u << 10);
    const size_t up_pay = 64, up_hist = (up_pay + part_len + 63 + 64) & ~(size_t)63, up_out = (up_hist + hist_len + 63) & ~(size_t)63;      // (history right-aligned in front of the output)
    if (one_up) {
        if (h_in.ensure(up_out + 256) || d_out.ensure(up_out + out_room + 64) || res.ensure(sizeof(ResultRec))) return make_err(LZ4F_ERROR_allocation_failed);
        uint8_t* hp = (uint8_t*)h_in.p;
        memcpy(hp, entries.data(), tbytes);
        memcpy(hp + up_pay, frame_part, part_len);
        if (hist_len) memcpy(hp + up_out - hist_len, hist, hist_len);
        { std::lock_guard<std::mutex> up(up_token(device)); HIP_TRY(hipMemcpyAsync(d_out.p, hp, up_out, hipMemcpyHostToDevice, st)); }
        DecompressJob j; memset(&j, 0, sizeof(j));
        j.d_frame = (const uint8_t*)d_out.p + up_pay; j.frame_cap = part_len; j.d_dst = (uint8_t*)d_out.p + up_out; j.dst_cap = out_room; j.hist0 = hist_len;
        j.block_size = (uint32_t)ph.max_block; j.linked = linked; j.block_checksum = ph.info.blockChecksumFlag != 0;
        j.table_direct = (lz4f_mi355x_block*)d_out.p; j.n_blocks = 1; j.max_blocks = 1;
        size_t r = launch_decompress(j, (lz4f_mi355x_result*)res.p);
        if (is_err(r)) return r;
        ResultRec* hr = (ResultRec*)(hp + up_out + 64);
        if (h_out.ensure(ph.max_block + 64)) return make_err(LZ4F_ERROR_allocation_failed);
        HIP_TRY(hipMemcpyAsync(h_out.p, j.d_dst, ph.max_block, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(hr, res.p, sizeof(ResultRec), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (hr->status != ST_OK) { set_last_error("device decode status %u at block %u", hr->status, hr->first_bad_block); return status_to_err(hr->status); }
        *got = hr->size;
        if (hr->size > fetch_room) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
        memcpy(fetch_to, h_out.p, hr->size);
        return 0;
    }
    if (h_in.ensure((src_pinned ? 0 : part_len) + tbytes + hist_len + 128) || d_in.ensure(part_len + 64) || d_out.ensure(hist_len + out_room + 64) ||
        res.ensure(sizeof(ResultRec)) || table.ensure((nb + 1) * sizeof(BlockOut)))
        return make_err(LZ4F_ERROR_allocation_failed);
    uint8_t* hp = (uint8_t*)h_in.p;
    const size_t at_tab = src_pinned ? 0 : part_len;
    memcpy(hp + at_tab, entries.data(), tbytes);
    if (hist_len) memcpy(hp + at_tab + tbytes, hist, hist_len);
    {
        std::lock_guard<std::mutex> up(up_token(device));
        if (src_pinned) HIP_TRY(hipMemcpyAsync(d_in.p, frame_part, part_len, hipMemcpyHostToDevice, st));
        else HIP_TRY(staged_h2d(d_in.p, hp, frame_part, part_len, st));
        HIP_TRY(hipMemcpyAsync(table.p, hp + at_tab, tbytes, hipMemcpyHostToDevice, st));
        if (hist_len) HIP_TRY(hipMemcpyAsync(d_out.p, hp + at_tab + tbytes, hist_len, hipMemcpyHostToDevice, st));
        if (part_len >= ((size_t)8 << 20)) HIP_TRY(hipStreamSynchronize(st));
    }
    DecompressJob j; memset(&j, 0, sizeof(j));
    j.d_frame = (const uint8_t*)d_in.p; j.frame_cap = part_len; j.d_dst = (uint8_t*)d_out.p + hist_len; j.dst_cap = out_room; j.hist0 = hist_len;
    j.block_size = (uint32_t)ph.max_block; j.linked = linked; j.block_checksum = ph.info.blockChecksumFlag != 0;
    j.table_in_place = true; j.n_blocks = (uint32_t)nb; j.max_blocks = (uint32_t)nb;
    size_t r = launch_decompress(j, (lz4f_mi355x_result*)res.p);
    if (is_err(r)) return r;
    ResultRec* hr = (ResultRec*)(hp + at_tab + tbytes + hist_len + 8 - ((at_tab + tbytes + hist_len) & 7) + 8);
    const bool with_out = fetch_to && nb == 1 && ph.max_block <= (256u << 10) && !h_out.ensure(ph.max_block + 64);
    if (with_out) HIP_TRY(hipMemcpyAsync(h_out.p, (const uint8_t*)d_out.p + hist_len, ph.max_block, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hr, res.p, sizeof(ResultRec), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (hr->status != ST_OK) { set_last_error("device decode status %u at block %u", hr->status, hr->first_bad_block); return status_to_err(hr->status); }
    *got = hr->size;
    if (fetch_to) {
        if (hr->size > fetch_room) return make_err(LZ4F_ERROR_dstMaxSize_tooSmall);
        if (with_out) memcpy(fetch_to, h_out.p, hr->size);
        else { const size_t r2 = slab_fetch(fetch_to, hr->size, hist_len, false); if (is_err(r2)) return r2; }
    }
    return 0;
}

size_t lz4f_mi355x_engine::run_decode_slab(const uint8_t* frame_part, size_t part_len, const std::vector<lz4f_mi355x_block>& entries,
                                           const ParsedHeader& ph, const uint8_t* hist, size_t hist_len, uint8_t* dst, size_t dst_room, size_t* got)
{
    const bool linked = ph.info.blockMode == LZ4F_blockLinked;
    if (!linked) hist_len = 0;
    size_t n = 0;
    size_t r = slab_decode(frame_part, part_len, entries, ph, hist, hist_len, false, &n, dst, dst_room);
    if (is_err(r)) return r;
    *got = n;
    return 0;
}

size_t lz4f_mi355x_engine::decompress_block_host(const uint8_t* payload, uint32_t csize, bool bck, const uint8_t* hist, size_t hist_len,
                                                 uint8_t* dst, uint32_t dst_cap, bool linked, uint32_t block_size, uint32_t* decoded)
{
    ParsedHeader ph; memset(&ph, 0, sizeof(ph));
    ph.max_block = block_size;
    ph.info.blockMode = linked ? LZ4F_blockLinked : LZ4F_blockIndependent;
    ph.info.blockChecksumFlag = bck ? LZ4F_blockChecksumEnabled : LZ4F_noBlockChecksum;
    std::vector<lz4f_mi355x_block> e(1);
    e[0].src_off = 0; e[0].dst_off = 0; e[0].word = csize; e[0].dst_size = dst_cap < block_size ? dst_cap : block_size;
    size_t got = 0;
    size_t r = run_decode_slab(payload, (size_t)csize + (bck ? 4 : 0), e, ph, hist, hist_len, dst, dst_cap, &got);
    if (is_err(r)) return r;
    *decoded = (uint32_t)got;
    return 0;
}

size_t lz4f_mi355x_engine::decompress_frame_host(const uint8_t* s, size_t n, const ParsedHeader& ph, uint8_t* dst, size_t cap,
                                                 size_t* decoded, size_t* consumed)
{
    const size_t SLAB_SRC = (size_t)256 << 20, SLAB_DST = (size_t)512 << 20;
    const size_t crc = ph.info.blockChecksumFlag ? 4 : 0;
    const bool linked = ph.info.blockMode == LZ4F_blockLinked;
    auto rd32 = [&](size_t at) { return (uint32_t)s[at] | ((uint32_t)s[at + 1] << 8) | ((uint32_t)s[at + 2] << 16) | ((uint32_t)s[at + 3] << 24); };
    size_t pos = ph.header_size, out = 0;
    bool end = false;
    std::vector<lz4f_mi355x_block> entries;
    while (!end) {
        entries.clear();
        const size_t slab_src = pos; size_t prov = 0;
        for (;;) {
            if (n - pos < 4) return make_err(LZ4F_ERROR_frameHeader_incomplete);
            const uint32_t w = rd32(pos);
            if (w == 0) { end = true; break; }
            const size_t csz = w & 0x7FFFFFFFu;
            if (csz > ph.max_block) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
            if (n - pos - 4 < csz + crc) return make_err(LZ4F_ERROR_frameHeader_incomplete);
            if (!entries.empty() && ((pos - slab_src) + 4 + csz + crc > SLAB_SRC || prov + ph.max_block > SLAB_DST)) break;
            lz4f_mi355x_block e;
            e.src_off = pos + 4 - slab_src; e.dst_off = prov; e.word = w;
            e.dst_size = (uint32_t)ph.max_block;
            entries.push_back(e);
            pos += 4 + csz + crc; prov += ph.max_block;
        }
        if (!entries.empty()) {
            size_t got = 0;
            const size_t hl = linked ? std::min(out, (size_t)65536) : 0;
            size_t r = run_decode_slab(s + slab_src, pos - slab_src, entries, ph, dst + out - hl, hl, dst + out, cap - out, &got);
            if (is_err(r)) return r;
            out += got;
        }
    }
    pos += 4;                                                   // EndMark
    if (ph.info.contentSize && ph.info.contentSize != out) return make_err(LZ4F_ERROR_frameSize_wrong);
    if (ph.info.contentChecksumFlag) {
        if (n - pos < 4) return make_err(LZ4F_ERROR_frameHeader_incomplete);
        if (rd32(pos) != xxh32_host(dst, out)) return make_err(LZ4F_ERROR_contentChecksum_invalid);   // serial by construction: host
        pos += 4;
    }
    *decoded = out; *consumed = pos;
    return 0;
}

// ---- the trailer's block list made on the host (same bytes k_trailer_plan / k_trailer_copy write for a frame without a sequence index) ----
namespace lz4f {
bool BlockList::add_blocks(const uint8_t* b, size_t n, uint64_t frame_off, bool bck)
{
    size_t pos = 0;
    while (pos < n) {
        if (n - pos < 4) return false;
        const uint32_t w = (uint32_t)b[pos] | ((uint32_t)b[pos + 1] << 8) | ((uint32_t)b[pos + 2] << 16) | ((uint32_t)b[pos + 3] << 24);
        const size_t step = 4 + (size_t)(w & 0x7FFFFFFFu) + (bck ? 4 : 0);
        if (w == 0 || step > n - pos) return false;
        at.push_back(frame_off + pos);
        pos += step;
    }
    return true;
}
size_t host_trailer_size(uint64_t F, uint64_t n_blocks)
{
    if (n_blocks == 0) return 0;
    if (n_blocks > 0x7FFFFFFFull) return make_err(LZ4F_ERROR_frameSize_wrong);
    const uint64_t list_at = (F + 8 + 15) & ~(uint64_t)15, n_list = (n_blocks + 1) & ~1ull;
    const uint64_t total = list_at + n_list * 8 + sizeof(TrailerFoot) - F;
    if (total - 8 >= 0xFFFFFFFFull) return make_err(LZ4F_ERROR_frameSize_wrong);        // (a skippable frame's size field is 32 bits)
    return (size_t)total;
}
void host_write_trailer(uint8_t* t, uint64_t F, const uint64_t* at, uint32_t n_blocks)
{
    const uint64_t list_at = (F + 8 + 15) & ~(uint64_t)15, n_list = ((uint64_t)n_blocks + 1) & ~1ull, ix_at = list_at + n_list * 8;
    const uint64_t total = ix_at + sizeof(TrailerFoot) - F;
    const uint32_t sz = (uint32_t)(total - 8);
    t[0] = 0x5E; t[1] = 0x2A; t[2] = 0x4D; t[3] = 0x18; t[4] = (uint8_t)sz; t[5] = (uint8_t)(sz >> 8); t[6] = (uint8_t)(sz >> 16); t[7] = (uint8_t)(sz >> 24);
    memset(t + 8, 0, (size_t)(list_at - F - 8));
    for (uint64_t i = 0; i < n_list; i++) { const uint64_t v = i < n_blocks ? at[i] : 0; memcpy(t + (list_at - F) + i * 8, &v, 8); }
    const TrailerFoot f{0u, 0u, 0u, 0u, TR_FOOT, n_blocks, total};
    memcpy(t + (ix_at - F), &f, sizeof(f));
}
}  // namespace lz4f

// ------------------------------------------------------------------------------------------------
// C ABI: engine + device-pointer entry points
extern "C" {

const char* lz4f_mi355x_last_error(void) { return lz4f::last_error(); }

int lz4f_mi355x_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t lz4f_mi355x_set_device(int device)
{
    int n = lz4f_mi355x_device_count();
    if (device < 0 || device >= n) { set_last_error("device %d out of range (%d devices)", device, n); return make_err(LZ4F_ERROR_GENERIC); }
    lz4f::t_device = device;
    return 0;
}

size_t lz4f_mi355x_engine_create(lz4f_mi355x_engine** out, int device, void* hipStream, int borrowStream)
{
    if (!out) return make_err(LZ4F_ERROR_GENERIC);
    return new_engine(out, device, hipStream, borrowStream != 0);
}
size_t lz4f_mi355x_engine_free(lz4f_mi355x_engine* e) { delete e; return 0; }
void lz4f_mi355x_release_engines(void) { lz4f::release_idle_engines(); }
void* lz4f_mi355x_host_alloc(size_t size)
{
    void* p = nullptr;
    if (hipSetDevice(lz4f::selected_device()) != hipSuccess || hipHostMalloc(&p, size ? size : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); set_last_error("hipHostMalloc(%zu) failed", size); return nullptr; }
    return p;
}
void lz4f_mi355x_host_free(void* p) { if (p) (void)hipHostFree(p); }
void* lz4f_mi355x_engine_stream(lz4f_mi355x_engine* e) { return e ? e->stream : nullptr; }

size_t lz4f_mi355x_engine_set_deterministic(lz4f_mi355x_engine* e, int enable)
{
    if (!e) return make_err(LZ4F_ERROR_GENERIC);
    if (enable) e->sw.e1_solo |= 1u; else e->sw.e1_solo &= ~1u;
    return 0;
}
size_t lz4f_mi355x_engine_set_timing(lz4f_mi355x_engine* e, int enable)
{
    if (!e) return make_err(LZ4F_ERROR_GENERIC);
    e->timing = enable != 0;
    return 0;
}
size_t lz4f_mi355x_engine_get_timing(lz4f_mi355x_engine* e, float* ms) { return lz4f_mi355x_engine_get_timing_n(e, ms, LZ4F_MI355X_TIMING_SLOTS_V1); }
size_t lz4f_mi355x_engine_get_timing_n(lz4f_mi355x_engine* e, float* ms, size_t n)
{
    if (!e || !ms) return make_err(LZ4F_ERROR_GENERIC);
    size_t r = e->sync();
    if (is_err(r)) return r;
    for (int s = 0; s < LZ4F_MI355X_TIMING_SLOTS && (size_t)s < n; s++) {
        ms[s] = 0.f;
        if (e->ev_used[s] && e->ev[2 * s] && e->ev[2 * s + 1]) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, (hipEvent_t)e->ev[2 * s], (hipEvent_t)e->ev[2 * s + 1]) == hipSuccess) ms[s] = t;
        }
    }
    return 0;
}

size_t lz4f_mi355x_dev_workspace_size(size_t srcSize, const LZ4F_preferences_t* prefs)
{
    size_t bs = block_size_of(prefs ? prefs->frameInfo.blockSizeID : 0);
    if (!bs) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
    uint32_t ch = pick_chunk_size((uint32_t)bs);
    size_t nchunks = (srcSize + bs - 1) / bs * (bs / ch) + 1;
    // (the record pool as an engine with default switches sizes it: 12288 records of 8 bytes per 64 KiB tile = 1.5 bytes per input byte)
    return nchunks * sizeof(ChunkInfo) + (size_t)(rec_pool_at((uint32_t)nchunks) + rec_pool_records((uint32_t)nchunks, ch / 4 + 1, 0)) * 8
           + ((srcSize + bs - 1) / bs + 1) * (sizeof(BlockOut) + 4) + 4096;
}

size_t lz4f_mi355x_dev_compressFrame(lz4f_mi355x_engine* e, void* d_dst, size_t dstCapacity, const void* d_src, size_t srcSize,
                                     const LZ4F_preferences_t* prefs, lz4f_mi355x_result* d_result, lz4f_mi355x_block* d_table)
{
    if (!e) return make_err(LZ4F_ERROR_GENERIC);
    LZ4F_preferences_t p; memset(&p, 0, sizeof(p));
    if (prefs) p = *prefs;
    if (p.frameInfo.blockSizeID == 0) p.frameInfo.blockSizeID = LZ4F_max64KB;
    const size_t bs = block_size_of(p.frameInfo.blockSizeID);
    if (!bs) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
    if (p.compressionLevel > 2) { set_last_error("only the fast encoder (level <= 2) exists"); return make_err(LZ4F_ERROR_compressionLevel_invalid); }
    if (p.frameInfo.contentSize && p.frameInfo.contentSize != srcSize) return make_err(LZ4F_ERROR_frameSize_wrong);
    lz4f_mi355x_engine::CompressJob j; memset(&j, 0, sizeof(j));
    j.d_src = (const uint8_t*)d_src; j.src_size = srcSize; j.first_off = 0; j.block_size = (uint32_t)bs;
    j.linked = p.frameInfo.blockMode == LZ4F_blockLinked; j.block_checksum = p.frameInfo.blockChecksumFlag != 0; j.endmark = true;
    j.content_checksum = p.frameInfo.contentChecksumFlag != 0;
    j.header_size = (uint32_t)write_frame_header(j.header, p);
    return e->launch_compress(j, (uint8_t*)d_dst, dstCapacity, d_result, d_table);
}

size_t lz4f_mi355x_dev_index_size(size_t srcSize, const LZ4F_preferences_t* prefs)
{
    size_t bs = block_size_of(prefs ? prefs->frameInfo.blockSizeID : 0);
    if (!bs) bs = 65536;
    const uint32_t ch = pick_chunk_size((uint32_t)bs);
    const uint32_t nb = (uint32_t)((srcSize + bs - 1) / bs), cpb = (uint32_t)(bs / ch);
    return ix_entries_at(nb, cpb) + ix_typical_entries(srcSize, nb * cpb) * sizeof(IxEntry);
}

size_t lz4f_mi355x_trailer_bound(size_t srcSize, const LZ4F_preferences_t* prefs)
{
    size_t bs = block_size_of(prefs ? prefs->frameInfo.blockSizeID : 0);
    if (!bs) bs = 65536;
    return lz4f_mi355x_dev_index_size(srcSize, prefs) + ((srcSize + bs - 1) / bs + 2) * 8 + 128;
}

size_t lz4f_mi355x_dev_compressFrameIndexed(lz4f_mi355x_engine* e, void* d_dst, size_t dstCapacity, const void* d_src, size_t srcSize,
                                            const LZ4F_preferences_t* prefs, lz4f_mi355x_result* d_result, lz4f_mi355x_block* d_table, void* d_index,
                                            size_t indexCapacity)
{
    const bool inband = d_index == nullptr && indexCapacity == LZ4F_MI355X_INBAND;
    if (!e || (!inband && (!d_table || !d_result))) return make_err(LZ4F_ERROR_GENERIC);
    LZ4F_preferences_t p; memset(&p, 0, sizeof(p));
    if (prefs) p = *prefs;
    if (p.frameInfo.blockSizeID == 0) p.frameInfo.blockSizeID = LZ4F_max64KB;
    const size_t bs = block_size_of(p.frameInfo.blockSizeID);
    if (!bs) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
    if (p.compressionLevel > 2) { set_last_error("only the fast encoder (level <= 2) exists"); return make_err(LZ4F_ERROR_compressionLevel_invalid); }
    if (p.frameInfo.contentSize && p.frameInfo.contentSize != srcSize) return make_err(LZ4F_ERROR_frameSize_wrong);
    lz4f_mi355x_engine::CompressJob j; memset(&j, 0, sizeof(j));
    j.d_src = (const uint8_t*)d_src; j.src_size = srcSize; j.first_off = 0; j.block_size = (uint32_t)bs;
    j.linked = p.frameInfo.blockMode == LZ4F_blockLinked; j.block_checksum = p.frameInfo.blockChecksumFlag != 0; j.endmark = true;
    j.content_checksum = p.frameInfo.contentChecksumFlag != 0;
    j.header_size = (uint32_t)write_frame_header(j.header, p);
    return e->launch_compress(j, (uint8_t*)d_dst, dstCapacity, d_result, d_table, d_index, inband ? LZ4F_MI355X_INBAND : (d_index ? indexCapacity : 0));
}

size_t lz4f_mi355x_dev_decompressBlocksIndexed(lz4f_mi355x_engine* e, void* d_dst, size_t dstCapacity, const void* d_frame, size_t frameCapacity,
                                               const lz4f_mi355x_block* d_table, uint32_t n_blocks, const LZ4F_frameInfo_t* info,
                                               const void* d_index, size_t indexSize, lz4f_mi355x_result* d_result)
{
    if (!e || !d_table || !info) return make_err(LZ4F_ERROR_GENERIC);
    const size_t bs = block_size_of(info->blockSizeID);
    if (!bs) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
    lz4f_mi355x_engine::DecompressJob j; memset(&j, 0, sizeof(j));
    j.d_frame = (const uint8_t*)d_frame; j.frame_cap = frameCapacity; j.d_dst = (uint8_t*)d_dst; j.dst_cap = dstCapacity; j.hist0 = 0;
    j.block_size = (uint32_t)bs; j.linked = info->blockMode == LZ4F_blockLinked; j.block_checksum = info->blockChecksumFlag != 0;
    j.d_table = d_table; j.n_blocks = n_blocks; j.max_blocks = n_blocks; j.d_index = (void*)d_index; j.index_size = d_index ? indexSize : 0;
    return e->launch_decompress(j, d_result);
}

size_t lz4f_mi355x_dev_decompressFrame(lz4f_mi355x_engine* e, void* d_dst, size_t dstCapacity, const void* d_frame, size_t frameCapacity,
                                       lz4f_mi355x_result* d_result)
{
    if (!e) return make_err(LZ4F_ERROR_GENERIC);
    if (hipSetDevice(e->device) != hipSuccess) { set_last_error("hipSetDevice failed"); return make_err(LZ4F_ERROR_GENERIC); }
    // the descriptor (7..19 bytes) decides block size / linked / checksum flags: peek it (the only host sync here)
    uint8_t hdr[32]; memset(hdr, 0, sizeof(hdr));
    const size_t peek = frameCapacity < 19 ? frameCapacity : 19;
    if (peek < 7) return make_err(LZ4F_ERROR_frameHeader_incomplete);
    TrailerFoot foot; memset(&foot, 0, sizeof(foot));              // the stream's last 32 bytes: this library's trailer, if it is one
    const bool may_trail = frameCapacity >= 64 && ((uintptr_t)d_frame & 15) == 0 && !e->sw.no_trailer;
    if (hipMemcpyAsync(hdr, d_frame, peek, hipMemcpyDeviceToHost, (hipStream_t)e->stream) != hipSuccess ||
        (may_trail && hipMemcpyAsync(&foot, (const uint8_t*)d_frame + frameCapacity - sizeof(foot), sizeof(foot), hipMemcpyDeviceToHost, (hipStream_t)e->stream) != hipSuccess) ||
        hipStreamSynchronize((hipStream_t)e->stream) != hipSuccess) { set_last_error("header peek failed"); return make_err(LZ4F_ERROR_GENERIC); }
    lz4f_mi355x_engine::DecompressJob j; memset(&j, 0, sizeof(j));
    j.d_frame = (const uint8_t*)d_frame; j.frame_cap = frameCapacity; j.d_dst = (uint8_t*)d_dst; j.dst_cap = dstCapacity; j.hist0 = 0;
    j.block_size = 65536; j.linked = false; j.block_checksum = false; j.max_blocks = 1;
    const uint32_t magic = (uint32_t)hdr[0] | (hdr[1] << 8) | (hdr[2] << 16) | ((uint32_t)hdr[3] << 24);
    if (magic == 0x184D2204u) {
        ParsedHeader ph;
        size_t hs = parse_frame_header(hdr, peek, &ph);
        if (is_err(hs)) return hs;
        j.block_size = (uint32_t)ph.max_block; j.linked = ph.info.blockMode == LZ4F_blockLinked; j.block_checksum = ph.info.blockChecksumFlag != 0;
        j.content_checksum = ph.info.contentChecksumFlag != 0;
        const uint64_t by_dst = dstCapacity / ph.max_block + 2;
        const uint64_t by_src = frameCapacity / 5 + 2;                // every block costs at least 5 frame bytes
        uint64_t mb = by_dst < by_src ? by_dst : by_src;
        if (mb > 0x7FFFFFFFull) mb = 0x7FFFFFFFull;
        j.max_blocks = (uint32_t)mb;
        // a trailer (frame_dev.cuh): where it says the size words are, and the sequence index.  Only as hints: the kernels check both
        if (may_trail && foot.magic == TR_FOOT && foot.n_blocks && foot.n_blocks <= j.max_blocks && foot.total >= 8 + sizeof(foot) && foot.total <= frameCapacity - hs) {
            const uint64_t at = frameCapacity - foot.total;
            const uint64_t list_at = (at + 8 + 15) & ~(uint64_t)15, n_list = ((uint64_t)foot.n_blocks + 1) & ~1ull, ix_at = list_at + n_list * 8;
            if (ix_at + sizeof(foot) <= frameCapacity) {
                j.hint_list = (const uint64_t*)((const uint8_t*)d_frame + list_at); j.hint_n = foot.n_blocks;      // (frame_cap stays the whole buffer: `at` is a claim)
                const uint64_t ix_bytes = frameCapacity - sizeof(foot) - ix_at;
                if (ix_bytes >= sizeof(IxHeader) && foot.total_seqs) { j.d_index = (void*)((const uint8_t*)d_frame + ix_at); j.index_size = (size_t)ix_bytes; j.ix_seqs = foot.total_seqs; j.ix_entries = foot.total_entries; }
            }
        }
    }
    return e->launch_decompress(j, d_result);
}

size_t lz4f_mi355x_dev_decompressBlocks(lz4f_mi355x_engine* e, void* d_dst, size_t dstCapacity, const void* d_frame, size_t frameCapacity,
                                        const lz4f_mi355x_block* d_table, uint32_t n_blocks, const LZ4F_frameInfo_t* info,
                                        lz4f_mi355x_result* d_result)
{
    if (!e || !d_table || !info) return make_err(LZ4F_ERROR_GENERIC);
    const size_t bs = block_size_of(info->blockSizeID);
    if (!bs) return make_err(LZ4F_ERROR_maxBlockSize_invalid);
    lz4f_mi355x_engine::DecompressJob j; memset(&j, 0, sizeof(j));
    j.d_frame = (const uint8_t*)d_frame; j.frame_cap = frameCapacity; j.d_dst = (uint8_t*)d_dst; j.dst_cap = dstCapacity; j.hist0 = 0;
    j.block_size = (uint32_t)bs; j.linked = info->blockMode == LZ4F_blockLinked; j.block_checksum = info->blockChecksumFlag != 0;
    j.d_table = d_table; j.n_blocks = n_blocks; j.max_blocks = n_blocks;
    return e->launch_decompress(j, d_result);
}

size_t lz4f_mi355x_dev_xxh32(lz4f_mi355x_engine* e, const void* d_base, const uint64_t* d_off, const uint32_t* d_len, uint32_t n_blocks, uint32_t* d_out)
{
    if (!e) return make_err(LZ4F_ERROR_GENERIC);
    if (hipSetDevice(e->device) != hipSuccess) { set_last_error("hipSetDevice failed"); return make_err(LZ4F_ERROR_GENERIC); }
    if (n_blocks == 0) return 0;
    constexpr int W = 4;
    hipLaunchKernelGGL((k_xxh32_ranges<W>), dim3((n_blocks + W - 1) / W), dim3(64 * W), 0, (hipStream_t)e->stream, (const uint8_t*)d_base, d_off, d_len,
                       n_blocks, d_out);
    if (hipGetLastError() != hipSuccess) { set_last_error("xxh32 launch failed"); return make_err(LZ4F_ERROR_GENERIC); }
    return 0;
}

}  // extern "C"
