// Direct peer-to-peer reassembly: the all-gather of olx_field_allgather without RCCL (OLX_GATHER=p2p).
//
// xGMI is point to point -- 7 links per GPU, one to every peer of the node.  A ring all-gather moves every shard over ONE link
// per step (7 steps); here every rank PULLS the shard of each of its 7 peers at the same time, one copy stream per peer, so all
// 7 inbound links carry data at once (~7x the ring for the 537 MB shards of BASELINE configs[2]).  The
// shards are read straight out of the peers' output buffers, which the ranks map into each other's address space with HIP IPC
// (hipIpcGetMemHandle / hipIpcOpenMemHandle: works across processes on one node, also for two processes that share ONE
// device -- which is how tests/test_gpu_p2p.py runs two real ranks on the builder's single GPU).
//
// Cross-process ordering goes through a small POSIX shared-memory control block of generation counters, never through GPU
// spinning and never through blocking stream callbacks (HIP runs those on the runtime's one signal-handler thread):
//     ready[r][g & 1]       2 g + b: generation g of rank r's |p| block is complete, in ITS output buffer b
//     pulled[i][r][g & 1]   last generation (of that parity) of rank r's block that rank i has finished copying
// Every rank runs one worker thread.  A gather of generation g out of the own buffer b:  wait for the own field kernel (event) ->
// ready[me][g & 1] = 2 g + b -> per peer r: wait ready[r][g & 1] >= 2 g, read the peer's buffer index out of it, enqueue the copy on
// stream r -> wait for the copies -> pulled[me][r][g & 1] = g.  The slots go by GENERATION parity -- the ranks issue the same gathers
// in the same order -- not by buffer index: a rank may launch without gathering (a clock ramp, a discarded step), so the same
// generation can sit in buffer 0 on one rank and in buffer 1 on another.  olx_field_launch, before it overwrites buffer b again,
// waits until every peer has pulled the generation that last sat in it (the outputs are double-buffered, so the copies of step s
// overlap the kernel of step s + 1); between generations g and g + 2 both buffers have been rewritten, so slot g & 1 is free again
// by the time g + 2 is published.  Any wait gives up after OLX_P2P_TIMEOUT_S (default 60) seconds, raises the block's abort flag so
// that the other ranks give up too, and the next C-ABI call reports it.
//
// The aggregate over foci (olx_field_allreduce_aggregate / olx_field_reduce_scatter_aggregate) takes the same road: every rank's partial
// aggregate (max |p|, summed intensity share) is IPC-mapped by its peers, rank r OWNS slice r of the volume, pulls slice r of every
// peer's partial over the 7 links at once ((N - 1) / N of one volume pair inbound, like a reduce-scatter), reduces in rank order (every
// rank therefore ends up with the same bits), and -- all-reduce only -- the ranks then pull each other's reduced slices.  Counters:
//     agg_ready[r]   partial of generation g complete in rank r's aggregate buffers
//     agg_reduced[r] rank r's own slice holds the global values of generation g
//     agg_done[r]    rank r has finished every pull of generation g (its peers may overwrite their aggregate buffers again)
#include "olx_ctx.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

namespace {

constexpr int P2P_MAX_RANKS = 16;
constexpr char P2P_MAGIC[8] = {'O', 'L', 'X', 'P', '2', 'P', '2', 0};

struct P2PControl {                                  // lives in POSIX shared memory, zero-filled at creation
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> abort;
    std::atomic<uint64_t> ready[P2P_MAX_RANKS][2];
    std::atomic<uint64_t> pulled[P2P_MAX_RANKS][P2P_MAX_RANKS][2];
    std::atomic<uint64_t> agg_ready[P2P_MAX_RANKS], agg_reduced[P2P_MAX_RANKS], agg_done[P2P_MAX_RANKS];
};

struct P2PBlob {                                     // what a rank publishes after every plan (OLX_P2P_BLOB_BYTES)
    char magic[8];
    hipIpcMemHandle_t mem[2];
    uint64_t count;                                  // floats per block (planned foci x slab voxels)
    int32_t device, pid;
    hipIpcMemHandle_t agg[2];                        // aggregate buffers: max |p|, mean intensity (the second only with agg_i != 0)
    uint64_t agg_vox;                                // voxels of one aggregate volume
    int32_t agg_i, reserved;
};
static_assert(sizeof(P2PBlob) <= OLX_P2P_BLOB_BYTES, "blob must fit the ABI constant");

struct Job { int kind /* 0 gather, 1 aggregate */; int b; uint64_t gen; size_t count; bool scatter, with_i; };

}  // namespace

struct P2PState {
    std::string shm_name;
    P2PControl* ctl = nullptr;
    bool owner = false;
    int nranks = 1, rank = 0;
    double timeout_s = 60.0;
    const float* peer[P2P_MAX_RANKS][2] = {};        // IPC mappings of the peers' output buffers
    bool imported = false;
    uint64_t peer_count = 0;
    hipStream_t copy_stream[P2P_MAX_RANKS] = {};
    const float* peer_agg[P2P_MAX_RANKS][2] = {};    // IPC mappings of the peers' aggregate buffers
    uint64_t agg_vox = 0;
    bool agg_with_i = false;
    float* agg_scratch = nullptr;                    // [rank][2][slice]: the peers' partials of this rank's slice
    size_t agg_scratch_cap = 0;
    uint64_t agg_gen = 0, agg_done_gen = 0;          // aggregate generations issued / finished by this rank's worker
    uint64_t gen = 0;                                // generations issued by this rank
    uint64_t buf_gen[2] = {0, 0};                    // generation that last used output buffer b as its source
    // worker
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> jobs;
    uint64_t done_gen = 0;
    bool stop = false;
    std::string error;                               // set by the worker, reported by the next API call
};

static bool p2p_wait(P2PState* s, const std::atomic<uint64_t>& v, uint64_t want) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (v.load(std::memory_order_acquire) < want) {
        if (s->ctl->abort.load(std::memory_order_relaxed)) return false;
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(20)); else std::this_thread::yield();
        if ((spins & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > s->timeout_s) {
            s->ctl->abort.store(1);
            return false;
        }
    }
    return true;
}

// slice owner: global values of its slice out of the N partials, in rank order (own partial at its place) -- same bits on every rank
__global__ __launch_bounds__(256) void p2p_reduce_slice_k(float* __restrict__ own_p, float* __restrict__ own_i, const float* __restrict__ scratch,
                                                           int nranks, int me, size_t n, size_t stride) {
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (size_t)gridDim.x * blockDim.x) {
        float m = 0.f, sum = 0.f;
        for (int r = 0; r < nranks; ++r) {
            m = fmaxf(m, r == me ? own_p[v] : scratch[((size_t)r * 2) * stride + v]);
            if (own_i) sum += r == me ? own_i[v] : scratch[((size_t)r * 2 + 1) * stride + v];
        }
        own_p[v] = m;
        if (own_i) own_i[v] = sum;
    }
}

static inline void p2p_slice(size_t vox, int nranks, int r, size_t& lo, size_t& hi) {
    const size_t chunk = ((vox + nranks - 1) / nranks + 3) & ~(size_t)3;        // whole float4s
    lo = std::min(vox, chunk * r); hi = std::min(vox, lo + chunk);
}

// one aggregate exchange on the worker thread (see the header)
static std::string p2p_aggregate_job(olx_ctx* c, P2PState* s, const Job& j) {
    const size_t vox = j.count;
    size_t lo, hi;
    p2p_slice(vox, s->nranks, s->rank, lo, hi);
    const size_t n = hi - lo, stride = s->agg_scratch_cap / (2 * (size_t)s->nranks);
    float* own[2] = {c->d_agg_p, j.with_i ? c->d_agg_i : nullptr};
    if (hipEventSynchronize(c->ev_agg) != hipSuccess) return "p2p: waiting for the local aggregate failed";
    s->ctl->agg_ready[s->rank].store(j.gen, std::memory_order_release);
    for (int k = 1; k < s->nranks; ++k) {                    // slice `rank` of every peer's partial, all links at once
        const int r = (s->rank + k) % s->nranks;
        if (!p2p_wait(s, s->ctl->agg_ready[r], j.gen)) return "p2p: timed out waiting for a peer's partial aggregate (or a peer aborted)";
        for (int q = 0; q < 2 && n; ++q)
            if (own[q] && hipMemcpyAsync(s->agg_scratch + ((size_t)r * 2 + q) * stride, s->peer_agg[r][q] + lo, sizeof(float) * n, hipMemcpyDeviceToDevice,
                                         s->copy_stream[r]) != hipSuccess) return "p2p: pull of a partial aggregate failed";
    }
    for (int r = 0; r < s->nranks; ++r)
        if (r != s->rank && hipStreamSynchronize(s->copy_stream[r]) != hipSuccess) return "p2p: copy stream failed";
    if (n) {
        hipLaunchKernelGGL(p2p_reduce_slice_k, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, s->copy_stream[s->rank], own[0] + lo,
                           own[1] ? own[1] + lo : nullptr, s->agg_scratch, s->nranks, s->rank, n, stride);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s->copy_stream[s->rank]) != hipSuccess) return "p2p: slice reduction failed";
    }
    s->ctl->agg_reduced[s->rank].store(j.gen, std::memory_order_release);
    if (!j.scatter) {                                        // all-reduce: every rank's reduced slice into place
        for (int k = 1; k < s->nranks; ++k) {
            const int r = (s->rank + k) % s->nranks;
            if (!p2p_wait(s, s->ctl->agg_reduced[r], j.gen)) return "p2p: timed out waiting for a peer's reduced slice (or a peer aborted)";
            size_t rlo, rhi;
            p2p_slice(vox, s->nranks, r, rlo, rhi);
            for (int q = 0; q < 2 && rhi > rlo; ++q)
                if (own[q] && hipMemcpyAsync(own[q] + rlo, s->peer_agg[r][q] + rlo, sizeof(float) * (rhi - rlo), hipMemcpyDeviceToDevice, s->copy_stream[r]) != hipSuccess)
                    return "p2p: pull of a reduced slice failed";
        }
        for (int r = 0; r < s->nranks; ++r)
            if (r != s->rank && hipStreamSynchronize(s->copy_stream[r]) != hipSuccess) return "p2p: copy stream failed";
    }
    s->ctl->agg_done[s->rank].store(j.gen, std::memory_order_release);
    return "";
}

static void p2p_worker(olx_ctx* c) {
    P2PState* s = c->p2p;
    hipSetDevice(c->device);
    for (;;) {
        Job j;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [&] { return s->stop || !s->jobs.empty(); });
            if (s->stop && s->jobs.empty()) return;
            j = s->jobs.front();
        }
        std::string err;
        if (j.kind == 1) {
            err = p2p_aggregate_job(c, s, j);
            if (!err.empty()) { (void)hipGetLastError(); s->ctl->abort.store(1); }
            {
                std::lock_guard<std::mutex> lk(s->mu);
                if (!err.empty() && s->error.empty()) s->error = err;
                s->agg_done_gen = j.gen;
                s->jobs.pop_front();
            }
            s->cv.notify_all();
            continue;
        }
        const size_t bytes = sizeof(float) * j.count;
        if (hipEventSynchronize(c->ev_field[j.b]) != hipSuccess) err = "p2p: waiting for the field kernel failed";
        const int slot = (int)(j.gen & 1);
        if (err.empty()) s->ctl->ready[s->rank][slot].store(2 * j.gen + (uint64_t)j.b, std::memory_order_release);
        // own block first (local copy), then every peer as soon as it reports its block complete
        if (err.empty() && hipMemcpyAsync(c->d_gather + j.count * s->rank, c->d_pmag[j.b], bytes, hipMemcpyDeviceToDevice, s->copy_stream[s->rank]) != hipSuccess)
            err = "p2p: local copy failed";
        for (int k = 1; err.empty() && k < s->nranks; ++k) {
            const int r = (s->rank + k) % s->nranks;         // staggered start: rank i begins with peer i + 1
            if (!p2p_wait(s, s->ctl->ready[r][slot], 2 * j.gen)) { err = "p2p: timed out waiting for a peer's block (or a peer aborted)"; break; }
            const uint64_t pub = s->ctl->ready[r][slot].load(std::memory_order_acquire);
            if ((pub >> 1) != j.gen) { err = "p2p: a peer is a generation ahead (the ranks did not issue the same gathers)"; break; }
            if (hipMemcpyAsync(c->d_gather + j.count * r, s->peer[r][pub & 1], bytes, hipMemcpyDeviceToDevice, s->copy_stream[r]) != hipSuccess)
                err = "p2p: peer copy failed";
        }
        for (int r = 0; r < s->nranks; ++r) {
            if (hipStreamSynchronize(s->copy_stream[r]) != hipSuccess && err.empty()) err = "p2p: copy stream failed";
            if (err.empty()) s->ctl->pulled[s->rank][r][slot].store(j.gen, std::memory_order_release);
        }
        if (!err.empty()) { (void)hipGetLastError(); s->ctl->abort.store(1); }
        {
            std::lock_guard<std::mutex> lk(s->mu);
            if (!err.empty() && s->error.empty()) s->error = err;
            s->done_gen = j.gen;
            s->jobs.pop_front();
        }
        s->cv.notify_all();
    }
}

static int p2p_map_control(olx_ctx* c, P2PState* s, bool create) {
    const int fd = shm_open(s->shm_name.c_str(), create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) return fail(c, OLX_ECOMM, "p2p: shm_open(%s) failed", s->shm_name.c_str());
    if (create && ftruncate(fd, sizeof(P2PControl)) != 0) { close(fd); shm_unlink(s->shm_name.c_str()); return fail(c, OLX_ECOMM, "p2p: ftruncate failed"); }
    void* m = mmap(nullptr, sizeof(P2PControl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fail(c, OLX_ECOMM, "p2p: mmap of the control block failed");
    s->ctl = static_cast<P2PControl*>(m);      // (a fresh segment is zero-filled: every counter starts at 0)
    s->owner = create;
    return OLX_OK;
}

bool olx_p2p_requested() {
    const char* g = getenv("OLX_GATHER");
    return g && !strcmp(g, "p2p");
}

bool olx_p2p_is_id(const void* id_bytes) { return memcmp(id_bytes, P2P_MAGIC, sizeof P2P_MAGIC) == 0; }

// rank 0: create the control block; the 128-byte id carries its name
int olx_p2p_unique_id(olx_ctx* c, void* id_bytes) {
    if (c->p2p) return fail(c, OLX_ESTATE, "p2p: transport already initialised");
    P2PState* s = new P2PState();
    char name[96];
    snprintf(name, sizeof name, "/olx_p2p_%d_%llx", (int)getpid(), (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    s->shm_name = name;
    int rc = p2p_map_control(c, s, true);
    if (rc) { delete s; return rc; }
    c->p2p = s;
    memset(id_bytes, 0, OLX_UNIQUE_ID_BYTES);
    memcpy(id_bytes, P2P_MAGIC, sizeof P2P_MAGIC);
    strncpy(static_cast<char*>(id_bytes) + 8, name, OLX_UNIQUE_ID_BYTES - 9);
    return OLX_OK;
}

int olx_p2p_init(olx_ctx* c, const void* id_bytes, int nranks, int rank) {
    if (nranks > P2P_MAX_RANKS) return fail(c, OLX_EINVAL, "p2p: at most %d ranks", P2P_MAX_RANKS);
    P2PState* s = c->p2p;
    const char* name = static_cast<const char*>(id_bytes) + 8;
    if (s && s->shm_name != name) return fail(c, OLX_ESTATE, "p2p: id does not belong to this context's control block");
    if (!s) {
        s = new P2PState();
        s->shm_name = name;
        int rc = p2p_map_control(c, s, false);
        if (rc) { delete s; return rc; }
        c->p2p = s;
    }
    s->nranks = nranks; s->rank = rank;
    if (const char* t = getenv("OLX_P2P_TIMEOUT_S")) { const double v = atof(t); if (v > 0) s->timeout_s = v; }
    for (int r = 0; r < nranks; ++r) HIPCHK(c, hipStreamCreateWithFlags(&s->copy_stream[r], hipStreamNonBlocking));
    // the last rank to attach removes the NAME: the mappings stay valid, and a rank 0 that dies later cannot leak /dev/shm/olx_p2p_*
    if (s->ctl->attached.fetch_add(1) + 1 == (uint32_t)nranks) shm_unlink(s->shm_name.c_str());
    s->worker = std::thread(p2p_worker, c);
    return OLX_OK;
}

int olx_p2p_attached(olx_ctx* c) { return (c->p2p && c->p2p->ctl) ? (int)c->p2p->ctl->attached.load() : 0; }

static void p2p_close_peers(P2PState* s) {
    for (int r = 0; r < P2P_MAX_RANKS; ++r)
        for (int b = 0; b < 2; ++b) {
            if (s->peer[r][b] && r != s->rank) hipIpcCloseMemHandle(const_cast<float*>(s->peer[r][b]));
            s->peer[r][b] = nullptr;
            if (s->peer_agg[r][b] && r != s->rank) hipIpcCloseMemHandle(const_cast<float*>(s->peer_agg[r][b]));
            s->peer_agg[r][b] = nullptr;
        }
    s->imported = false; s->agg_vox = 0;
}

static int p2p_drain_locked(olx_ctx* c, P2PState* s, std::unique_lock<std::mutex>& lk) {
    s->cv.wait(lk, [&] { return s->jobs.empty(); });
    if (!s->error.empty()) { const std::string e = s->error; s->error.clear(); return fail(c, OLX_ECOMM, "%s", e.c_str()); }
    return OLX_OK;
}

// every gather issued so far has landed in this rank's gather buffer
int olx_p2p_drain(olx_ctx* c) {
    P2PState* s = c->p2p;
    if (!s) return OLX_OK;
    std::unique_lock<std::mutex> lk(s->mu);
    return p2p_drain_locked(c, s, lk);
}

int olx_p2p_destroy(olx_ctx* c) {
    P2PState* s = c->p2p;
    if (!s) return OLX_OK;
    if (s->worker.joinable()) {
        { std::lock_guard<std::mutex> lk(s->mu); s->stop = true; }
        s->cv.notify_all();
        s->worker.join();
    }
    p2p_close_peers(s);
    for (int r = 0; r < P2P_MAX_RANKS; ++r) if (s->copy_stream[r]) hipStreamDestroy(s->copy_stream[r]);
    if (s->agg_scratch) hipFree(s->agg_scratch);
    if (s->ctl) munmap(s->ctl, sizeof(P2PControl));
    if (s->owner) shm_unlink(s->shm_name.c_str());      // (already gone when every rank attached: ENOENT is fine)
    delete s;
    c->p2p = nullptr;
    return OLX_OK;
}

// after every olx_field_plan: publish the IPC handles of this rank's (double-buffered) output blocks
int olx_p2p_export(olx_ctx* c, void* blob_out) {
    P2PState* s = c->p2p;
    if (!s) return fail(c, OLX_ESTATE, "olx_comm_export: the communicator does not use the p2p transport");
    if (!c->planned || !c->d_pmag[0] || !c->d_pmag[1]) return fail(c, OLX_ESTATE, "olx_comm_export: plan first (with the communicator initialised)");
    { int rc = olx_p2p_drain(c); if (rc) return rc; }
    P2PBlob b{};
    memcpy(b.magic, P2P_MAGIC, sizeof P2P_MAGIC);
    for (int k = 0; k < 2; ++k) HIPCHK(c, hipIpcGetMemHandle(&b.mem[k], c->d_pmag[k]));
    b.count = (uint64_t)c->fp.vox * (uint64_t)c->plan_foci;
    b.device = c->device; b.pid = (int)getpid();
    // the aggregate buffers travel with the blocks (allocated here if no aggregate has been formed yet)
    const bool with_i = (c->flags & OLX_OUT_INTENSITY) != 0;
    if (!c->d_agg_p) HIPCHK(c, hipMalloc((void**)&c->d_agg_p, sizeof(float) * c->out_cap));
    if (with_i && !c->d_agg_i) HIPCHK(c, hipMalloc((void**)&c->d_agg_i, sizeof(float) * c->out_cap));
    HIPCHK(c, hipIpcGetMemHandle(&b.agg[0], c->d_agg_p));
    if (with_i) HIPCHK(c, hipIpcGetMemHandle(&b.agg[1], c->d_agg_i));
    b.agg_vox = (uint64_t)c->fp.vox; b.agg_i = with_i ? 1 : 0;
    memset(blob_out, 0, OLX_P2P_BLOB_BYTES);
    memcpy(blob_out, &b, sizeof b);
    return OLX_OK;
}

int olx_p2p_import(olx_ctx* c, const void* blobs) {
    P2PState* s = c->p2p;
    if (!s) return fail(c, OLX_ESTATE, "olx_comm_import: the communicator does not use the p2p transport");
    { int rc = olx_p2p_drain(c); if (rc) return rc; }
    HIPCHK(c, hipSetDevice(c->device));
    p2p_close_peers(s);
    const uint64_t mine = (uint64_t)c->fp.vox * (uint64_t)c->plan_foci;
    for (int r = 0; r < s->nranks; ++r) {
        P2PBlob b;
        memcpy(&b, static_cast<const unsigned char*>(blobs) + (size_t)r * OLX_P2P_BLOB_BYTES, sizeof b);
        if (memcmp(b.magic, P2P_MAGIC, sizeof P2P_MAGIC) != 0) return fail(c, OLX_EINVAL, "olx_comm_import: blob %d is not a p2p export", r);
        if (b.count != mine) return fail(c, OLX_EINVAL, "olx_comm_import: rank %d planned %llu floats per block, this rank %llu (equal blocks needed)", r,
                                         (unsigned long long)b.count, (unsigned long long)mine);
        for (int k = 0; k < 2; ++k) {
            if (r == s->rank) { s->peer[r][k] = c->d_pmag[k]; continue; }
            void* p = nullptr;
            HIPCHK(c, hipIpcOpenMemHandle(&p, b.mem[k], hipIpcMemLazyEnablePeerAccess));
            s->peer[r][k] = static_cast<const float*>(p);
        }
        const bool with_i = (c->flags & OLX_OUT_INTENSITY) != 0;
        if (b.agg_vox != (uint64_t)c->fp.vox || (b.agg_i != 0) != with_i)
            return fail(c, OLX_EINVAL, "olx_comm_import: rank %d aggregates %llu voxels%s, this rank %lld%s (equal volumes needed)", r, (unsigned long long)b.agg_vox,
                        b.agg_i ? " + intensity" : "", (long long)c->fp.vox, with_i ? " + intensity" : "");
        for (int k = 0; k < (with_i ? 2 : 1); ++k) {
            if (r == s->rank) { s->peer_agg[r][k] = k ? c->d_agg_i : c->d_agg_p; continue; }
            void* p = nullptr;
            HIPCHK(c, hipIpcOpenMemHandle(&p, b.agg[k], hipIpcMemLazyEnablePeerAccess));
            s->peer_agg[r][k] = static_cast<const float*>(p);
        }
    }
    s->peer_count = mine;
    s->agg_vox = (uint64_t)c->fp.vox; s->agg_with_i = (c->flags & OLX_OUT_INTENSITY) != 0;
    s->imported = true;
    return OLX_OK;
}

// olx_field_allgather, p2p transport: hand generation g = (this rank's gather count) of buffer c->cur to the worker
int olx_p2p_allgather(olx_ctx* c) {
    P2PState* s = c->p2p;
    const size_t count = (size_t)c->fp.vox * c->plan_foci;
    if (!s->imported || s->peer_count != count)
        return fail(c, OLX_ESTATE, "olx_field_allgather (p2p): exchange olx_comm_export / olx_comm_import after the plan first");
    const size_t need = count * s->nranks;
    if (c->gather_cap < need) {
        { int rc = olx_p2p_drain(c); if (rc) return rc; }
        if (c->d_gather) hipFree(c->d_gather);
        c->d_gather = nullptr; c->gather_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_gather, sizeof(float) * need));
        c->gather_cap = need;
    }
    const int b = c->cur;
    HIPCHK(c, hipEventRecord(c->ev_field[b], c->stream));
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (!s->error.empty()) { const std::string e = s->error; s->error.clear(); return fail(c, OLX_ECOMM, "%s", e.c_str()); }
        // (one job per buffer in flight at most: the caller's next launch into this buffer waits in olx_p2p_before_overwrite)
        s->buf_gen[b] = ++s->gen;
        s->jobs.push_back(Job{0, b, s->gen, count, false, false});
    }
    s->cv.notify_all();
    c->gather_pending[b] = true;
    return OLX_OK;
}

// olx_field_launch is about to overwrite output buffer b: its last generation must have been pulled by every rank (incl. this one)
int olx_p2p_before_overwrite(olx_ctx* c, int b) {
    P2PState* s = c->p2p;
    const uint64_t g = s->buf_gen[b];
    if (g == 0) return OLX_OK;
    {   // the own worker must have issued (and finished) generation g first
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->done_gen >= g || !s->error.empty(); });
        if (!s->error.empty()) { const std::string e = s->error; s->error.clear(); return fail(c, OLX_ECOMM, "%s", e.c_str()); }
    }
    for (int r = 0; r < s->nranks; ++r)
        if (!p2p_wait(s, s->ctl->pulled[r][s->rank][g & 1], g)) return fail(c, OLX_ECOMM, "p2p: rank %d did not pull generation %llu of this rank's block in time", r, (unsigned long long)g);
    return OLX_OK;
}

// ---- aggregate exchange (olx_field_allreduce_aggregate / olx_field_reduce_scatter_aggregate, p2p transport) ----

// the aggregate buffers are about to be overwritten: this rank's last exchange must be over on EVERY rank that pulls from them
int olx_p2p_aggregate_before_overwrite(olx_ctx* c) {
    P2PState* s = c->p2p;
    if (!s || s->agg_gen == 0) return OLX_OK;
    const uint64_t g = s->agg_gen;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->agg_done_gen >= g || !s->error.empty(); });
        if (!s->error.empty()) { const std::string e = s->error; s->error.clear(); return fail(c, OLX_ECOMM, "%s", e.c_str()); }
    }
    for (int r = 0; r < s->nranks; ++r)
        if (!p2p_wait(s, s->ctl->agg_done[r], g)) return fail(c, OLX_ECOMM, "p2p: rank %d did not finish aggregate exchange %llu in time", r, (unsigned long long)g);
    return OLX_OK;
}

// the local partial has been enqueued on c->stream and c->ev_agg recorded behind it: hand the exchange to the worker
int olx_p2p_aggregate(olx_ctx* c, bool scatter, bool with_i) {
    P2PState* s = c->p2p;
    if (!s->imported || s->agg_vox != (uint64_t)c->fp.vox || s->agg_with_i != with_i || s->peer_agg[s->rank][0] != c->d_agg_p)
        return fail(c, OLX_ESTATE, "aggregate exchange (p2p): exchange olx_comm_export / olx_comm_import after the plan first");
    size_t lo, hi;
    p2p_slice((size_t)c->fp.vox, s->nranks, 0, lo, hi);
    const size_t need = 2 * (size_t)s->nranks * (hi - lo);
    if (s->agg_scratch_cap < need) {
        { int rc = olx_p2p_drain(c); if (rc) return rc; }
        if (s->agg_scratch) hipFree(s->agg_scratch);
        s->agg_scratch = nullptr; s->agg_scratch_cap = 0;
        HIPCHK(c, hipMalloc((void**)&s->agg_scratch, sizeof(float) * need));
        s->agg_scratch_cap = need;
    }
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (!s->error.empty()) { const std::string e = s->error; s->error.clear(); return fail(c, OLX_ECOMM, "%s", e.c_str()); }
        s->jobs.push_back(Job{1, 0, ++s->agg_gen, (size_t)c->fp.vox, scatter, with_i});
    }
    s->cv.notify_all();
    return OLX_OK;
}
