// pipeline.hip -- frame-level executor: HIP streams ("lanes") with one host thread each, encoders batched.
//
// A frame of the hot path is ~25 short kernels.  The front half (ring image, response, keypoints, voxel hash,
// patch gather) and the pair half (match, RANSAC) are latency bound and leave most of the 256 CUs idle; the
// 3D-CAE encoder in between is MFMA bound but pays ~47 us of fixed cost per launch set (weights into
// registers, pipeline fill, tails) on ~120 us of work.  One host thread cannot even issue the kernels as
// fast as the GPU retires them.  The executor therefore
//   * runs the fronts of consecutive frames concurrently, one lane (stream + issue thread + voxel map +
//     workspace) each, writing their bit-packed patches into a batch buffer;
//   * encodes `batch` frames with ONE launch set on a separate stream (batch x 3072 patches: 167 -> 146 -> 135 us
//     per frame at 1 -> 2 -> 3 frames), the descriptors landing in each frame's own rows;
//   * runs the pairs back on the lanes, `n_lanes` frames behind the fronts, so that a lane does not idle waiting
//     for the encoder: lane j issues  F(i)  P(i - n_lanes)  F(i + n_lanes)  P(i) ...
// Cross-stream edges are HIP events: front(i) -> encoder(batch of i) -> pair(i); frame i's pair also needs the
// rows of frame i-1, which the same event covers (its batch is the same or an earlier one on the encoder stream).
//
// Host protocol (the submitting thread):   begin(stream) -> submit(job) ... -> flush(stream)
#include "caelo_internal.h"

#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int RING = 256;         // job slots in flight
constexpr int BATCH_RING = RING;  // batch records (a batch holds >= 1 job, so never more batches than jobs)
constexpr int MAX_BITS_BUFFERS = 12;
constexpr int64_t FRAME_PATCHES = (int64_t)CAELO_MAX_KEYPTS * 3;

struct Slot {
    caelo_frame_job job;
    hipEvent_t fronted = nullptr;  // recorded on the job's lane after its front half was enqueued
    bool front_rec = false;        // host side: that record call has been made
    uint64_t batch = 0;            // absolute batch number
    int index = 0;                 // position inside the batch
    bool retired = false;          // host side: the job's last task (its pair) has been enqueued on the GPU
};

struct Batch {
    hipEvent_t t0 = nullptr, t1 = nullptr;  // CAELO_PIPE_TIMING: encoder launch set begin / end (timed events)
    hipEvent_t encoded = nullptr;  // recorded on the encoder stream after the batch was enqueued
    bool enc_rec = false;
    uint64_t first = 0;            // sequence number of its first job
    int count = 0;
};

enum TaskKind { FRONT, PAIR, ENCODE };
struct Task {
    TaskKind kind;
    uint64_t id;  // job sequence number, or absolute batch number
};

struct Worker {
    hipStream_t stream = nullptr;
    hipEvent_t joined = nullptr;
    std::deque<Task> queue;  // guarded by caelo_pipeline::mu
    std::thread thread;
    // lanes only
    caelo_voxmap *map = nullptr;
    void *ws_extract = nullptr, *ws_match = nullptr, *ws_ransac = nullptr;
};

}  // namespace

struct caelo_pipeline {
    caelo_ctx *ctx = nullptr;
    std::vector<Worker> lanes;
    std::vector<Worker> encoders;  // batch b is encoded on encoders[b % size]: consecutive batches may overlap
    std::vector<void *> enc_ws;
    int batch = 1;       // frames per encoder launch set
    int n_bits = 2;      // batch buffers of bit-packed patches: front of batch b + n_bits waits for encoder b
    uint64_t *bits[MAX_BITS_BUFFERS] = {nullptr};
    Slot slots[RING];
    Batch batches[BATCH_RING];
    std::mutex mu;
    std::condition_variable cv;  // one condvar for every state change: a handful of threads, a few events per frame
    uint64_t submitted = 0;      // next job sequence number
    uint64_t epoch_base = 0;     // first job of the current begin..flush epoch (lanes and batches restart there)
    uint64_t next_batch = 0;     // absolute number of the batch being filled
    uint64_t tasks_queued = 0, tasks_done = 0;
    uint64_t retired = 0;        // every job < retired has had all its tasks enqueued (slot / event reuse)
    bool stop = false;
    int error = 0;
    std::string error_text;
    hipEvent_t begun = nullptr;
    std::atomic<int64_t> stat_jobs{0}, stat_issue_ns{0}, stat_wait_ns{0};
    bool timing = false;             // CAELO_PIPE_TIMING=1: time the encoder launch sets with HIP events
    std::vector<uint64_t> timed;     // batches enqueued since the last flush
    double enc_busy_ms = 0, enc_span_ms = 0;
};

namespace {

inline int64_t now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void fail(caelo_pipeline *p, int rc) {
    std::lock_guard<std::mutex> g(p->mu);
    if (!p->error) {
        p->error = rc;
        p->error_text = caelo_last_error();
    }
}

// After the first failure nothing more is launched (later tasks would read buffers the failed one never wrote);
// the bookkeeping still runs so that nobody waits forever, and flush() reports the error.
bool failed(caelo_pipeline *p) {
    std::lock_guard<std::mutex> g(p->mu);
    return p->error != 0;
}

int hip_rc(hipError_t e, const char *what) {
    if (e == hipSuccess) return CAELO_OK;
    caelo_set_error("caelo_pipeline: %s failed: %s", what, hipGetErrorString(e));
    return CAELO_ERR_HIP;
}

// block (host) until another worker has made `pred` true; returns the ns spent waiting
template <class Pred>
int64_t wait_for(caelo_pipeline *p, Pred pred) {
    const int64_t t0 = now_ns();
    std::unique_lock<std::mutex> g(p->mu);
    p->cv.wait(g, pred);
    return now_ns() - t0;
}

int run_front(caelo_pipeline *p, Worker &lane, uint64_t seq, int64_t *waited) {
    Slot &sl = p->slots[seq % RING];
    const caelo_frame_job &j = sl.job;
    int rc = CAELO_OK;
    const bool skip = failed(p);
    if (!skip && sl.batch >= (uint64_t)p->n_bits) {  // the batch buffer is free once encoder(batch - n_bits) has run
        Batch &old = p->batches[(sl.batch - p->n_bits) % BATCH_RING];
        *waited += wait_for(p, [&] { return old.enc_rec; });
        rc = hip_rc(hipStreamWaitEvent(lane.stream, old.encoded, 0), "hipStreamWaitEvent");
    }
    const caelo_extract_args xa = {p->ctx, lane.map, j.pc, j.n, j.dist_channels, j.mode, j.rows + 60, 64, j.rows, 64,
                                   j.rows + 63, 64, j.key_pixels, j.n_key, j.flags, j.status, lane.ws_extract,
                                   p->bits[sl.batch % p->n_bits] + (size_t)sl.index * (CAELO_FRAME_BUF_BYTES / 8)};
    if (!skip && rc == CAELO_OK) rc = extract_check(xa);
    if (!skip && rc == CAELO_OK) rc = extract_front_launch(xa, lane.stream);
    if (rc == CAELO_OK) rc = hip_rc(hipEventRecord(sl.fronted, lane.stream), "hipEventRecord");
    {
        std::lock_guard<std::mutex> g(p->mu);
        sl.front_rec = true;  // set even on failure: the encoder must not wait forever
    }
    p->cv.notify_all();
    return rc;
}

int run_encode(caelo_pipeline *p, uint64_t b, int64_t *waited) {
    Batch &bt = p->batches[b % BATCH_RING];
    Worker &enc = p->encoders[b % p->encoders.size()];
    int rc = CAELO_OK;
    caelo_enc_out outs;
    outs.per_frame = FRAME_PATCHES;
    for (int i = 0; i < bt.count && rc == CAELO_OK; ++i) {
        Slot &sl = p->slots[(bt.first + i) % RING];
        *waited += wait_for(p, [&] { return sl.front_rec; });
        rc = hip_rc(hipStreamWaitEvent(enc.stream, sl.fronted, 0), "hipStreamWaitEvent");
        outs.base[i] = sl.job.rows;
    }
    if (p->timing && rc == CAELO_OK) (void)hipEventRecord(bt.t0, enc.stream);
    if (rc == CAELO_OK && !failed(p))
    {
        // frame buffers of the batch: [3072][64] patches + de-duplication tables each, only distinct patches are encoded
        const caelo_enc_in in = {(const unsigned long long *)p->bits[b % p->n_bits], (int64_t)(CAELO_FRAME_BUF_BYTES / 8),
                                 (int32_t)FRAME_PATCHES, bt.count, 1, 1};
        rc = encode_batch_impl(p->ctx, p->bits[b % p->n_bits], bt.count * FRAME_PATCHES, 3, outs, 64,
                               p->enc_ws[b % p->encoders.size()], enc.stream, nullptr, &in);
    }
    if (p->timing && rc == CAELO_OK) (void)hipEventRecord(bt.t1, enc.stream);
    if (rc == CAELO_OK) rc = hip_rc(hipEventRecord(bt.encoded, enc.stream), "hipEventRecord");
    {
        std::lock_guard<std::mutex> g(p->mu);
        if (p->timing) p->timed.push_back(b);
        bt.enc_rec = true;
    }
    p->cv.notify_all();
    return rc;
}

int run_pair(caelo_pipeline *p, Worker &lane, uint64_t seq, int64_t *waited) {
    Slot &sl = p->slots[seq % RING];
    const caelo_frame_job &j = sl.job;
    if (j.pair == CAELO_PAIR_NONE || failed(p)) return CAELO_OK;
    const float *prev_rows = j.prev_rows;
    const int32_t *prev_n = j.prev_n_key;
    if (j.pair == CAELO_PAIR_CHAIN) {
        const Slot &ps = p->slots[(seq - 1) % RING];
        prev_rows = ps.job.rows;
        prev_n = ps.job.n_key;
    }
    // descriptors of this frame and of its chained predecessor (same batch, or the one before on another encoder stream)
    Batch &bt = p->batches[sl.batch % BATCH_RING];
    *waited += wait_for(p, [&] { return bt.enc_rec; });
    int rc = hip_rc(hipStreamWaitEvent(lane.stream, bt.encoded, 0), "hipStreamWaitEvent");
    if (rc == CAELO_OK && j.pair == CAELO_PAIR_CHAIN && p->slots[(seq - 1) % RING].batch != sl.batch) {
        Batch &pb = p->batches[p->slots[(seq - 1) % RING].batch % BATCH_RING];
        *waited += wait_for(p, [&] { return pb.enc_rec; });
        rc = hip_rc(hipStreamWaitEvent(lane.stream, pb.encoded, 0), "hipStreamWaitEvent");
    }
    if (rc == CAELO_OK)
        rc = caelo_match(p->ctx, prev_rows, 64, CAELO_MAX_KEYPTS, prev_n, j.rows, 64, CAELO_MAX_KEYPTS, j.n_key, 60, j.pair_idx,
                         lane.ws_match, lane.stream);
    if (rc == CAELO_OK)
        rc = caelo_ransac(p->ctx, prev_rows + 60, 64, j.rows + 60, 64, j.pair_idx, CAELO_MAX_KEYPTS, j.n_key, j.rand, j.result,
                          j.inlier_mask, lane.ws_ransac, lane.stream);
    return rc;
}

void worker_main(caelo_pipeline *p, Worker *w) {
    (void)hipSetDevice(p->ctx->device);
    for (;;) {
        Task t;
        {
            std::unique_lock<std::mutex> g(p->mu);
            p->cv.wait(g, [&] { return p->stop || !w->queue.empty(); });
            if (w->queue.empty()) return;  // stop requested and nothing left
            t = w->queue.front();
            w->queue.pop_front();
        }
        const int64_t t0 = now_ns();
        int64_t waited = 0;
        int rc;
        if (t.kind == FRONT) rc = run_front(p, *w, t.id, &waited);
        else if (t.kind == PAIR) rc = run_pair(p, *w, t.id, &waited);
        else rc = run_encode(p, t.id, &waited);
        if (rc != CAELO_OK) fail(p, rc);
        p->stat_issue_ns += now_ns() - t0 - waited;
        p->stat_wait_ns += waited;
        if (t.kind == PAIR) p->stat_jobs += 1;
        {
            std::lock_guard<std::mutex> g(p->mu);
            ++p->tasks_done;
            if (t.kind == PAIR) {
                p->slots[t.id % RING].retired = true;
                while (p->retired < p->submitted && p->slots[p->retired % RING].retired) ++p->retired;
            }
        }
        p->cv.notify_all();
    }
}

void enqueue(caelo_pipeline *p, Worker &w, TaskKind kind, uint64_t id) {  // p->mu held
    w.queue.push_back(Task{kind, id});
    ++p->tasks_queued;
}

// p->mu held: hand the batch being filled (if it holds a job) to the encoder and open the next one
void close_batch(caelo_pipeline *p) {
    Batch &bt = p->batches[p->next_batch % BATCH_RING];
    if (bt.count == 0) return;
    enqueue(p, p->encoders[p->next_batch % p->encoders.size()], ENCODE, p->next_batch);
    ++p->next_batch;
    p->batches[p->next_batch % BATCH_RING].count = 0;
}

int drain(caelo_pipeline *p) {
    std::unique_lock<std::mutex> g(p->mu);
    p->cv.wait(g, [&] { return p->tasks_done == p->tasks_queued; });
    if (p->error) {
        caelo_set_error("caelo_pipeline: %s", p->error_text.c_str());
        const int rc = p->error;
        p->error = 0;
        return rc;
    }
    return CAELO_OK;
}

void destroy_worker(Worker &w) {
    if (w.stream) (void)hipStreamSynchronize(w.stream);
    if (w.map) caelo_voxmap_destroy(w.map);
    if (w.ws_extract) (void)hipFree(w.ws_extract);
    if (w.ws_match) (void)hipFree(w.ws_match);
    if (w.ws_ransac) (void)hipFree(w.ws_ransac);
    if (w.joined) (void)hipEventDestroy(w.joined);
    if (w.stream) (void)hipStreamDestroy(w.stream);
}

}  // namespace

CAELO_API void caelo_pipeline_destroy(caelo_pipeline *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(p->mu);
        p->stop = true;
    }
    p->cv.notify_all();
    for (Worker &l : p->lanes)
        if (l.thread.joinable()) l.thread.join();
    for (Worker &e : p->encoders)
        if (e.thread.joinable()) e.thread.join();
    for (Worker &l : p->lanes) destroy_worker(l);
    for (Worker &e : p->encoders) destroy_worker(e);
    for (Slot &s : p->slots)
        if (s.fronted) (void)hipEventDestroy(s.fronted);
    for (Batch &b : p->batches) {
        if (b.encoded) (void)hipEventDestroy(b.encoded);
        if (b.t0) (void)hipEventDestroy(b.t0);
        if (b.t1) (void)hipEventDestroy(b.t1);
    }
    for (uint64_t *b : p->bits)
        if (b) (void)hipFree(b);
    for (void *w : p->enc_ws)
        if (w) (void)hipFree(w);
    if (p->begun) (void)hipEventDestroy(p->begun);
    delete p;
}

CAELO_API int caelo_pipeline_create(caelo_ctx *c, int n_lanes, int batch, int64_t max_points, caelo_pipeline **out) {
    CAELO_REQUIRE(c && out, "null argument");
    CAELO_REQUIRE(n_lanes >= 1 && n_lanes <= 16, "n_lanes must be in [1, 16]");
    CAELO_REQUIRE(batch >= 1 && batch <= CAELO_ENC_MAX_FRAMES && batch <= n_lanes, "batch must be in [1, min(8, n_lanes)]");
    CAELO_REQUIRE(c->has_resp && c->has_enc, "weights not set");
    CAELO_HIP(hipSetDevice(c->device));
    caelo_pipeline *p = new caelo_pipeline();
    p->ctx = c;
    p->lanes.resize(n_lanes);
    p->batch = batch;
    // the fronts run up to n_lanes frames ahead of the encoder: enough batch buffers that they never wait for it
    p->n_bits = (n_lanes + batch - 1) / batch + 2;
    if (p->n_bits > MAX_BITS_BUFFERS) p->n_bits = MAX_BITS_BUFFERS;
    int rc = CAELO_OK;
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == CAELO_OK) {
            caelo_set_error("caelo_pipeline_create: %s failed: %s", what, hipGetErrorString(e));
            rc = CAELO_ERR_HIP;
        }
    };
    hip_ok(hipEventCreateWithFlags(&p->begun, hipEventDisableTiming), "hipEventCreate");
    for (Slot &s : p->slots) hip_ok(hipEventCreateWithFlags(&s.fronted, hipEventDisableTiming), "hipEventCreate");
    for (Batch &b : p->batches) hip_ok(hipEventCreateWithFlags(&b.encoded, hipEventDisableTiming), "hipEventCreate");
    p->timing = getenv("CAELO_PIPE_TIMING") && atoi(getenv("CAELO_PIPE_TIMING")) > 0;
    if (p->timing)
        for (Batch &b : p->batches) {
            hip_ok(hipEventCreate(&b.t0), "hipEventCreate");
            hip_ok(hipEventCreate(&b.t1), "hipEventCreate");
        }
    for (Worker &l : p->lanes) {
        hip_ok(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking), "hipStreamCreate");
        hip_ok(hipEventCreateWithFlags(&l.joined, hipEventDisableTiming), "hipEventCreate");
        hip_ok(hipMalloc(&l.ws_extract, (size_t)caelo_extract_ws_bytes()), "hipMalloc");
        if (rc == CAELO_OK) hip_ok(hipMemset(l.ws_extract, 0, (size_t)caelo_extract_ws_bytes()), "hipMemset");
        hip_ok(hipMalloc(&l.ws_match, (size_t)caelo_match_ws_bytes(CAELO_MAX_KEYPTS)), "hipMalloc");
        hip_ok(hipMalloc(&l.ws_ransac, (size_t)caelo_ransac_ws_bytes()), "hipMalloc");
        // match / ransac workspaces are self-cleaning: zero once, every call leaves them zeroed where it matters
        if (rc == CAELO_OK) hip_ok(hipMemset(l.ws_match, 0, (size_t)caelo_match_ws_bytes(CAELO_MAX_KEYPTS)), "hipMemset");
        if (rc == CAELO_OK) hip_ok(hipMemset(l.ws_ransac, 0, (size_t)caelo_ransac_ws_bytes()), "hipMemset");
        if (rc == CAELO_OK) rc = caelo_voxmap_create(c, max_points, &l.map);
    }
    // one encoder stream measured best (5.3 k frames/s at 6 lanes x 3 frames; 2 streams 4.7 k, 3 streams 4.9 k: the
    // persistent encoder kernels of two batches only get in each other's way); CAELO_ENC_STREAMS overrides
    const int n_enc = getenv("CAELO_ENC_STREAMS") ? atoi(getenv("CAELO_ENC_STREAMS")) : 1;
    p->encoders.resize(n_enc >= 1 && n_enc <= 4 ? n_enc : 1);
    p->enc_ws.assign(p->encoders.size(), nullptr);
    for (size_t i = 0; i < p->encoders.size(); ++i) {
        hip_ok(hipStreamCreateWithFlags(&p->encoders[i].stream, hipStreamNonBlocking), "hipStreamCreate");
        hip_ok(hipEventCreateWithFlags(&p->encoders[i].joined, hipEventDisableTiming), "hipEventCreate");
        hip_ok(hipMalloc(&p->enc_ws[i], (size_t)caelo_encode_ws_bytes(batch * FRAME_PATCHES)), "hipMalloc");
        if (rc == CAELO_OK) hip_ok(hipMemset(p->enc_ws[i], 0, 256), "hipMemset");  // stage-1 work counter (self-cleaning)
    }
    for (int i = 0; i < p->n_bits; ++i)
        hip_ok(hipMalloc((void **)&p->bits[i], (size_t)batch * CAELO_FRAME_BUF_BYTES), "hipMalloc");
    if (rc != CAELO_OK) {
        caelo_pipeline_destroy(p);
        return rc;
    }
    for (Worker &l : p->lanes) l.thread = std::thread(worker_main, p, &l);
    for (Worker &e : p->encoders) e.thread = std::thread(worker_main, p, &e);
    *out = p;
    return CAELO_OK;
}

CAELO_API int caelo_pipeline_lanes(const caelo_pipeline *p) { return p ? (int)p->lanes.size() : 0; }

CAELO_API int caelo_pipeline_stats(caelo_pipeline *p, int64_t *out_host) {
    CAELO_REQUIRE(p && out_host, "null argument");
    out_host[0] = p->stat_jobs.exchange(0);
    out_host[1] = p->stat_issue_ns.exchange(0);
    out_host[2] = p->stat_wait_ns.exchange(0);
    out_host[3] = (int64_t)p->lanes.size();
    out_host[4] = (int64_t)(p->enc_busy_ms * 1e6);  // CAELO_PIPE_TIMING: ns the encoder stream was inside a launch set ...
    out_host[5] = (int64_t)(p->enc_span_ms * 1e6);  // ... out of this many ns between the first begin and the last end (last flush)
    return CAELO_OK;
}

CAELO_API int caelo_pipeline_begin(caelo_pipeline *p, void *stream) {
    CAELO_REQUIRE(p, "null argument");
    int rc = drain(p);
    if (rc) return rc;
    CAELO_HIP(hipEventRecord(p->begun, caelo_stream(stream)));
    for (Worker &l : p->lanes) CAELO_HIP(hipStreamWaitEvent(l.stream, p->begun, 0));
    for (Worker &e : p->encoders) CAELO_HIP(hipStreamWaitEvent(e.stream, p->begun, 0));
    return CAELO_OK;
}

CAELO_API int caelo_pipeline_submit(caelo_pipeline *p, const caelo_frame_job *job) {
    CAELO_REQUIRE(p && job, "null argument");
    CAELO_REQUIRE(job->pc && job->rows && job->key_pixels && job->n_key && job->flags && job->status, "null frame buffer");
    CAELO_REQUIRE(job->n > 3, "PC.shape[0] > 3 (SphericalRing.py:73)");
    CAELO_REQUIRE(job->pair >= CAELO_PAIR_NONE && job->pair <= CAELO_PAIR_EXPLICIT, "bad pair mode");
    if (job->pair != CAELO_PAIR_NONE)
        CAELO_REQUIRE(job->rand && job->result && job->inlier_mask && job->pair_idx, "null pair buffer");
    if (job->pair == CAELO_PAIR_EXPLICIT) CAELO_REQUIRE(job->prev_rows, "explicit pair without prev_rows");
    const uint64_t L = p->lanes.size();
    std::unique_lock<std::mutex> g(p->mu);
    if (job->pair == CAELO_PAIR_CHAIN && p->submitted == 0) {
        g.unlock();
        caelo_set_error("caelo_pipeline_submit: the first job has no predecessor to chain to");
        return CAELO_ERR_ARG;
    }
    // Slot seq % RING (and the batch record / events that go with it) is reused: job seq - RING, its chained successor
    // and the batches that consume its batch's `encoded` event (its own pairs, the fronts n_bits batches later) must
    // have been enqueued.  Pairs lag the submissions by n_lanes jobs only, so with RING >> the margin below this never
    // waits on work that only a later submit would release.
    const uint64_t margin = (uint64_t)(p->n_bits + 1) * p->batch + L + 2;
    p->cv.wait(g, [&] { return p->submitted + margin <= p->retired + RING; });
    const uint64_t seq = p->submitted++;
    const uint64_t e = seq - p->epoch_base;  // index inside the epoch
    Slot &sl = p->slots[seq % RING];
    Batch &bt = p->batches[p->next_batch % BATCH_RING];
    if (bt.count == 0) {
        bt.first = seq;
        bt.enc_rec = false;
    }
    sl.job = *job;
    sl.front_rec = false;
    sl.retired = false;
    sl.batch = p->next_batch;
    sl.index = bt.count++;
    Worker &lane = p->lanes[e % L];
    enqueue(p, lane, FRONT, seq);
    if (e >= L) enqueue(p, lane, PAIR, seq - L);  // the pair of the frame this lane handled n_lanes frames ago
    if (bt.count == p->batch) close_batch(p);
    g.unlock();
    p->cv.notify_all();
    return CAELO_OK;
}

CAELO_API int caelo_pipeline_flush(caelo_pipeline *p, void *stream) {
    CAELO_REQUIRE(p, "null argument");
    const uint64_t L = p->lanes.size();
    {
        std::lock_guard<std::mutex> g(p->mu);
        close_batch(p);  // a partial last batch
        const uint64_t n = p->submitted - p->epoch_base;
        for (uint64_t e = n > L ? n - L : 0; e < n; ++e) enqueue(p, p->lanes[e % L], PAIR, p->epoch_base + e);
        p->epoch_base = p->submitted;
    }
    p->cv.notify_all();
    int rc = drain(p);
    if (p->timing && !p->timed.empty()) {   // diagnostic mode: synchronises
        for (Worker &e : p->encoders) (void)hipStreamSynchronize(e.stream);
        float ms = 0, busy = 0;
        for (uint64_t b : p->timed)
            if (hipEventElapsedTime(&ms, p->batches[b % BATCH_RING].t0, p->batches[b % BATCH_RING].t1) == hipSuccess) busy += ms;
        float span = 0;
        (void)hipEventElapsedTime(&span, p->batches[p->timed.front() % BATCH_RING].t0, p->batches[p->timed.back() % BATCH_RING].t1);
        p->enc_busy_ms = busy;
        p->enc_span_ms = span;
        p->timed.clear();
    }
    for (Worker &l : p->lanes) {
        CAELO_HIP(hipEventRecord(l.joined, l.stream));
        CAELO_HIP(hipStreamWaitEvent(caelo_stream(stream), l.joined, 0));
    }
    for (Worker &e : p->encoders) {
        CAELO_HIP(hipEventRecord(e.joined, e.stream));
        CAELO_HIP(hipStreamWaitEvent(caelo_stream(stream), e.joined, 0));
    }
    return rc;
}
