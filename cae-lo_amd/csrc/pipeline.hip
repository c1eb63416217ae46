// pipeline.hip -- frame-level executor: a few HIP streams ("lanes"), one host thread each.
//
// A frame of the hot path is ~25 short kernels, half of them latency-bound (keypoint select, voxel
// hash build, RANSAC replay) and half MFMA-bound (the 3D-CAE encoder).  Back to back on one stream
// they leave most of the 256 CUs idle most of the time, and a single host thread cannot even issue
// them as fast as the GPU retires them.  The executor therefore runs whole frames round-robin on
// `n_lanes` streams, each fed by its own host thread with its own voxel map and workspaces.  Two
// cross-lane edges, one HIP event each:
//   * pair (i-1, i) needs the rows of frame i-1;
//   * optional (CAELO_ENC_DEPTH=d, off by default): the encoder of frame i starts after the encoder of frame
//     i - d has finished.  Measured on MI355X: no gain (4.42k vs 4.43k frames/s at 3 lanes) -- the persistent
//     encoder kernels hold every CU for their whole duration, so the latency-bound kernels of the other lanes
//     stretch behind them whether or not the encoders themselves are staggered.
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

constexpr int RING = 128;  // job slots in flight (events are recycled through this ring)

struct Slot {
    hipEvent_t extracted = nullptr;  // recorded on the job's lane after its extract was enqueued
    hipEvent_t encoded = nullptr;    // the same point (extract ends with the encoder); separate object, separate consumer
    bool recorded = false;           // host-side: the record calls above have been made
    bool done = false;               // host-side: everything of the job has been enqueued
    caelo_frame_job job;
};

struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t joined = nullptr;
    caelo_voxmap *map = nullptr;
    void *ws_extract = nullptr, *ws_match = nullptr, *ws_ransac = nullptr;
    std::deque<uint64_t> queue;  // sequence numbers, guarded by caelo_pipeline::mu
    std::thread worker;
};

}  // namespace

struct caelo_pipeline {
    caelo_ctx *ctx = nullptr;
    std::vector<Lane> lanes;
    Slot slots[RING];
    std::mutex mu;
    std::condition_variable cv;  // one condvar for every state change: a handful of threads, a few events per frame
    uint64_t submitted = 0;      // next sequence number
    uint64_t retired = 0;        // every job < retired is done
    bool stop = false;
    int error = 0;
    std::string error_text;
    hipEvent_t begun = nullptr;
    int enc_depth = 0;  // > 0: encoders of at most this many frames in flight (0 or >= n_lanes: unconstrained)
    std::atomic<int64_t> stat_jobs{0}, stat_issue_ns{0}, stat_wait_ns{0};
};

namespace {

void fail(caelo_pipeline *p, int rc) {
    std::lock_guard<std::mutex> g(p->mu);
    if (!p->error) {
        p->error = rc;
        p->error_text = caelo_last_error();
    }
}

inline int64_t now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void run_job(caelo_pipeline *p, Lane &lane, uint64_t seq) {
    const int64_t t_start = now_ns();
    int64_t t_wait = 0;
    Slot &sl = p->slots[seq % RING];
    const caelo_frame_job &j = sl.job;
    const caelo_extract_args xa = {p->ctx, lane.map, j.pc, j.n, j.dist_channels, j.mode, j.rows + 60, 64, j.rows, 64,
                                   j.rows + 63, 64, j.key_pixels, j.n_key, j.flags, j.status, lane.ws_extract};
    int rc = extract_check(xa);
    if (rc == CAELO_OK) rc = extract_front_launch(xa, lane.stream);
    if (rc == CAELO_OK && p->enc_depth > 0 && (uint64_t)p->enc_depth < p->lanes.size() && seq >= (uint64_t)p->enc_depth) {
        Slot &es = p->slots[(seq - p->enc_depth) % RING];  // encoder token: behind frame seq - enc_depth
        {
            const int64_t t0 = now_ns();
            std::unique_lock<std::mutex> g(p->mu);
            p->cv.wait(g, [&] { return es.recorded; });
            t_wait += now_ns() - t0;
        }
        if (hipStreamWaitEvent(lane.stream, es.encoded, 0) != hipSuccess) {
            caelo_set_error("caelo_pipeline: hipStreamWaitEvent failed");
            rc = CAELO_ERR_HIP;
        }
    }
    if (rc == CAELO_OK) rc = extract_encode_launch(xa, lane.stream);
    if (rc == CAELO_OK && (hipEventRecord(sl.extracted, lane.stream) != hipSuccess ||
                           hipEventRecord(sl.encoded, lane.stream) != hipSuccess)) {
        caelo_set_error("caelo_pipeline: hipEventRecord failed");
        rc = CAELO_ERR_HIP;
    }
    {
        std::lock_guard<std::mutex> g(p->mu);
        sl.recorded = true;  // set even on failure: a successor must not wait forever
    }
    p->cv.notify_all();
    if (rc == CAELO_OK && j.pair != CAELO_PAIR_NONE) {
        const float *prev_rows = j.prev_rows;
        const int32_t *prev_n = j.prev_n_key;
        if (j.pair == CAELO_PAIR_CHAIN) {
            Slot &ps = p->slots[(seq - 1) % RING];
            prev_rows = ps.job.rows;
            prev_n = ps.job.n_key;
            if (p->lanes.size() > 1) {  // the predecessor ran on another lane
                {
                    const int64_t t0 = now_ns();
                    std::unique_lock<std::mutex> g(p->mu);
                    p->cv.wait(g, [&] { return ps.recorded; });
                    t_wait += now_ns() - t0;
                }
                if (hipStreamWaitEvent(lane.stream, ps.extracted, 0) != hipSuccess) {
                    caelo_set_error("caelo_pipeline: hipStreamWaitEvent failed");
                    rc = CAELO_ERR_HIP;
                }
            }
        }
        if (rc == CAELO_OK)
            rc = caelo_match(p->ctx, prev_rows, 64, CAELO_MAX_KEYPTS, prev_n, j.rows, 64, CAELO_MAX_KEYPTS, j.n_key, 60,
                             j.pair_idx, lane.ws_match, lane.stream);
        if (rc == CAELO_OK)
            rc = caelo_ransac(p->ctx, prev_rows + 60, 64, j.rows + 60, 64, j.pair_idx, CAELO_MAX_KEYPTS, j.n_key, j.rand,
                              j.result, j.inlier_mask, lane.ws_ransac, lane.stream);
    }
    if (rc != CAELO_OK) fail(p, rc);
    p->stat_jobs += 1;
    p->stat_wait_ns += t_wait;
    p->stat_issue_ns += now_ns() - t_start - t_wait;
    {
        std::lock_guard<std::mutex> g(p->mu);
        sl.done = true;
        while (p->retired < p->submitted && p->slots[p->retired % RING].done) ++p->retired;
    }
    p->cv.notify_all();
}

void worker_main(caelo_pipeline *p, int li) {
    (void)hipSetDevice(p->ctx->device);
    Lane &lane = p->lanes[li];
    for (;;) {
        uint64_t seq;
        {
            std::unique_lock<std::mutex> g(p->mu);
            p->cv.wait(g, [&] { return p->stop || !lane.queue.empty(); });
            if (lane.queue.empty()) return;  // stop requested and nothing left
            seq = lane.queue.front();
            lane.queue.pop_front();
        }
        run_job(p, lane, seq);
    }
}

int drain(caelo_pipeline *p) {
    std::unique_lock<std::mutex> g(p->mu);
    p->cv.wait(g, [&] { return p->retired == p->submitted; });
    if (p->error) {
        caelo_set_error("caelo_pipeline: %s", p->error_text.c_str());
        const int rc = p->error;
        p->error = 0;
        return rc;
    }
    return CAELO_OK;
}

}  // namespace

CAELO_API void caelo_pipeline_destroy(caelo_pipeline *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(p->mu);
        p->stop = true;
    }
    p->cv.notify_all();
    for (Lane &l : p->lanes)
        if (l.worker.joinable()) l.worker.join();
    for (Lane &l : p->lanes) {
        if (l.stream) (void)hipStreamSynchronize(l.stream);
        if (l.map) caelo_voxmap_destroy(l.map);
        if (l.ws_extract) (void)hipFree(l.ws_extract);
        if (l.ws_match) (void)hipFree(l.ws_match);
        if (l.ws_ransac) (void)hipFree(l.ws_ransac);
        if (l.joined) (void)hipEventDestroy(l.joined);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    for (Slot &s : p->slots) {
        if (s.extracted) (void)hipEventDestroy(s.extracted);
        if (s.encoded) (void)hipEventDestroy(s.encoded);
    }
    if (p->begun) (void)hipEventDestroy(p->begun);
    delete p;
}

CAELO_API int caelo_pipeline_create(caelo_ctx *c, int n_lanes, int64_t max_points, caelo_pipeline **out) {
    CAELO_REQUIRE(c && out, "null argument");
    CAELO_REQUIRE(n_lanes >= 1 && n_lanes <= 16, "n_lanes must be in [1, 16]");
    CAELO_REQUIRE(c->has_resp && c->has_enc, "weights not set");
    CAELO_HIP(hipSetDevice(c->device));
    caelo_pipeline *p = new caelo_pipeline();
    p->ctx = c;
    p->lanes.resize(n_lanes);
    int rc = CAELO_OK;
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == CAELO_OK) {
            caelo_set_error("caelo_pipeline_create: %s failed: %s", what, hipGetErrorString(e));
            rc = CAELO_ERR_HIP;
        }
    };
    hip_ok(hipEventCreateWithFlags(&p->begun, hipEventDisableTiming), "hipEventCreate");
    for (Slot &s : p->slots) {
        hip_ok(hipEventCreateWithFlags(&s.extracted, hipEventDisableTiming), "hipEventCreate");
        hip_ok(hipEventCreateWithFlags(&s.encoded, hipEventDisableTiming), "hipEventCreate");
    }
    if (const char *e = getenv("CAELO_ENC_DEPTH")) p->enc_depth = atoi(e) > 0 ? atoi(e) : 0;
    for (Lane &l : p->lanes) {
        hip_ok(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking), "hipStreamCreate");
        hip_ok(hipEventCreateWithFlags(&l.joined, hipEventDisableTiming), "hipEventCreate");
        hip_ok(hipMalloc(&l.ws_extract, (size_t)caelo_extract_ws_bytes()), "hipMalloc");
        hip_ok(hipMalloc(&l.ws_match, (size_t)caelo_match_ws_bytes(CAELO_MAX_KEYPTS)), "hipMalloc");
        hip_ok(hipMalloc(&l.ws_ransac, (size_t)caelo_ransac_ws_bytes()), "hipMalloc");
        // match / ransac workspaces are self-cleaning: zero once, every call leaves them zeroed where it matters
        if (rc == CAELO_OK) hip_ok(hipMemset(l.ws_match, 0, (size_t)caelo_match_ws_bytes(CAELO_MAX_KEYPTS)), "hipMemset");
        if (rc == CAELO_OK) hip_ok(hipMemset(l.ws_ransac, 0, (size_t)caelo_ransac_ws_bytes()), "hipMemset");
        if (rc == CAELO_OK) rc = caelo_voxmap_create(c, max_points, &l.map);
    }
    if (rc != CAELO_OK) {
        caelo_pipeline_destroy(p);
        return rc;
    }
    for (int i = 0; i < n_lanes; ++i) p->lanes[i].worker = std::thread(worker_main, p, i);
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
    return CAELO_OK;
}

CAELO_API int caelo_pipeline_begin(caelo_pipeline *p, void *stream) {
    CAELO_REQUIRE(p, "null argument");
    int rc = drain(p);
    if (rc) return rc;
    CAELO_HIP(hipEventRecord(p->begun, caelo_stream(stream)));
    for (Lane &l : p->lanes) CAELO_HIP(hipStreamWaitEvent(l.stream, p->begun, 0));
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
    std::unique_lock<std::mutex> g(p->mu);
    if (job->pair == CAELO_PAIR_CHAIN && p->submitted == 0) {
        g.unlock();
        caelo_set_error("caelo_pipeline_submit: the first job has no predecessor to chain to");
        return CAELO_ERR_ARG;
    }
    // slot seq % RING is free once job seq - RING and the jobs that consume its events (the chained successor,
    // the encoder-token successor seq - RING + enc_depth) have been enqueued
    p->cv.wait(g, [&] { return p->submitted + 1 + (uint64_t)(p->enc_depth > 1 ? p->enc_depth : 1) <= p->retired + RING; });
    const uint64_t seq = p->submitted++;
    Slot &sl = p->slots[seq % RING];
    sl.job = *job;
    sl.recorded = sl.done = false;
    p->lanes[seq % p->lanes.size()].queue.push_back(seq);
    g.unlock();
    p->cv.notify_all();
    return CAELO_OK;
}

CAELO_API int caelo_pipeline_flush(caelo_pipeline *p, void *stream) {
    CAELO_REQUIRE(p, "null argument");
    int rc = drain(p);
    for (Lane &l : p->lanes) {
        CAELO_HIP(hipEventRecord(l.joined, l.stream));
        CAELO_HIP(hipStreamWaitEvent(caelo_stream(stream), l.joined, 0));
    }
    return rc;
}
