// pipeline.hip -- frame-level executor: batches of frames behind single launches, three stages on three HIP streams (four when
// the runtime has hardware queues to spare: the voxel maps of a batch are then built beside its key-point chain).
//
// A frame of the hot path is ~25 short kernels.  The front half (ring image, response, keypoints, voxel hash, patch
// gather) and the pair half (match, RANSAC) are latency bound and leave most of the 256 CUs idle; the 3D-CAE encoder
// in between is MFMA bound.  Round 1 ran whole frames round-robin on six streams with an issue thread each: throughput
// then depended on how the runtime dealt those streams onto its hardware queues (stream creation order, lane count).
// This executor instead makes every launch wide:
//   * `batch` consecutive frames share ONE launch of every front kernel (blockIdx.z = frame, caelo_frame_set): the
//     tails, the single-workgroup stretches (keypoint selection) and the launch gaps are paid once per batch;
//   * the encoder launch set covers the batch's patches (it always could);
//   * the pairs of the batch share one match launch and one RANSAC launch per threshold level (caelo_pair_set).
// Stage k of batch b runs on its own stream: front(b+1) || encode(b) || pairs(b-1), ordered by events -- three
// streams in all, whatever the batch size, issued by the calling thread (~25 launches per batch).
// Cross-stage buffers: the bit-packed patches (front -> encoder) rotate through `n_buffers` batch buffers; descriptors,
// key points and poses land directly in the caller's per-frame buffers.
//
// Host protocol:   begin(stream) -> submit(job) ... -> flush(stream)
#include "caelo_internal.h"

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include <stdlib.h>

#include <chrono>
#include <vector>

namespace {
// Events that only order work of THIS device (stage hand-offs, the voxel stream's fork / join, the scans' arrival and release):
// no system-scope fence when they are recorded -- by default hipEventRecord writes the caches back and invalidates them so that
// the host and other devices see the data, which nobody behind these events needs (CAELO_PIPE_SYSTEM_FENCES=1 restores it).
// The events a caller's stream or the host waits on (caelo_pipeline_flush, caelo_pipeline_wait_encoded, and enc_done, which
// caelo_pipeline_sync_encoded synchronises on: results read by the host or by another GPU's collective) keep the fence.
inline unsigned local_event_flags() {
    static const bool sys = getenv("CAELO_PIPE_SYSTEM_FENCES") && atoi(getenv("CAELO_PIPE_SYSTEM_FENCES")) != 0;
    return sys ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence);
}
constexpr int MAX_BUFFERS = 4;
constexpr int64_t FRAME_PATCHES = (int64_t)CAELO_MAX_KEYPTS * 3;

inline int64_t now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

struct caelo_pipeline {
    caelo_ctx *ctx = nullptr;
    int batch = 1, n_buffers = 2;
    int64_t max_points = 0;
    hipStream_t sF = nullptr, sE = nullptr, sP = nullptr;
    hipStream_t sV = nullptr;  // voxel maps of a batch, beside the front stream's key-point chain (null: on sF)
    hipEvent_t vox_fork = nullptr, vox_join = nullptr;
    // front stage: one voxel map + workspace per frame of a batch (front stages are serial on sF)
    caelo_voxmap *maps[CAELO_FB_MAX] = {nullptr};
    void *ws_extract[CAELO_FB_MAX] = {nullptr};
    // front -> encoder: [batch] frame buffers (bit-packed patches + de-duplication tables) per hand-off buffer
    uint64_t *bits[MAX_BUFFERS] = {nullptr};
    void *enc_ws = nullptr;
    // pair stage (serial on sP)
    void *ws_match[CAELO_FB_MAX] = {nullptr}, *ws_ransac[CAELO_FB_MAX] = {nullptr};
    hipEvent_t front_done[MAX_BUFFERS] = {nullptr}, enc_done[MAX_BUFFERS] = {nullptr};
    hipEvent_t begun = nullptr, joined[3] = {nullptr};
    // caelo_pipeline_wait_stream / _release_scans / _wait_encoded: a ring of events each, so that an event is not recorded again
    // while a wait on its previous record may still sit in a queue
    static constexpr int EXT_RING = 16;
    hipEvent_t ext_in[EXT_RING] = {nullptr}, ext_out[EXT_RING] = {nullptr}, ext_enc[EXT_RING] = {nullptr};
    std::vector<hipEvent_t> up_arrived;   // caelo_pipeline_run_uploading: a batch's scans are in device memory
    unsigned n_ext_in = 0, n_ext_out = 0, n_ext_enc = 0;
    // host state
    std::vector<caelo_frame_job> pending;
    uint64_t n_batches = 0, submitted = 0;
    int since_begin = 0;  // batches issued since caelo_pipeline_begin
    int pace = 1;         // caelo_pipeline_set_pace
    std::vector<int> plan;  // batch sizes of the next run (caelo_pipeline_expect); empty or used up = full batches
    bool have_last = false;
    bool failed = false;  // a launch of a batch failed after the batch was counted: its enc_done was never recorded, so nothing may
                          // pace on or wait for "the batch before" until caelo_pipeline_begin starts a new run
    caelo_frame_job last = {};
    int64_t stat_jobs = 0, stat_issue_ns = 0, stat_batches = 0;
    // ---- the host half of the exact RANSAC inside the pipeline (jobs with result_host): a batch's certificates are copied to
    // pinned host memory once its pair stage is through -- the ISSUING thread finds that out two batches later, when it has
    // nothing to wait for (no device-side wait in any queue) -- and a certifier thread runs certify_record on them while the GPU
    // works on later batches (three threads, a record each: a batch's eight pairs are through in ~40 us, which is what the last
    // batches of a run cost after the GPU is done).
    static constexpr int CERT_RING = 6;
    struct CertItem {
        const caelo_ransac_cert *dev;
        caelo_pose_result *res;
        uint8_t *mask;
        const double *rand_host, *rand_dev;
        int32_t *info;
    };
    struct CertTask {
        CertItem item[CAELO_FB_MAX];
        int n = 0, remaining = 0;
        uint64_t batch_no = 0;
        caelo_ransac_cert *host = nullptr;   // pinned, coherent [batch]: the certifier reads the records here
        caelo_ransac_cert *host_dev = nullptr;   // the same memory as the kernels address it (zero-copy mode)
        hipEvent_t pair_done = nullptr, copied = nullptr;
        int state = 0;                       // 0 free, 1 issued (pair stage queued), 2 copy queued / being certified
    };
    CertTask cert_ring[CERT_RING];
    hipStream_t sC = nullptr;                // the certificates' copy stream
    uint64_t cert_seq = 0;                   // tasks created
    std::deque<int> cert_issued;             // (issuing thread) slots in state 1, oldest first
    std::deque<std::pair<int, int>> cert_queue;   // (slot, record) handed to the certifier threads
    std::mutex cert_mu;
    std::condition_variable cert_cv;
    static constexpr int CERT_THREADS = 8, CERT_THREADS_DEFAULT = 3;
    std::thread cert_thread[CERT_THREADS];
    bool cert_stop = false, cert_started = false;
    bool cert_zero_copy = true;   // the kernels write the records straight into pinned host memory (no copy command, no copy stream)
    int cert_failed = 0;                     // a record could not be certified (no BLAS bound, LAPACK failure): reported by the flush
    int64_t stat_cert_pairs = 0, stat_cert_evals = 0, stat_cert_ns = 0, stat_cert_drain_ns = 0;
};

namespace {

// one record of a task: wait for the task's copy, run the host half on the record; the last record of a task frees its slot
// (certifier threads; at a flush the issuing thread too)
void cert_process(caelo_pipeline *p, int slot, int i, std::vector<double> &draws) {
    caelo_pipeline::CertTask &t = p->cert_ring[slot];
    int failed = !p->cert_zero_copy && hipEventSynchronize(t.copied) != hipSuccess;
    const int64_t t0 = now_ns();
    int32_t evals = 0;
    if (!failed) {
        const caelo_pipeline::CertItem &it = t.item[i];
        int st = certify_record(t.host[i], it.rand_host, it.res, it.mask, CAELO_MAX_KEYPTS, &evals);
        if (st == 1) {   // the pair escalates beyond 0.4 m and no host copy of its draws was given: fetch them (rare)
            draws.resize((size_t)CAELO_RANSAC_LEVELS * CAELO_RANSAC_MAX_TRIALS * 4);
            if (hipMemcpy(draws.data(), it.rand_dev, draws.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) st = -1;
            else st = certify_record(t.host[i], draws.data(), it.res, it.mask, CAELO_MAX_KEYPTS, &evals);
        }
        // st 3: the kernels left no record for a job that HAS a pair (every item of a task has one) -- with cert_only they wrote no
        // result either, so nothing valid exists for this pair: a failure of the run, not a value.  (st 2, more than 1024 matches,
        // stays a status: the caller is told which pair and uses caelo_host_ransac.)
        if (st < 0 || st == 3) failed = 1;
        if (it.info) { it.info[0] = evals; it.info[1] = st; }
    }
    bool last;
    {
        std::lock_guard<std::mutex> lk(p->cert_mu);
        if (failed) p->cert_failed = 1;
        p->stat_cert_pairs += 1;
        p->stat_cert_evals += evals;
        p->stat_cert_ns += now_ns() - t0;
        last = --t.remaining == 0;
        if (last) t.state = 0;
    }
    if (last) p->cert_cv.notify_all();
}

void cert_worker(caelo_pipeline *p) {
    (void)hipSetDevice(p->ctx->device);
    std::vector<double> draws;
    for (;;) {
        std::pair<int, int> w;
        {
            std::unique_lock<std::mutex> lk(p->cert_mu);
            p->cert_cv.wait(lk, [&] { return p->cert_stop || !p->cert_queue.empty(); });
            if (p->cert_queue.empty()) return;   // (stop)
            w = p->cert_queue.front();
            p->cert_queue.pop_front();
        }
        cert_process(p, w.first, w.second, draws);
    }
}

// (issuing thread) the certificates of every issued batch except the newest `keep`: wait for the pair stage (long finished when
// keep = 2: the thread has just waited for the encoder of the batch before the newest), queue the copy, hand the task over
int cert_drain(caelo_pipeline *p, size_t keep) {
    const int64_t t_in = now_ns();
    struct Acc { caelo_pipeline *p; int64_t t; ~Acc() { p->stat_cert_drain_ns += now_ns() - t; } } acc{p, t_in};
    // A HIP failure here must not leave a task in state 1 with nothing queued: cert_task / cert_wait_idle wait for state 0 and the
    // flush, begin and destroy calls would hang instead of returning the error.  The popped task and every task still issued are
    // given back (their results are never written; the flush reports cert_failed).
    auto abandon = [&](caelo_pipeline::CertTask &cur, hipError_t e, const char *what) {
        {
            std::lock_guard<std::mutex> lk(p->cert_mu);
            cur.state = 0;
            cur.remaining = 0;
            for (const int s : p->cert_issued) {
                p->cert_ring[s].state = 0;
                p->cert_ring[s].remaining = 0;
            }
            p->cert_failed = 1;
        }
        p->cert_issued.clear();
        p->cert_cv.notify_all();
        caelo_set_error("caelo_pipeline: %s failed while handing certificates to the host half: %s", what, hipGetErrorString(e));
        return CAELO_ERR_HIP;
    };
    while (p->cert_issued.size() > keep) {
        const int slot = p->cert_issued.front();
        p->cert_issued.pop_front();
        caelo_pipeline::CertTask &t = p->cert_ring[slot];
        hipError_t e = hipEventSynchronize(t.pair_done);
        if (e != hipSuccess) return abandon(t, e, "hipEventSynchronize");
        for (int i = 0; i < t.n && !p->cert_zero_copy;) {   // records that lie back to back on the device leave in one copy
            int j = i + 1;
            while (j < t.n && t.item[j].dev == t.item[j - 1].dev + 1) ++j;
            e = hipMemcpyAsync(t.host + i, t.item[i].dev, (size_t)(j - i) * sizeof(caelo_ransac_cert), hipMemcpyDeviceToHost, p->sC);
            if (e != hipSuccess) return abandon(t, e, "hipMemcpyAsync");
            i = j;
        }
        if (!p->cert_zero_copy && (e = hipEventRecord(t.copied, p->sC)) != hipSuccess) return abandon(t, e, "hipEventRecord");
        {
            std::lock_guard<std::mutex> lk(p->cert_mu);
            t.state = 2;
            t.remaining = t.n;
            for (int i = 0; i < t.n; ++i) p->cert_queue.emplace_back(slot, i);
        }
        p->cert_cv.notify_all();
    }
    return CAELO_OK;
}

// a free task for the batch being issued (blocks while the certifier is CERT_RING batches behind)
int cert_task(caelo_pipeline *p, caelo_pipeline::CertTask **out, int *slot_out) {
    if (!p->cert_started) {
        CAELO_REQUIRE(certify_record(caelo_ransac_cert(), nullptr, nullptr, nullptr, 0, nullptr) != -1,
                      "result_host given but no BLAS is bound (caelo_host_bind_blas)");
        CAELO_HIP(hipStreamCreateWithFlags(&p->sC, hipStreamNonBlocking));
        {   // CAELO_CERT_ZEROCOPY=0: certificates in device memory (the job's `cert`), copied to the host by a copy command
            const char *e = getenv("CAELO_CERT_ZEROCOPY");
            p->cert_zero_copy = !(e && atoi(e) == 0);
        }
        for (caelo_pipeline::CertTask &t : p->cert_ring) {
            CAELO_HIP(hipHostMalloc((void **)&t.host, (size_t)p->batch * sizeof(caelo_ransac_cert), hipHostMallocCoherent | hipHostMallocMapped));
            CAELO_HIP(hipHostGetDevicePointer((void **)&t.host_dev, t.host, 0));
            CAELO_HIP(hipEventCreateWithFlags(&t.pair_done, hipEventDisableTiming));
            CAELO_HIP(hipEventCreateWithFlags(&t.copied, hipEventDisableTiming));
        }
        {   // CAELO_CERT_THREADS (1 .. 8, default 3): certifier threads of every pipeline of the process (a scheduling knob, no arithmetic)
            const char *e = getenv("CAELO_CERT_THREADS");
            int nt = e ? atoi(e) : caelo_pipeline::CERT_THREADS_DEFAULT;
            nt = nt < 1 ? 1 : (nt > caelo_pipeline::CERT_THREADS ? caelo_pipeline::CERT_THREADS : nt);
            for (int i = 0; i < nt; ++i) p->cert_thread[i] = std::thread(cert_worker, p);
        }
        p->cert_started = true;
    }
    const int slot = (int)(p->cert_seq % caelo_pipeline::CERT_RING);
    caelo_pipeline::CertTask &t = p->cert_ring[slot];
    if (t.state == 1) {   // (only with more than CERT_RING batches issued and never drained: cannot happen, the drain runs per batch)
        const int rc = cert_drain(p, 0);
        if (rc) return rc;
    }
    {
        std::unique_lock<std::mutex> lk(p->cert_mu);
        p->cert_cv.wait(lk, [&] { return t.state == 0; });
    }
    ++p->cert_seq;
    t.n = 0;
    *out = &t;
    *slot_out = slot;
    return CAELO_OK;
}

int cert_wait_idle(caelo_pipeline *p) {
    if (!p->cert_started) return CAELO_OK;
    static const bool verbose = getenv("CAELO_PIPE_VERBOSE") != nullptr;
    const int64_t tv0 = now_ns();
    const int rc = cert_drain(p, 0);
    const int64_t tv1 = now_ns();
    {   // the last batches of a run: this thread has nothing else to do and takes its share of the queue
        std::vector<double> draws;
        for (;;) {
            std::pair<int, int> w(-1, -1);
            {
                std::lock_guard<std::mutex> lk(p->cert_mu);
                if (!p->cert_queue.empty()) { w = p->cert_queue.back(); p->cert_queue.pop_back(); }
            }
            if (w.first < 0) break;
            cert_process(p, w.first, w.second, draws);
        }
    }
    const int64_t tv2 = now_ns();
    std::unique_lock<std::mutex> lk(p->cert_mu);
    p->cert_cv.wait(lk, [&] {
        for (const caelo_pipeline::CertTask &t : p->cert_ring)
            if (t.state != 0) return false;
        return true;
    });
    if (verbose) fprintf(stderr, "cert_wait_idle: drain (pair stages + copies queued) %.1f us, own share %.1f us, wait for the others %.1f us\n",
                         (tv1 - tv0) / 1e3, (tv2 - tv1) / 1e3, (now_ns() - tv2) / 1e3);
    if (rc) return rc;
    if (p->cert_failed) {
        p->cert_failed = 0;
        caelo_set_error("caelo_pipeline: a certificate could not be evaluated on the host (dgesdd failed or the copy did)");
        return CAELO_ERR_HIP;
    }
    return CAELO_OK;
}

int issue_batch_impl(caelo_pipeline *p) {
    const int n = (int)p->pending.size();
    if (n == 0) return CAELO_OK;
    // a batch never holds more frames than the pipeline has maps / workspaces for (a caller that kept submitting after a failed
    // submit used to get here with batch + 1 pending jobs)
    CAELO_REQUIRE(n <= p->batch && n <= CAELO_FB_MAX, "internal: more pending frames than the batch size");
    const int64_t t0 = now_ns();
    const uint64_t k = p->n_batches;
    const int nb = (int)(k % (uint64_t)p->n_buffers);
    const std::vector<caelo_frame_job> &jobs = p->pending;
    // every argument check comes before the first launch: a rejected batch leaves no trace (n_batches, events, buffers)
    caelo_extract_args xa[CAELO_FB_MAX];
    for (int i = 0; i < n; ++i) {
        const caelo_frame_job &j = jobs[i];
        xa[i] = {p->ctx, p->maps[i], j.pc, j.n, j.dist_channels, j.mode, j.rows + 60, 64, j.rows, 64, j.rows + 63, 64,
                 j.key_pixels, j.n_key, j.flags, j.status, p->ws_extract[i],
                 p->bits[nb] + (size_t)i * (CAELO_FRAME_BUF_BYTES / 8)};
        const int rc = extract_check(xa[i]);
        if (rc) return rc;
    }
    p->n_batches = k + 1;
    // ---- front: the hand-off buffer is free once the encoder of batch k - n_buffers has read it
    if (k >= (uint64_t)p->n_buffers) CAELO_HIP(hipStreamWaitEvent(p->sF, p->enc_done[nb], 0));
    int rc = extract_front_set(xa, n, p->sF, p->sV, p->vox_fork, p->vox_join);
    if (rc) return rc;
    CAELO_HIP(hipEventRecord(p->front_done[nb], p->sF));
    // The certificates of the batches whose pair stage has long finished go to the host half HERE: the front stage of this batch is
    // queued (what the encoder waits for next -- between the pacing wait at the end of the previous call and these launches every
    // microsecond of the issuing thread is a microsecond of the batch), the rest of the call has slack.  All but the two newest
    // issued batches: the thread waited for the encoder of the batch before the newest at the end of the previous call (a caller
    // that paces itself -- pace -1, Pipeline.run_uploading -- waits AFTER the call: one more batch of slack).
    if (!p->cert_issued.empty()) {
        if ((rc = cert_drain(p, p->pace >= 0 ? 1 : 2))) return rc;
    }
    const int64_t t1 = now_ns();
    // ---- encoder: one launch set for the batch; only the distinct patches of each frame are encoded
    CAELO_HIP(hipStreamWaitEvent(p->sE, p->front_done[nb], 0));
    {
        caelo_enc_out outs;
        outs.per_frame = FRAME_PATCHES;
        for (int i = 0; i < n; ++i) outs.base[i] = jobs[i].rows;
        static const int yield = getenv("CAELO_ENC_YIELD") ? atoi(getenv("CAELO_ENC_YIELD")) : 5;  // bit 0: stage 1 leaves a fifth of its slots; bit 1 / bit 2: conv3 half / a quarter of its (10.84 / 10.94 k frames/s; none: 10.90 k)
        const caelo_enc_in in = {(const unsigned long long *)p->bits[nb], (int64_t)(CAELO_FRAME_BUF_BYTES / 8), (int32_t)FRAME_PATCHES, n, 1, yield};
        rc = encode_batch_impl(p->ctx, p->bits[nb], n * FRAME_PATCHES, 3, outs, 64, p->enc_ws, p->sE, nullptr, &in);
        if (rc) return rc;
    }
    CAELO_HIP(hipEventRecord(p->enc_done[nb], p->sE));
    const int64_t t2 = now_ns();
    // ---- pairs: frame i against its predecessor (the previous batch's last frame for i = 0) or an explicit one
    caelo_pair_set ps = {};
    ps.faults = p->ctx->faults;
    caelo_pipeline::CertTask *ctask = nullptr;
    int cslot = -1;
    for (int i = 0; i < n; ++i) {
        const caelo_frame_job &j = jobs[i];
        if (j.pair == CAELO_PAIR_NONE) continue;
        const float *prev_rows = j.prev_rows;
        const int32_t *prev_n = j.prev_n_key;
        if (j.pair == CAELO_PAIR_CHAIN) {
            const caelo_frame_job &pj = i > 0 ? jobs[i - 1] : p->last;
            prev_rows = pj.rows;
            prev_n = pj.n_key;
        }
        caelo_pair_dev &d = ps.p[ps.n];
        d.f0 = prev_rows; d.n0 = prev_n; d.f1 = j.rows; d.n1 = j.n_key;
        d.pc0 = prev_rows + 60; d.pc1 = j.rows + 60;
        d.pair_idx = j.pair_idx; d.ws_match = p->ws_match[ps.n]; d.ws_ransac = p->ws_ransac[ps.n];
        d.rand = j.rand; d.result = j.result; d.mask = j.inlier_mask; d.cert = j.cert;
        d.cert_only = j.result_host ? 1 : 0;   // the certifier threads produce this pair's result: no k_ransac_finish for it
        ++ps.n;
        if (j.result_host) {   // the host half for this pair (certifier thread)
            if (!ctask && (rc = cert_task(p, &ctask, &cslot))) return rc;
            if (p->cert_zero_copy) {
                // the kernels write the record straight into the task's pinned host memory: no copy command, nothing for a copy to
                // flush.  (A caller that also wants the record on the device -- job.cert -- gets it only in the copy mode.)
                ps.p[ps.n - 1].cert = ctask->host_dev + ctask->n;
                ctask->host[ctask->n].magic = 0;
            } else {
                CAELO_REQUIRE(j.cert, "result_host needs cert (CAELO_CERT_ZEROCOPY=0)");
            }
            ctask->item[ctask->n++] = {j.cert, j.result_host, j.mask_host, j.rand_host, j.rand, j.info_host};
        }
    }
    if (ps.n > 0) {
        CAELO_HIP(hipStreamWaitEvent(p->sP, p->enc_done[nb], 0));  // this batch's descriptors; the predecessor's came earlier on sE
        if ((rc = match_set(ps, 64, CAELO_MAX_KEYPTS, 64, CAELO_MAX_KEYPTS, 60, p->sP))) return rc;
        if ((rc = ransac_set(ps, 64, 64, CAELO_MAX_KEYPTS, p->sP))) return rc;
        if (ctask) {
            ctask->batch_no = k;
            CAELO_HIP(hipEventRecord(ctask->pair_done, p->sP));
            ctask->state = 1;
            p->cert_issued.push_back(cslot);
        }
    }
    p->last = jobs[n - 1];
    p->have_last = true;
    p->stat_jobs += n;
    p->stat_batches += 1;
    p->since_begin += 1;
    p->pending.clear();
    const int64_t t3 = now_ns();
    p->stat_issue_ns += t3 - t0;
    // ---- pacing: the issuing thread stays at most pace + 1 batches ahead of the encoder (CAELO_PIPE_PACE, default 1; -1 = never
    // waits; caelo_pipeline_set_pace).  Waits that sit unsatisfied in the hardware queues cost this pipeline throughput -- the deeper the host runs ahead,
    // the more of them: a 20-batch run 16.5 k -> 17.3 k frames/s with the pacing, a 120-batch run 18.2 k -> 18.5 k
    if (p->pace >= 0 && p->since_begin > p->pace)
        CAELO_HIP(hipEventSynchronize(p->enc_done[(int)((k - (uint64_t)p->pace) % (uint64_t)p->n_buffers)]));

    static const bool verbose = getenv("CAELO_PIPE_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "batch %llu n=%d issue us: front %.1f enc %.1f pair %.1f\n", (unsigned long long)k, n, (t1 - t0) / 1e3, (t2 - t1) / 1e3, (t3 - t2) / 1e3);
    return CAELO_OK;
}

// Whatever happens, nothing of a batch is kept: a caller that logs an error and keeps submitting starts a fresh batch.
int issue_batch(caelo_pipeline *p) {
    if (p->failed) {
        p->pending.clear();
        caelo_set_error("caelo_pipeline: a batch of this run failed; caelo_pipeline_begin starts a new run");
        return CAELO_ERR_ARG;
    }
    const uint64_t counted = p->n_batches;
    const int rc = issue_batch_impl(p);
    if (rc) {
        p->pending.clear();
        if (p->n_batches != counted) p->failed = true;   // (a rejected argument leaves no trace; a failed launch does)
    }
    return rc;
}

}  // namespace

CAELO_API void caelo_pipeline_destroy(caelo_pipeline *p) {
    if (!p) return;
    for (hipStream_t s : {p->sF, p->sE, p->sP, p->sV})
        if (s) (void)hipStreamSynchronize(s);
    if (p->cert_started) {
        (void)cert_wait_idle(p);
        {
            std::lock_guard<std::mutex> lk(p->cert_mu);
            p->cert_stop = true;
        }
        p->cert_cv.notify_all();
        for (std::thread &th : p->cert_thread)
            if (th.joinable()) th.join();
        for (caelo_pipeline::CertTask &t : p->cert_ring) {
            if (t.host) (void)hipHostFree(t.host);
            if (t.pair_done) (void)hipEventDestroy(t.pair_done);
            if (t.copied) (void)hipEventDestroy(t.copied);
        }
        if (p->sC) (void)hipStreamDestroy(p->sC);
    }
    for (int i = 0; i < CAELO_FB_MAX; ++i) {
        if (p->maps[i]) caelo_voxmap_destroy(p->maps[i]);
        for (void *w : {p->ws_extract[i], p->ws_match[i], p->ws_ransac[i]})
            if (w) (void)hipFree(w);
    }
    for (int i = 0; i < MAX_BUFFERS; ++i) {
        if (p->bits[i]) (void)hipFree(p->bits[i]);
        if (p->front_done[i]) (void)hipEventDestroy(p->front_done[i]);
        if (p->enc_done[i]) (void)hipEventDestroy(p->enc_done[i]);
    }
    if (p->enc_ws) (void)hipFree(p->enc_ws);
    if (p->begun) (void)hipEventDestroy(p->begun);
    for (hipEvent_t e : p->joined)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->up_arrived) (void)hipEventDestroy(e);
    for (hipEvent_t e : {p->vox_fork, p->vox_join})
        if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < caelo_pipeline::EXT_RING; ++i)
        for (hipEvent_t e : {p->ext_in[i], p->ext_out[i], p->ext_enc[i]})
            if (e) (void)hipEventDestroy(e);
    if (p->sV) (void)hipStreamDestroy(p->sV);
    if (p->sP && p->sP != p->sF) (void)hipStreamDestroy(p->sP);
    if (p->sE && p->sE != p->sF) (void)hipStreamDestroy(p->sE);
    if (p->sF) (void)hipStreamDestroy(p->sF);
    delete p;
}

CAELO_API int caelo_pipeline_create(caelo_ctx *c, int batch, int n_buffers, int64_t max_points, caelo_pipeline **out) {
    CAELO_REQUIRE(c && out, "null argument");
    CAELO_REQUIRE(batch >= 1 && batch <= CAELO_FB_MAX && batch <= CAELO_ENC_MAX_FRAMES, "batch must be in [1, 8]");
    CAELO_REQUIRE(n_buffers >= 2 && n_buffers <= MAX_BUFFERS, "n_buffers must be in [2, 4]");
    CAELO_REQUIRE(c->has_resp && c->has_enc, "weights not set");
    CAELO_HIP(hipSetDevice(c->device));
    caelo_pipeline *p = new caelo_pipeline();
    p->ctx = c;
    p->batch = batch;
    p->n_buffers = n_buffers;
    {   // CAELO_PIPE_PACE: the default of caelo_pipeline_set_pace for every pipeline of the process
        const char *e = getenv("CAELO_PIPE_PACE");
        const int v = e ? atoi(e) : 1;
        p->pace = v < -1 ? -1 : (v >= n_buffers ? n_buffers - 1 : v);
    }
    p->max_points = max_points;
    p->pending.reserve(CAELO_FB_MAX);
    int rc = CAELO_OK;
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == CAELO_OK) {
            caelo_set_error("caelo_pipeline_create: %s failed: %s", what, hipGetErrorString(e));
            rc = CAELO_ERR_HIP;
        }
    };
    // CAELO_PIPE_STREAMS=1: every stage on one stream (wide launches back to back, no overlap); 2: front + pairs share
    // a stream, the encoder has its own; 3 (default): one stream per stage
    const int n_streams = getenv("CAELO_PIPE_STREAMS") ? atoi(getenv("CAELO_PIPE_STREAMS")) : 3;
    hip_ok(hipStreamCreateWithFlags(&p->sF, hipStreamNonBlocking), "hipStreamCreate");
    if (n_streams >= 2) {
        // the encoder stream is the critical path of a batch (99 % busy, its kernels stretched by the other streams'): it gets the
        // highest priority, so that conv3 / Dense(200) / the head are handed CUs first (12.3 -> 12.5 k frames/s; CAELO_PIPE_ENC_PRIO=0
        // turns it off)
        int lo = 0, hi = 0;
        const bool want = !(getenv("CAELO_PIPE_ENC_PRIO") && atoi(getenv("CAELO_PIPE_ENC_PRIO")) == 0);
        if (!want || hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
            hipStreamCreateWithPriority(&p->sE, hipStreamNonBlocking, hi) != hipSuccess) {
            (void)hipGetLastError();  // a runtime without stream priorities: an ordinary stream, nothing else changes
            p->sE = nullptr;
            hip_ok(hipStreamCreateWithFlags(&p->sE, hipStreamNonBlocking), "hipStreamCreate");
        }
    }
    else p->sE = p->sF;
    // (the pair stream at the highest or lowest priority was tried: 10.59 / 10.60 k frames/s against 10.57 k, and k_match_mfma
    // waits for CUs just as long -- the encoder's persistent workgroups do not give theirs up)
    if (n_streams >= 3) hip_ok(hipStreamCreateWithFlags(&p->sP, hipStreamNonBlocking), "hipStreamCreate");
    else p->sP = p->sF;
    // A fourth stream only pays when the runtime has more than its default four hardware queues (GPU_MAX_HW_QUEUES >= 8 in the
    // environment before HIP starts): otherwise it shares a queue with a stage it should overlap (8.98 vs 11.5 k frames/s).
    {
        const char *q = getenv("GPU_MAX_HW_QUEUES"), *v = getenv("CAELO_PIPE_VOX_STREAM");
        const bool want = v ? atoi(v) != 0 : (q && atoi(q) >= 8);
        if (n_streams >= 3 && want) {
            hip_ok(hipStreamCreateWithFlags(&p->sV, hipStreamNonBlocking), "hipStreamCreate");
            hip_ok(hipEventCreateWithFlags(&p->vox_fork, local_event_flags()), "hipEventCreate");
            hip_ok(hipEventCreateWithFlags(&p->vox_join, local_event_flags()), "hipEventCreate");
        }
    }
    hip_ok(hipEventCreateWithFlags(&p->begun, local_event_flags()), "hipEventCreate");
    for (int i = 0; i < caelo_pipeline::EXT_RING; ++i) {
        hip_ok(hipEventCreateWithFlags(&p->ext_in[i], local_event_flags()), "hipEventCreate");
        hip_ok(hipEventCreateWithFlags(&p->ext_out[i], local_event_flags()), "hipEventCreate");
        hip_ok(hipEventCreateWithFlags(&p->ext_enc[i], hipEventDisableTiming), "hipEventCreate");
    }
    for (hipEvent_t &e : p->joined) hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    for (int i = 0; i < n_buffers; ++i) {
        hip_ok(hipEventCreateWithFlags(&p->front_done[i], local_event_flags()), "hipEventCreate");
        // enc_done is what caelo_pipeline_sync_encoded blocks the HOST on before the rows are read back or handed to another
        // GPU's collective: it keeps the system-scope release (the other internal events order this device's work only)
        hip_ok(hipEventCreateWithFlags(&p->enc_done[i], hipEventDisableTiming), "hipEventCreate");
        hip_ok(hipMalloc((void **)&p->bits[i], (size_t)batch * CAELO_FRAME_BUF_BYTES), "hipMalloc");
    }
    const size_t xws = (size_t)caelo_extract_ws_bytes(), mws = (size_t)caelo_match_ws_bytes(CAELO_MAX_KEYPTS), rws = (size_t)caelo_ransac_ws_bytes();
    for (int i = 0; i < batch; ++i) {
        hip_ok(hipMalloc(&p->ws_extract[i], xws), "hipMalloc");
        hip_ok(hipMalloc(&p->ws_match[i], mws), "hipMalloc");
        hip_ok(hipMalloc(&p->ws_ransac[i], rws), "hipMalloc");
        // match / ransac workspaces are self-cleaning: zero once, every call leaves them zeroed where it matters
        if (rc == CAELO_OK) hip_ok(hipMemset(p->ws_extract[i], 0, xws), "hipMemset");
        if (rc == CAELO_OK) hip_ok(hipMemset(p->ws_match[i], 0, mws), "hipMemset");
        if (rc == CAELO_OK) hip_ok(hipMemset(p->ws_ransac[i], 0, rws), "hipMemset");
        if (rc == CAELO_OK) rc = caelo_voxmap_create(c, max_points, &p->maps[i]);
    }
    hip_ok(hipMalloc(&p->enc_ws, (size_t)caelo_encode_ws_bytes(batch * FRAME_PATCHES)), "hipMalloc");
    if (rc == CAELO_OK) hip_ok(hipMemset(p->enc_ws, 0, CAELO_ENC_WS_HEADER), "hipMemset");  // stage-1 work counters (self-cleaning)
    if (rc != CAELO_OK) {
        caelo_pipeline_destroy(p);
        return rc;
    }
    *out = p;
    return CAELO_OK;
}

CAELO_API int caelo_pipeline_batch(const caelo_pipeline *p) { return p ? p->batch : 0; }

CAELO_API int caelo_pipeline_stats(caelo_pipeline *p, int64_t *out_host) {
    CAELO_REQUIRE(p && out_host, "null argument");
    out_host[0] = p->stat_jobs;
    out_host[1] = p->stat_issue_ns;
    out_host[2] = p->stat_batches;
    out_host[3] = p->batch;
    out_host[4] = p->n_buffers;
    out_host[5] = 1 + (p->sE != p->sF) + (p->sP != p->sF) + (p->sV != nullptr);  // HIP streams in use
    p->stat_jobs = p->stat_issue_ns = p->stat_batches = 0;
    return CAELO_OK;
}

CAELO_API int caelo_pipeline_cert_stats(caelo_pipeline *p, int64_t *out_host) {
    CAELO_REQUIRE(p && out_host, "null argument");
    std::lock_guard<std::mutex> lk(p->cert_mu);
    out_host[0] = p->stat_cert_pairs;   // pairs the host half has finished since the last call
    out_host[1] = p->stat_cert_evals;   // hypotheses it evaluated for them
    out_host[2] = p->stat_cert_ns;      // certifier-thread time spent on them
    out_host[3] = p->stat_cert_drain_ns;  // issuing-thread time spent handing certificates over (event waits, copies queued)
    p->stat_cert_pairs = p->stat_cert_evals = p->stat_cert_ns = p->stat_cert_drain_ns = 0;
    return CAELO_OK;
}

CAELO_API int caelo_pipeline_begin(caelo_pipeline *p, void *stream) {
    CAELO_REQUIRE(p, "null argument");
    p->pending.clear();
    p->since_begin = 0;
    if (p->cert_started) (void)cert_wait_idle(p);
    if (p->failed) {   // the stage streams may hold half a batch: drain them, forget the chain
        for (hipStream_t s : {p->sF, p->sE, p->sP, p->sV})
            if (s) (void)hipStreamSynchronize(s);
        p->failed = false;
        p->have_last = false;
    }
    CAELO_HIP(hipEventRecord(p->begun, caelo_stream(stream)));
    for (hipStream_t s : {p->sF, p->sE, p->sP}) CAELO_HIP(hipStreamWaitEvent(s, p->begun, 0));
    return CAELO_OK;
}

// A run whose length is not a multiple of the batch size: nothing overlaps the front stage of the FIRST batch (the encoder, the
// critical resource, idles meanwhile) nor the encoder + pair stages of the LAST one, so neither should be the odd one out.  The
// hint spreads the frames evenly over ceil(n / batch) batches, the smaller ones first: 20 frames go 6 + 7 + 7 (9.8 k frames/s;
// 4 + 8 + 8: 9.5 k; 8 + 8 + 4: 9.5 k; 4 x 5: 9.0 k; 2 + 8 + 8 + 2 costs more in launch sets than it gains).  Without the hint
// every batch is full and the remainder goes last.  Results do not depend on the plan (tests/test_gpu_parity.py).
CAELO_API int caelo_pipeline_get_pace(const caelo_pipeline *p) { return p ? p->pace : 0; }

CAELO_API int caelo_upload_many(void *const *dst, const void *const *src, const size_t *bytes, int n, void *stream) {
    CAELO_REQUIRE(n >= 0 && (n == 0 || (dst && src && bytes)), "caelo_upload_many: null argument");
    for (int i = 0; i < n; ++i) {
        CAELO_REQUIRE(dst[i] && src[i], "caelo_upload_many: null buffer");
        if (bytes[i]) CAELO_HIP(hipMemcpyAsync(dst[i], src[i], bytes[i], hipMemcpyHostToDevice, caelo_stream(stream)));
    }
    return CAELO_OK;
}

// The loop of the upload modes in native code (round 6).  Pipeline.run_uploading / run_loaded paced themselves from Python: wait for a
// batch's scans, submit it, queue the copy of the batch `ahead` further on, wait for the encoder of the batch before -- and between
// that wait and the next batch's front launches sat ~60 us of interpreter (event objects, slices, ctypes), every one of them a
// microsecond of the batch (the front stream started 81 us behind the encoder's stage 1 instead of 14: 15-16 k frames/s with uploads
// against 19.9 k resident, profiles/r06_upload_native.txt).  One copy command per batch: batch b of the call (jobs [b * batch, ...)) goes
// from src[b] to dst[b], bytes[b] -- or, with a loader (caelo_seqloader), from its ring slot to dev_slots[(b0 + b) % n_slots], the
// point counts of the batch's jobs filled in from the loader.  Between caelo_pipeline_begin and the flush, both done here.
CAELO_API int caelo_pipeline_run_uploading(caelo_pipeline *p, caelo_frame_job *jobs, int64_t k, int64_t nb, caelo_seqloader *loader, int64_t b0,
                                           void *const *dst, const void *const *src, const size_t *bytes, int n_slots, const void *ring_host,
                                           int64_t slot_bytes, int ahead, void *copy_stream, void *stream, int64_t *times_ns_host) {
    CAELO_REQUIRE(p && jobs && k > 0 && nb > 0 && dst && ahead >= 1 && copy_stream, "caelo_pipeline_run_uploading: bad argument");
    CAELO_REQUIRE(loader ? (n_slots >= ahead + 2 && ring_host && slot_bytes > 0) : (src && bytes), "caelo_pipeline_run_uploading: bad copy description");
    CAELO_REQUIRE((k + p->batch - 1) / p->batch == nb, "caelo_pipeline_run_uploading: k frames do not make nb batches");
    const int B = p->batch;
    hipStream_t copy = caelo_stream(copy_stream);
    const int n_ev = ahead + 2;
    while ((int)p->up_arrived.size() < n_ev) {
        hipEvent_t e;
        CAELO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->up_arrived.push_back(e);
    }
    int64_t tw[4] = {0, 0, 0, 0};
    auto upload = [&](int64_t b) -> int {
        const int64_t lo = b * B, hi = lo + B < k ? lo + B : k;
        void *d = dst[loader ? (b0 + b) % n_slots : b];
        const void *sp;
        size_t nbytes;
        if (loader) {
            int32_t slot = 0;
            int64_t npts[CAELO_FB_MAX];
            const int64_t t0 = now_ns();
            const int rc = caelo_seqloader_wait(loader, b0 + b, &slot, npts);
            tw[0] += now_ns() - t0;
            if (rc) return rc;
            for (int64_t i = lo; i < hi; ++i) jobs[i].n = npts[i - lo];
            sp = (const char *)ring_host + (size_t)slot * (size_t)slot_bytes;
            nbytes = (size_t)slot_bytes;
        } else {
            sp = src[b];
            nbytes = bytes[b];
        }
        if (nbytes) CAELO_HIP(hipMemcpyAsync(d, sp, nbytes, hipMemcpyHostToDevice, copy));
        CAELO_HIP(hipEventRecord(p->up_arrived[(size_t)(b % n_ev)], copy));
        return CAELO_OK;
    };
    int rc = caelo_pipeline_expect(p, 0);   // full batches, the remainder last: the slots are laid out that way
    if (rc) return rc;
    const int pace = p->pace;
    p->pace = -1;                            // this loop paces itself: the copies go out BEFORE the thread waits
    rc = caelo_pipeline_begin(p, stream);
    if (rc == CAELO_OK && hipStreamWaitEvent(copy, p->begun, 0) != hipSuccess) rc = CAELO_ERR_HIP;   // (an earlier run may still read the slots)
    for (int64_t b = 0; rc == CAELO_OK && b < (ahead < nb ? ahead : nb); ++b) rc = upload(b);
    for (int64_t b = 0; rc == CAELO_OK && b < nb; ++b) {
        const int64_t lo = b * B, hi = lo + B < k ? lo + B : k;
        int64_t t0 = now_ns();
        if (hipEventSynchronize(p->up_arrived[(size_t)(b % n_ev)]) != hipSuccess) { caelo_set_error("caelo_pipeline_run_uploading: a copy failed"); rc = CAELO_ERR_HIP; break; }
        if (loader && (rc = caelo_seqloader_release(loader, b0 + b))) break;   // the copy is through: the loader may refill the slot
        int64_t t1 = now_ns();
        tw[1] += t1 - t0;
        if ((rc = caelo_pipeline_submit_many(p, jobs + lo, hi - lo))) break;
        t0 = now_ns();
        tw[2] += t0 - t1;
        if (b + ahead < nb && (rc = upload(b + ahead))) break;
        if (hi - lo == B && (rc = caelo_pipeline_sync_encoded(p, 1))) break;   // (a partial last batch is only issued by the flush)
        tw[3] += now_ns() - t0;
    }
    const int rc2 = caelo_pipeline_flush(p, stream);
    p->pace = pace;
    if (times_ns_host) for (int i = 0; i < 4; ++i) times_ns_host[i] = tw[i];   // waiting for the loader, for arrivals, submitting, copy issue + pacing
    return rc ? rc : rc2;
}

CAELO_API int caelo_pipeline_set_pace(caelo_pipeline *p, int lag) {
    CAELO_REQUIRE(p, "null argument");
    CAELO_REQUIRE(lag >= -1 && lag < p->n_buffers, "pace: -1 (the issuing thread never waits) .. buffers - 1");
    p->pace = lag;
    return CAELO_OK;
}

CAELO_API int caelo_pipeline_expect(caelo_pipeline *p, int64_t n_frames) {
    CAELO_REQUIRE(p && n_frames >= 0, "bad argument");
    p->plan.clear();
    if (n_frames > 0 && n_frames % p->batch != 0) {
        const int64_t k = (n_frames + p->batch - 1) / p->batch, base = n_frames / k, extra = n_frames % k;
        for (int64_t i = 0; i < k; ++i) p->plan.push_back((int)(base + (i >= k - extra ? 1 : 0)));
    }
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
    if (job->pair != CAELO_PAIR_NONE && job->result_host) CAELO_REQUIRE(job->mask_host, "result_host needs mask_host");
    CAELO_REQUIRE((((uintptr_t)job->cert) & 15u) == 0, "certificate not 16-byte aligned");
    if (job->pair == CAELO_PAIR_CHAIN && !p->have_last && p->pending.empty()) {
        caelo_set_error("caelo_pipeline_submit: the first job has no predecessor to chain to");
        return CAELO_ERR_ARG;
    }
    if (!p->pending.empty() && (p->pending[0].mode != job->mode)) {  // the frames of a launch share one mode
        const int rc = issue_batch(p);
        if (rc) return rc;
    }
    p->pending.push_back(*job);
    ++p->submitted;
    int target = p->since_begin < (int)p->plan.size() ? p->plan[p->since_begin] : p->batch;
    {   // CAELO_PIPE_PLAN="4,6,6,4": explicit batch sizes since begin (experiments on short runs); full batches after the list
        static const std::vector<int> env_plan = [] {
            std::vector<int> v;
            const char *e = getenv("CAELO_PIPE_PLAN");
            while (e && *e) {
                v.push_back(atoi(e));
                while (*e && *e != ',') ++e;
                if (*e == ',') ++e;
            }
            return v;
        }();
        if (p->since_begin < (int)env_plan.size() && env_plan[p->since_begin] >= 1 && env_plan[p->since_begin] <= p->batch)
            target = env_plan[p->since_begin];
    }
    if ((int)p->pending.size() >= target) return issue_batch(p);
    return CAELO_OK;
}

CAELO_API int caelo_pipeline_submit_many(caelo_pipeline *p, const caelo_frame_job *jobs, int64_t n) {
    CAELO_REQUIRE(p && (jobs || n == 0) && n >= 0, "bad argument");
    for (int64_t i = 0; i < n; ++i) {
        const int rc = caelo_pipeline_submit(p, jobs + i);
        if (rc) return rc;
    }
    return CAELO_OK;
}

// Scans that arrive while the pipeline runs (a copy stream uploading batch b + 1 during batch b, like the producer process of
// PoseEstimation.py:214-245): the front stage of every batch submitted from now on starts after what `stream` holds now ...
CAELO_API int caelo_pipeline_wait_stream(caelo_pipeline *p, void *stream) {
    CAELO_REQUIRE(p, "null argument");
    hipEvent_t e = p->ext_in[p->n_ext_in++ % caelo_pipeline::EXT_RING];
    CAELO_HIP(hipEventRecord(e, caelo_stream(stream)));
    CAELO_HIP(hipStreamWaitEvent(p->sF, e, 0));   // the voxel stream forks from sF inside every batch
    return CAELO_OK;
}

// ... and `stream` may overwrite the scan buffers of every batch ISSUED so far once their front stages (the only readers of a
// scan: projection, ring fill, voxel map) are done.
CAELO_API int caelo_pipeline_release_scans(caelo_pipeline *p, void *stream) {
    CAELO_REQUIRE(p, "null argument");
    hipEvent_t e = p->ext_out[p->n_ext_out++ % caelo_pipeline::EXT_RING];
    CAELO_HIP(hipEventRecord(e, p->sF));
    CAELO_HIP(hipStreamWaitEvent(caelo_stream(stream), e, 0));
    return CAELO_OK;
}

// ... and `stream` waits for the rows of every frame of the batches issued so far (the encoder stage writes descriptors, the front
// stage before it the key points and the validity column; the pair stage only reads them).
CAELO_API int caelo_pipeline_wait_encoded(caelo_pipeline *p, void *stream) {
    CAELO_REQUIRE(p, "null argument");
    hipEvent_t e = p->ext_enc[p->n_ext_enc++ % caelo_pipeline::EXT_RING];
    CAELO_HIP(hipEventRecord(e, p->sE));
    CAELO_HIP(hipStreamWaitEvent(caelo_stream(stream), e, 0));
    return CAELO_OK;
}

// The same hand-over paced by the HOST: returns once the rows of every batch issued so far EXCEPT THE LAST `lag` are written.
// What a caller enqueues afterwards on any stream of this device needs no device-side wait -- a wait that sits unsatisfied in a
// hardware queue for a batch's time costs this pipeline a quarter of its rate (measured: caelo_pipeline_wait_encoded once per
// batch, 17.4 k -> 12.5 k frames/s, with or without the event's system fence, with 8 or 24 hardware queues).
CAELO_API int caelo_pipeline_sync_encoded(caelo_pipeline *p, int lag) {
    CAELO_REQUIRE(p, "null argument");
    CAELO_REQUIRE(lag >= 0 && lag < p->n_buffers, "lag must be below the number of hand-off buffers");
    CAELO_REQUIRE(!p->failed, "a batch of this run failed: its rows were never written");
    if (p->since_begin <= lag) return CAELO_OK;   // nothing that old in this run
    const uint64_t k = p->n_batches - 1 - (uint64_t)lag;
    CAELO_HIP(hipEventSynchronize(p->enc_done[(int)(k % (uint64_t)p->n_buffers)]));
    return CAELO_OK;
}

CAELO_API int caelo_pipeline_flush(caelo_pipeline *p, void *stream) {
    CAELO_REQUIRE(p, "null argument");
    int rc = issue_batch(p);  // a partial last batch
    p->pending.clear();       // after a failure nothing of the batch is kept
    p->plan.clear();          // the hint of caelo_pipeline_expect holds for one run
    hipStream_t ss[3] = {p->sF, p->sE, p->sP};
    for (int i = 0; i < 3; ++i) {
        CAELO_HIP(hipEventRecord(p->joined[i], ss[i]));
        CAELO_HIP(hipStreamWaitEvent(caelo_stream(stream), p->joined[i], 0));
    }
    const int rc2 = cert_wait_idle(p);   // exact results requested: they are all written when this returns
    return rc ? rc : rc2;
}
