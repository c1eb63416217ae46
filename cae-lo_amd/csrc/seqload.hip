// seqload.hip -- host side of a sequence run: scans read from files into a pinned ring by native threads, RANSAC draws from
// NumPy's Mersenne Twister stream generated natively (host code only; no kernel in this file).
//
// Reference behaviour restated (never its code): PoseEstimation.py:214-245 prepares frame i + 1 in a generator process while the
// main loop matches frame i (np.fromfile(...).reshape(-1, 4), PoseEstimation.py:173-197 via the loaders of Match.py:46-72);
// RANSAC4RT draws its samples from NumPy's global generator, `np.random.random(4)` per iteration (Match.py:182-184).  Here the
// stream a pair consumes is RandomState(seed).random_sample(6000) -- MT19937 seeded by init_genrand, doubles by genrand_res53
// ((a >> 5) * 2^26 + (b >> 6)) / 2^53 -- so that results do not depend on sharding; caelo_host_random_sample reproduces it bit for
// bit (tests/test_abi_and_host.py compares with numpy.random.RandomState).
//
// round 5's run_sequence.py did both in Python threads (readinto + RandomState.seed / random_sample per frame under the GIL):
// 0.75-0.94 s of loader time and as much again of per-frame Python on the issuing thread for 4 541 frames, 3.8-4.6 k frames/s from
// page-cache files against 15 k for the pipeline with uploads.  caelo_seqloader: `threads` native threads pread() batch after batch
// into a caller-provided (pinned) ring of slots -- a slot = [batch][cap_points][4] f32 scans, then [batch][6000] f64 draws, so that ONE
// copy command moves a batch's scans and draws to a device slot of the same layout -- and keep a second copy of the draws in a longer
// host ring for the host half of the exact RANSAC (which reads a pair's draws, if it escalates, batches after the slot was reused); the
// issuing thread waits for a batch, uploads it and releases the slot when the copy is through.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "caelo_internal.h"

namespace {

// ---- MT19937 as numpy.random.RandomState(seed) runs it ---------------------------------------------------------------------------
struct Mt {
    uint32_t key[624];
    int pos;
    void seed(uint32_t s) {   // init_genrand (numpy: mt19937_seed)
        for (int i = 0; i < 624; ++i) {
            key[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
    }
    void gen() {
        const uint32_t N = 624, M = 397, A = 0x9908b0dfu, U = 0x80000000u, Lm = 0x7fffffffu;
        uint32_t y;
        int i = 0;
        for (; i < (int)(N - M); ++i) {
            y = (key[i] & U) | (key[i + 1] & Lm);
            key[i] = key[i + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
        }
        for (; i < (int)N - 1; ++i) {
            y = (key[i] & U) | (key[i + 1] & Lm);
            key[i] = key[i + (int)(M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
        }
        y = (key[N - 1] & U) | (key[0] & Lm);
        key[N - 1] = key[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
        pos = 0;
    }
    inline uint32_t next() {
        if (pos == 624) gen();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    inline double next_double() {   // genrand_res53 (numpy: mt19937_next_double)
        const int32_t a = (int32_t)(next() >> 5), b = (int32_t)(next() >> 6);
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
};

void random_sample(uint32_t seed, int64_t n, double *out) {
    Mt m;
    m.seed(seed);
    for (int64_t i = 0; i < n; ++i) out[i] = m.next_double();
}

}  // namespace

CAELO_API int caelo_host_random_sample(uint32_t seed, int64_t n, double *out_host) {
    CAELO_REQUIRE(out_host && n >= 0, "bad argument");
    random_sample(seed, n, out_host);
    return CAELO_OK;
}

// ---- the loader ------------------------------------------------------------------------------------------------------------------
struct caelo_seqloader {
    std::vector<std::string> paths;
    int64_t n = 0;           // frames
    int batch = 8, ring = 4;
    int64_t cap = 0;         // points per ring slot
    char *slots = nullptr;   // [ring] slots of slot_bytes: [batch][cap][4] f32 | [batch][CAELO_SEQ_DRAWS] f64
    double *keep = nullptr;  // nullable: [keep_ring][batch][CAELO_SEQ_DRAWS], batch b's draws at b % keep_ring
    int keep_ring = 0;
    int64_t slot_bytes = 0;
    int64_t seed_base = 0, first_frame = 0;
    int64_t n_batches = 0;
    // per batch: points of its frames (-1 = not loaded yet; -2 = failed), written by the workers
    std::vector<int64_t> n_points;
    std::vector<std::atomic<int>> *left = nullptr;   // frames of batch b still to load
    std::atomic<int64_t> next_item{0};               // next frame to hand to a worker
    int64_t released = 0;                            // batches released by the consumer (slots free below released + ring)
    std::mutex mu;
    std::condition_variable cv_space, cv_ready;
    std::vector<std::thread> workers;
    bool stop = false;
    int failed = 0;
    std::string error;
    int64_t stat_read_ns = 0, stat_draw_ns = 0, stat_wait_space_ns = 0;
};

namespace {

int64_t now_ns_() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

void seq_worker(caelo_seqloader *L) {
    int64_t t_read = 0, t_draw = 0, t_space = 0;
    for (;;) {
        const int64_t i = L->next_item.fetch_add(1);
        if (i >= L->n) break;
        const int64_t b = i / L->batch;
        {   // the slot of batch b is free once batch b - ring has been released
            const int64_t t0 = now_ns_();
            std::unique_lock<std::mutex> lk(L->mu);
            L->cv_space.wait(lk, [&] { return L->stop || b < L->released + L->ring; });
            t_space += now_ns_() - t0;
            if (L->stop) break;
        }
        const int slot = (int)(b % L->ring), j = (int)(i - b * L->batch);
        char *const sbase = L->slots + (size_t)slot * (size_t)L->slot_bytes;
        float *dst = (float *)sbase + (size_t)j * (size_t)L->cap * 4;
        int64_t np = -2;
        const int64_t t0 = now_ns_();
        const int fd = open(L->paths[i].c_str(), O_RDONLY);
        if (fd >= 0) {
            struct stat st;
            if (fstat(fd, &st) == 0 && st.st_size % 16 == 0 && st.st_size / 16 <= L->cap) {
                int64_t got = 0;
                while (got < st.st_size) {
                    const ssize_t r = pread(fd, (char *)dst + got, (size_t)(st.st_size - got), got);
                    if (r <= 0) break;
                    got += r;
                }
                if (got == st.st_size) np = st.st_size / 16;
            }
            close(fd);
        }
        const int64_t t1 = now_ns_();
        // the draws of pair (frame - 1, frame): RandomState(seed_base + frame - 1).random_sample(CAELO_SEQ_DRAWS)
        const int64_t seed = L->seed_base + L->first_frame + i - 1;
        double *dr = (double *)(sbase + (size_t)L->batch * (size_t)L->cap * 16) + (size_t)j * CAELO_SEQ_DRAWS;
        random_sample((uint32_t)(seed > 0 ? seed : 0), CAELO_SEQ_DRAWS, dr);
        if (L->keep) memcpy(L->keep + ((size_t)(b % L->keep_ring) * L->batch + j) * CAELO_SEQ_DRAWS, dr, sizeof(double) * CAELO_SEQ_DRAWS);
        t_read += t1 - t0;
        t_draw += now_ns_() - t1;
        bool done;
        {
            std::lock_guard<std::mutex> lk(L->mu);
            L->n_points[i] = np;
            if (np < 0) {
                L->failed = 1;
                if (L->error.empty()) L->error = "cannot read " + L->paths[i] + " (missing, not a multiple of 16 bytes, or more points than the ring slot holds)";
            }
            done = --(*L->left)[b] == 0;
        }
        if (done || np < 0) L->cv_ready.notify_all();
    }
    std::lock_guard<std::mutex> lk(L->mu);
    L->stat_read_ns += t_read;
    L->stat_draw_ns += t_draw;
    L->stat_wait_space_ns += t_space;
}

}  // namespace

CAELO_API int64_t caelo_seqloader_slot_bytes(int batch, int64_t cap_points) {
    return (int64_t)batch * cap_points * 16 + (int64_t)batch * CAELO_SEQ_DRAWS * 8;
}

CAELO_API int caelo_seqloader_create(const char *const *paths, int64_t n, int64_t first_frame, int batch, int ring_batches, int64_t cap_points,
                                     void *ring_host, double *draws_keep_host, int keep_batches, int64_t seed_base, int threads, caelo_seqloader **out) {
    CAELO_REQUIRE(paths && out && ring_host && n > 0 && batch >= 1 && batch <= CAELO_FB_MAX && ring_batches >= 2 && cap_points > 0 &&
                  threads >= 1 && threads <= 256 && (!draws_keep_host || keep_batches >= ring_batches), "caelo_seqloader_create: bad argument");
    caelo_seqloader *L = new caelo_seqloader();
    L->paths.assign(paths, paths + n);
    L->n = n; L->batch = batch; L->ring = ring_batches; L->cap = cap_points;
    L->slots = (char *)ring_host; L->keep = draws_keep_host; L->keep_ring = keep_batches; L->slot_bytes = caelo_seqloader_slot_bytes(batch, cap_points);
    L->seed_base = seed_base; L->first_frame = first_frame;
    L->n_batches = (n + batch - 1) / batch;
    L->n_points.assign((size_t)n, -1);
    L->left = new std::vector<std::atomic<int>>((size_t)L->n_batches);
    for (int64_t b = 0; b < L->n_batches; ++b) {
        const int64_t lo = b * batch, hi = lo + batch < n ? lo + batch : n;
        (*L->left)[b].store((int)(hi - lo));
    }
    for (int t = 0; t < threads; ++t) L->workers.emplace_back(seq_worker, L);
    *out = L;
    return CAELO_OK;
}

// blocks until batch b is in its ring slot: -> the slot, the point counts of its frames (n_points_host [batch]; frames past the end: 0)
CAELO_API int caelo_seqloader_wait(caelo_seqloader *L, int64_t b, int32_t *slot_host, int64_t *n_points_host) {
    CAELO_REQUIRE(L && slot_host && n_points_host && b >= 0 && b < L->n_batches, "caelo_seqloader_wait: bad argument");
    std::unique_lock<std::mutex> lk(L->mu);
    CAELO_REQUIRE(b >= L->released && b < L->released + L->ring, "caelo_seqloader_wait: the batch is outside the ring (release the batches in order)");
    L->cv_ready.wait(lk, [&] { return L->failed || (*L->left)[b].load() == 0; });
    if (L->failed) {
        caelo_set_error("caelo_seqloader: %s", L->error.c_str());
        return CAELO_ERR_ARG;
    }
    *slot_host = (int32_t)(b % L->ring);
    for (int j = 0; j < L->batch; ++j) {
        const int64_t i = b * L->batch + j;
        n_points_host[j] = i < L->n ? L->n_points[i] : 0;
    }
    return CAELO_OK;
}

// the consumer is through with batch b's slot (its upload has completed); batches are released in order
CAELO_API int caelo_seqloader_release(caelo_seqloader *L, int64_t b) {
    CAELO_REQUIRE(L, "null argument");
    {
        std::lock_guard<std::mutex> lk(L->mu);
        CAELO_REQUIRE(b == L->released, "caelo_seqloader_release: batches are released in order");
        L->released = b + 1;
    }
    L->cv_space.notify_all();
    return CAELO_OK;
}

// out_host[3]: nanoseconds the workers spent reading files, generating draws, waiting for a free slot (summed over the threads; valid
// after the last batch has been waited for)
CAELO_API int caelo_seqloader_stats(caelo_seqloader *L, int64_t *out_host) {
    CAELO_REQUIRE(L && out_host, "null argument");
    std::lock_guard<std::mutex> lk(L->mu);
    out_host[0] = L->stat_read_ns; out_host[1] = L->stat_draw_ns; out_host[2] = L->stat_wait_space_ns;
    return CAELO_OK;
}

CAELO_API void caelo_seqloader_destroy(caelo_seqloader *L) {
    if (!L) return;
    {
        std::lock_guard<std::mutex> lk(L->mu);
        L->stop = true;
        L->next_item.store(L->n);
    }
    L->cv_space.notify_all();
    for (std::thread &t : L->workers)
        if (t.joinable()) t.join();
    delete L->left;
    delete L;
}
