// In-process multi-GPU host path: a pool of workers, one host thread and one extractor context per listed device,
// that cuts a stream of host frames into contiguous, pair-aligned shards and runs each shard through its worker's
// pipelined host entry point (orbx_extract_batch: pinned upload and result rings, copy streams beside the compute
// streams).  Frames are independent, so nothing crosses devices; every worker writes its frames' results straight
// into the caller's arrays, which are therefore in frame order when the call returns.  Declared in include/orbx.h;
// built on the public C ABI only.
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/orbx.h"

extern "C" int orbx_host_frame_shard(int n_frames, int rank, int world, int* first, int* last) {
    if (n_frames < 0 || world < 1 || rank < 0 || rank >= world || !first || !last) return ORBX_E_INVALID;
    // pairs (2k, 2k + 1) stay together: the shards are runs of pairs whose sizes differ by at most one pair
    const int n_pairs = (n_frames + 1) / 2;
    const int base = n_pairs / world, rem = n_pairs % world;
    const int a = rank * base + std::min(rank, rem);
    const int b = a + base + (rank < rem ? 1 : 0);
    *first = std::min(2 * a, n_frames);
    *last = std::min(2 * b, n_frames);
    return ORBX_OK;
}

namespace {

struct Job {
    const uint8_t* const* imgs = nullptr;
    const int* strides = nullptr;
    int width = 0, height = 0, kp_cap = 0;
    orbx_kp* kps = nullptr;
    uint8_t* desc = nullptr;
    int* n_out = nullptr;
    int first = 0, last = 0;  // this worker's frames
};

struct Worker {
    orbx_ctx* ctx = nullptr;
    int device = 0, max_batch = 0;
    std::thread thread;
    std::mutex m;
    std::condition_variable cv;
    bool has_job = false, done = false, quit = false;
    Job job;
    int rc = ORBX_OK;

    void run() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return has_job || quit; });
                if (quit) return;
                j = job;
                has_job = false;
            }
            int code = ORBX_OK;
            // the shard in sub-batches of at most max_batch frames, cut at even frame numbers so that pairs stay whole
            const int step = std::max(2, max_batch & ~1);
            for (int f = j.first; f < j.last && code == ORBX_OK; f += step) {
                const int m_ = std::min(step, j.last - f);
                code = orbx_extract_batch(ctx, m_, j.imgs + f, j.width, j.height, j.strides + f, j.kps + (size_t)f * j.kp_cap, j.kp_cap,
                                          j.desc + (size_t)f * j.kp_cap * 32, j.n_out + f);
            }
            {
                std::lock_guard<std::mutex> l(m);
                rc = code;
                done = true;
            }
            cv.notify_all();
        }
    }
};

}  // namespace

struct orbx_pool {
    std::vector<Worker*> workers;
    std::mutex call;  // one stream at a time
};

extern "C" void orbx_pool_destroy(orbx_pool* p) {
    if (!p) return;
    for (Worker* w : p->workers) {
        if (w->thread.joinable()) {
            {
                std::lock_guard<std::mutex> l(w->m);
                w->quit = true;
            }
            w->cv.notify_all();
            w->thread.join();
        }
        if (w->ctx) orbx_destroy(w->ctx);
        delete w;
    }
    delete p;
}

extern "C" int orbx_pool_create(const orbx_params* params, const int* devices, int n_workers, orbx_pool** out) {
    if (!params || !devices || !out || n_workers < 1 || n_workers > 64 || params->max_batch < 2) return ORBX_E_INVALID;
    *out = nullptr;
    orbx_pool* p = new (std::nothrow) orbx_pool;
    if (!p) return ORBX_E_INVALID;
    for (int i = 0; i < n_workers; ++i) {
        Worker* w = new (std::nothrow) Worker;
        if (!w) {
            orbx_pool_destroy(p);
            return ORBX_E_INVALID;
        }
        p->workers.push_back(w);
        orbx_params q = *params;
        q.device_id = devices[i];
        w->device = devices[i];
        w->max_batch = params->max_batch;
        const int rc = orbx_create(&q, &w->ctx);
        if (rc != ORBX_OK) {
            orbx_pool_destroy(p);
            return rc;
        }
    }
    for (Worker* w : p->workers) w->thread = std::thread([w] { w->run(); });
    *out = p;
    return ORBX_OK;
}

extern "C" int orbx_pool_workers(const orbx_pool* p) { return p ? (int)p->workers.size() : 0; }

extern "C" int orbx_pool_extract_stream(orbx_pool* p, int n, const uint8_t* const* imgs, int width, int height,
                                        const int* strides, orbx_kp* kps, int kp_cap, uint8_t* desc, int* n_out) {
    if (!p || n < 1 || !imgs || !strides || !kps || !desc || !n_out) return ORBX_E_INVALID;
    std::lock_guard<std::mutex> call(p->call);
    const int world = (int)p->workers.size();
    int busy = 0;
    for (int r = 0; r < world; ++r) {
        Worker* w = p->workers[r];
        Job j;
        j.imgs = imgs;
        j.strides = strides;
        j.width = width;
        j.height = height;
        j.kp_cap = kp_cap;
        j.kps = kps;
        j.desc = desc;
        j.n_out = n_out;
        orbx_host_frame_shard(n, r, world, &j.first, &j.last);
        {
            std::lock_guard<std::mutex> l(w->m);
            w->done = j.first >= j.last;
            w->rc = ORBX_OK;
            if (!w->done) {
                w->job = j;
                w->has_job = true;
                ++busy;
            }
        }
        w->cv.notify_all();
    }
    int rc = ORBX_OK;
    for (Worker* w : p->workers) {
        std::unique_lock<std::mutex> l(w->m);
        w->cv.wait(l, [&] { return w->done; });
        if (rc == ORBX_OK) rc = w->rc;
    }
    (void)busy;
    return rc;
}
