// Several GPUs of one node from one process, behind the C ABI (no Python, no torch): the two ways the ORB
// front end shards.
//
//  * The brute-force descriptor database (the dense form of the best / second-best rule of
//    src/ORBmatcher.cc:53-76) is split into contiguous row blocks, one per device; every device scans its block
//    for all queries, the per-query packed words (best << 48 | second << 32 | global row) are exchanged with ONE
//    all-gather of nq x 8 bytes per device, and every device merges the candidates (k_bf_merge): the result
//    equals a scan of the whole database, ties across blocks included.
//  * Frames of a batch are independent: contiguous frame blocks, one context (one device) per block, no
//    exchange at all.
//
// The reference has neither (its front end is single-threaded); nothing here has a counterpart in it.
//
// Exchange: RCCL's ncclAllGather over xGMI when the devices are distinct (the library is opened at run time,
// librccl.so.1, so that programs that never ask for more than one device do not load it); device-to-device
// copies when a device is listed more than once (RCCL refuses that; it is how one GPU stands in for several
// in tests) or when ORBHIP_EXCHANGE=copy asks for it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "orb_internal.h"

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// Opens RCCL once per process; false (with the reason in *why) when it is not there.
bool rccl_open(Rccl** out, std::string* why) {
    static Rccl lib;
    static bool tried = false, ok = false;
    static std::string reason;
    if (!tried) {
        tried = true;
        lib.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib.handle) lib.handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib.handle) {
            reason = std::string("librccl.so.1 cannot be opened: ") + dlerror();
        } else {
            lib.CommInitAll = (decltype(lib.CommInitAll))dlsym(lib.handle, "ncclCommInitAll");
            lib.CommDestroy = (decltype(lib.CommDestroy))dlsym(lib.handle, "ncclCommDestroy");
            lib.AllGather = (decltype(lib.AllGather))dlsym(lib.handle, "ncclAllGather");
            lib.GroupStart = (decltype(lib.GroupStart))dlsym(lib.handle, "ncclGroupStart");
            lib.GroupEnd = (decltype(lib.GroupEnd))dlsym(lib.handle, "ncclGroupEnd");
            lib.GetErrorString = (decltype(lib.GetErrorString))dlsym(lib.handle, "ncclGetErrorString");
            ok = lib.CommInitAll && lib.CommDestroy && lib.AllGather && lib.GroupStart && lib.GroupEnd && lib.GetErrorString;
            if (!ok) reason = "librccl.so.1 lacks one of ncclCommInitAll / ncclAllGather / ncclGroupStart / ncclGroupEnd";
        }
    }
    *out = &lib;
    if (!ok) *why = reason;
    return ok;
}
}  // namespace

struct orb_bf_multi {
    int ndev = 0;
    std::vector<int> dev;
    std::vector<orb_bf*> shard;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> scanned;   // shard r's packed words are ready
    std::vector<void*> dQ, dPacked, dGath, dOut;
    size_t capQ = 0;                   // queries the per-device buffers hold
    void* pin = nullptr;               // page-locked staging: queries in, results out
    size_t pinBytes = 0;
    bool useRccl = false;
    Rccl* rccl = nullptr;
    std::vector<ncclComm_t> comm;
    int64_t rows = 0;
};

#define NCCL_TRY(m, call)                                                                              \
    do {                                                                                               \
        ncclResult_t r_ = (call);                                                                      \
        if (r_ != ncclSuccess) {                                                                       \
            orb_set_error("%s failed: %s (%s:%d)", #call, (m)->rccl->GetErrorString(r_), __FILE__, __LINE__); \
            return ORB_ERR_HIP;                                                                        \
        }                                                                                              \
    } while (0)

extern "C" void orb_bf_multi_destroy(orb_bf_multi* m) {
    if (!m) return;
    for (int r = 0; r < m->ndev; r++) {
        (void)hipSetDevice(m->dev[r]);
        (void)hipDeviceSynchronize();
    }
    if (m->useRccl)
        for (ncclComm_t c : m->comm)
            if (c) (void)m->rccl->CommDestroy(c);
    for (int r = 0; r < m->ndev; r++) {
        (void)hipSetDevice(m->dev[r]);
        if (r < (int)m->shard.size() && m->shard[r]) orb_bf_destroy(m->shard[r]);
        for (std::vector<void*>* v : {&m->dQ, &m->dPacked, &m->dGath, &m->dOut})
            if (r < (int)v->size() && (*v)[r]) (void)hipFree((*v)[r]);
        if (r < (int)m->scanned.size() && m->scanned[r]) (void)hipEventDestroy(m->scanned[r]);
        if (r < (int)m->stream.size() && m->stream[r]) (void)hipStreamDestroy(m->stream[r]);
    }
    if (m->pin) (void)hipHostFree(m->pin);
    delete m;
}

extern "C" int orb_bf_multi_create(const int* devices, int ndev, orb_bf_multi** out) {
    if (!devices || ndev < 1 || ndev > 64 || !out) {
        orb_set_error("orb_bf_multi_create: bad arguments (1 to 64 devices)");
        return ORB_ERR_INVALID;
    }
    *out = nullptr;
    int have = 0;
    HIP_TRY(hipGetDeviceCount(&have));
    bool distinct = true;
    for (int r = 0; r < ndev; r++) {
        if (devices[r] < 0 || devices[r] >= have) {
            orb_set_error("device %d not present (%d devices)", devices[r], have);
            return ORB_ERR_INVALID;
        }
        for (int s = 0; s < r; s++) distinct = distinct && devices[s] != devices[r];
    }
    orb_bf_multi* m = new orb_bf_multi();
    m->ndev = ndev;
    m->dev.assign(devices, devices + ndev);
    m->shard.assign(ndev, nullptr);
    m->stream.assign(ndev, nullptr);
    m->scanned.assign(ndev, nullptr);
    m->dQ.assign(ndev, nullptr);
    m->dPacked.assign(ndev, nullptr);
    m->dGath.assign(ndev, nullptr);
    m->dOut.assign(ndev, nullptr);
    int rc = ORB_OK;
    for (int r = 0; r < ndev && rc == ORB_OK; r++) {
        rc = orb_bf_create(devices[r], &m->shard[r]);  // (makes devices[r] current)
        if (rc == ORB_OK && hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking) != hipSuccess) rc = ORB_ERR_HIP;
        if (rc == ORB_OK && hipEventCreateWithFlags(&m->scanned[r], hipEventDisableTiming) != hipSuccess) rc = ORB_ERR_HIP;
        if (rc == ORB_ERR_HIP) orb_set_error("orb_bf_multi_create: stream / event creation failed on device %d", devices[r]);
    }
    const char* ex = getenv("ORBHIP_EXCHANGE");
    const bool wantCopy = ex && !strcmp(ex, "copy");
    if (rc == ORB_OK && distinct && !wantCopy) {
        std::string why;
        if (rccl_open(&m->rccl, &why)) {
            m->comm.assign(ndev, nullptr);
            ncclResult_t r = m->rccl->CommInitAll(m->comm.data(), ndev, devices);
            if (r != ncclSuccess) {
                orb_set_error("ncclCommInitAll over %d devices failed: %s", ndev, m->rccl->GetErrorString(r));
                rc = ORB_ERR_HIP;
            } else {
                m->useRccl = true;
            }
        } else if (ndev > 1) {
            orb_set_error("%s (ORBHIP_EXCHANGE=copy exchanges through device-to-device copies instead)", why.c_str());
            rc = ORB_ERR_UNSUPPORTED;
        }
    }
    if (rc != ORB_OK) {
        orb_bf_multi_destroy(m);
        return rc;
    }
    *out = m;
    return ORB_OK;
}

extern "C" int orb_bf_multi_exchange(const orb_bf_multi* m) { return m ? (m->useRccl ? 1 : 0) : -1; }

extern "C" int orb_bf_multi_set_database(orb_bf_multi* m, const uint8_t* db, int64_t n) {
    if (!m || n < 0 || (n > 0 && !db) || n > 0x7FFFFFFFll) {
        orb_set_error("orb_bf_multi_set_database: bad arguments (row numbers come back as int32_t, -1 = none: at most 2^31 - 1 rows)");
        return ORB_ERR_INVALID;
    }
    // contiguous row blocks [r n / R, (r + 1) n / R), uploaded by one host thread per device
    const int R = m->ndev;
    std::vector<int> rcs(R, ORB_OK);
    std::vector<std::string> msgs(R);
    auto load = [&](int r) {
        const int64_t lo = n * r / R, hi = n * (r + 1) / R;
        rcs[r] = orb_bf_set_database(m->shard[r], db + lo * 32, hi - lo, lo);
        if (rcs[r]) msgs[r] = orb_last_error();  // the message is per thread: hand it to the caller's
    };
    std::vector<std::thread> th;
    for (int r = 1; r < R; r++) th.emplace_back(load, r);
    load(0);
    for (std::thread& t : th) t.join();
    for (int r = 0; r < R; r++)
        if (rcs[r]) {
            orb_set_error("shard %d (device %d): %s", r, m->dev[r], msgs[r].c_str());
            return rcs[r];
        }
    m->rows = n;
    return ORB_OK;
}

static int multi_ensure(orb_bf_multi* m, int nq) {
    if ((size_t)nq <= m->capQ) return ORB_OK;
    const size_t cap = (size_t)nq + (size_t)nq / 4 + 64;
    for (int r = 0; r < m->ndev; r++) {
        HIP_TRY(hipSetDevice(m->dev[r]));
        HIP_TRY(hipDeviceSynchronize());
        for (void** p : {&m->dQ[r], &m->dPacked[r], &m->dGath[r], &m->dOut[r]}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        m->capQ = 0;
        HIP_TRY(hipMalloc(&m->dQ[r], cap * 32));
        HIP_TRY(hipMalloc(&m->dPacked[r], cap * 8));
        HIP_TRY(hipMalloc(&m->dGath[r], cap * 8 * m->ndev));
        HIP_TRY(hipMalloc(&m->dOut[r], cap * 8));
    }
    if (m->pin) (void)hipHostFree(m->pin);
    m->pin = nullptr;
    HIP_TRY(hipHostMalloc(&m->pin, cap * 40, hipHostMallocPortable));
    m->pinBytes = cap * 40;
    m->capQ = cap;
    return ORB_OK;
}

// Everything orb_bf_multi_query_top2 puts on the devices' streams; the caller waits for them whatever this returns.
static int multi_enqueue(orb_bf_multi* m, const uint8_t* q, int nq) {
    int rc;
    const int R = m->ndev;
    uint8_t* pin = (uint8_t*)m->pin;
    memcpy(pin, q, (size_t)nq * 32);
    // queries to every device, scan of every shard: all enqueued before anything is awaited
    for (int r = 0; r < R; r++) {
        HIP_TRY(hipSetDevice(m->dev[r]));
        HIP_TRY(hipMemcpyAsync(m->dQ[r], pin, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream[r]));
        if ((rc = orb_bf_query_packed_device(m->shard[r], (const uint8_t*)m->dQ[r], nq, (uint64_t*)m->dPacked[r], m->stream[r])))
            return rc;
        HIP_TRY(hipEventRecord(m->scanned[r], m->stream[r]));
    }
    // the one exchange step: every device receives every shard's packed words, shard s at [s * nq, (s + 1) * nq)
    if (m->useRccl) {
        NCCL_TRY(m, m->rccl->GroupStart());
        ncclResult_t bad = ncclSuccess;
        for (int r = 0; r < R && bad == ncclSuccess; r++)
            bad = m->rccl->AllGather(m->dPacked[r], m->dGath[r], (size_t)nq, ncclUint64, m->comm[r], m->stream[r]);
        // the group is closed whatever happened inside it: an open group would swallow every later RCCL call of the process
        const ncclResult_t end = m->rccl->GroupEnd();
        if (bad != ncclSuccess || end != ncclSuccess) {
            orb_set_error("ncclAllGather of %d packed words over %d devices failed: %s", nq, R,
                          m->rccl->GetErrorString(bad != ncclSuccess ? bad : end));
            return ORB_ERR_HIP;
        }
    } else {
        for (int r = 0; r < R; r++) {
            HIP_TRY(hipSetDevice(m->dev[r]));
            for (int s = 0; s < R; s++) {
                if (s != r) HIP_TRY(hipStreamWaitEvent(m->stream[r], m->scanned[s], 0));
                HIP_TRY(hipMemcpyPeerAsync((uint64_t*)m->dGath[r] + (size_t)s * nq, m->dev[r], m->dPacked[s], m->dev[s],
                                           (size_t)nq * 8, m->stream[r]));
            }
        }
    }
    // merge on every device (each then holds the answer); the caller's copy comes from the first
    for (int r = 0; r < R; r++) {
        int32_t* dIdx = (int32_t*)m->dOut[r];
        uint16_t* dB = (uint16_t*)((uint8_t*)m->dOut[r] + (size_t)nq * 4);
        uint16_t* dS = dB + nq;
        if ((rc = orb_bf_merge_packed_device(m->dev[r], (const uint64_t*)m->dGath[r], R, nq, dIdx, dB, dS, m->stream[r])))
            return rc;
    }
    HIP_TRY(hipSetDevice(m->dev[0]));
    HIP_TRY(hipMemcpyAsync(pin + (size_t)nq * 32, m->dOut[0], (size_t)nq * 8, hipMemcpyDeviceToHost, m->stream[0]));
    return ORB_OK;
}

extern "C" int orb_bf_multi_query_top2(orb_bf_multi* m, const uint8_t* q, int nq, int32_t* best_idx, uint16_t* best_dist,
                                       uint16_t* second_dist) {
    if (!m || nq < 0 || (nq > 0 && !q)) {
        orb_set_error("orb_bf_multi_query_top2: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (nq == 0) return ORB_OK;
    int rc = multi_ensure(m, nq);
    if (rc) return rc;
    rc = multi_enqueue(m, q, nq);
    // Success or not, nothing of this call may still be running when it returns: the streams read the page-locked
    // block and write the gather buffers, which the next call overwrites.
    for (int r = m->ndev - 1; r >= 0; r--) {
        hipError_t e = hipSetDevice(m->dev[r]);
        if (e == hipSuccess) e = hipStreamSynchronize(m->stream[r]);
        if (e != hipSuccess && rc == ORB_OK) {
            orb_set_error("orb_bf_multi_query_top2: device %d: %s", m->dev[r], hipGetErrorString(e));
            rc = ORB_ERR_HIP;
        }
    }
    if (rc) return rc;
    const uint8_t* o = (const uint8_t*)m->pin + (size_t)nq * 32;
    if (best_idx) memcpy(best_idx, o, (size_t)nq * 4);
    if (best_dist) memcpy(best_dist, o + (size_t)nq * 4, (size_t)nq * 2);
    if (second_dist) memcpy(second_dist, o + (size_t)nq * 6, (size_t)nq * 2);
    return ORB_OK;
}

// defined beside orb_extract_batch (orb_api.hip), whose two halves they use
int orb_extract_batch_over_contexts(orb_ctx* const* ctxs, int nctx, const uint8_t* const* imgs, int n, int w, int h, int stride,
                                    orb_keypoint* kps, uint8_t* desc, int cap_per_frame, int* n_out);
int orb_extract_batch_over_contexts_device(orb_ctx* const* ctxs, int nctx, const uint8_t* const* d_imgs, int n, int w, int h,
                                           int stride, int64_t frame_stride, orb_keypoint* const* d_kps, uint8_t* const* d_desc,
                                           int cap_per_frame, int* const* d_n_out, void* const* streams);

static int check_contexts(const char* who, orb_ctx* const* ctxs, int nctx) {
    if (!ctxs || nctx < 1) {
        orb_set_error("%s: bad arguments", who);
        return ORB_ERR_INVALID;
    }
    for (int r = 0; r < nctx; r++) {
        if (!ctxs[r]) {
            orb_set_error("%s: context %d is null", who, r);
            return ORB_ERR_INVALID;
        }
        for (int s = 0; s < r; s++)
            if (ctxs[s] == ctxs[r]) {
                orb_set_error("%s: contexts %d and %d are the same (a context runs one block at a time)", who, s, r);
                return ORB_ERR_INVALID;
            }
    }
    return ORB_OK;
}

extern "C" int orb_extract_batch_multi(orb_ctx* const* ctxs, int nctx, const uint8_t* const* imgs, int n, int w, int h,
                                       int stride, orb_keypoint* kps, uint8_t* desc, int cap_per_frame, int* n_out) {
    int rc = check_contexts("orb_extract_batch_multi", ctxs, nctx);
    if (rc) return rc;
    if (!imgs || n < 0 || !n_out) {
        orb_set_error("orb_extract_batch_multi: bad arguments");
        return ORB_ERR_INVALID;
    }
    // contiguous frame blocks [r n / R, (r + 1) n / R); results land in the caller's arrays at the frames' own
    // places, so the outcome does not depend on the number of contexts
    return orb_extract_batch_over_contexts(ctxs, nctx, imgs, n, w, h, stride, kps, desc, cap_per_frame, n_out);
}

extern "C" int orb_extract_batch_multi_device(orb_ctx* const* ctxs, int nctx, const uint8_t* const* d_imgs, int n, int w, int h,
                                              int stride, int64_t frame_stride, orb_keypoint* const* d_kps,
                                              uint8_t* const* d_desc, int cap_per_frame, int* const* d_n_out,
                                              void* const* streams) {
    int rc = check_contexts("orb_extract_batch_multi_device", ctxs, nctx);
    if (rc) return rc;
    if (!d_imgs || !d_kps || !d_desc || !d_n_out || n < 0 || w <= 0 || h <= 0) {
        orb_set_error("orb_extract_batch_multi_device: bad arguments");
        return ORB_ERR_INVALID;
    }
    return orb_extract_batch_over_contexts_device(ctxs, nctx, d_imgs, n, w, h, stride, frame_stride, d_kps, d_desc,
                                                  cap_per_frame, d_n_out, streams);
}
