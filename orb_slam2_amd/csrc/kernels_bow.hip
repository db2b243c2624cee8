// Bag-of-words side of the matcher for gfx950: the DBoW2 vocabulary tree
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h) and
// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:138-268).
//
//  * The tree is kept in device memory as flat arrays (descriptor words, child
//    lists in CSR form, word id and weight per node).  k_voc_transform walks it
//    for every descriptor of a frame at once: sixteen lanes per descriptor, one
//    child each, the smallest (distance, child position) taken with four
//    shuffles — the reference's "first child wins on equal distance"
//    (TemplatedVocabulary.h:1266-1277).
//  * The per-word bookkeeping of transform(features, BowVector&, FeatureVector&)
//    (std::map insertions, L1/L2 normalisation in double, :1136-1206) is a few
//    thousand scalar operations in a fixed order and stays on the host.
//  * SearchByBoW only compares features that fall under the same vocabulary
//    node, and a feature belongs to exactly one node: the node pairs are
//    independent of each other.  k_bow_match gives every common node one
//    wavefront, which walks the keyframe's features of that node in order (a
//    frame feature taken by an earlier one is skipped by the later ones, :188)
//    and scans the frame's features of the node 64 at a time; the taken flags
//    live in lane registers.  k_bow_finish applies the rotation histogram.
//  * k_tri_match walks the common nodes the same way for SearchForTriangulation
//    (include/ORBmatcher.h:133-144): per candidate the distance, the epipole test and
//    the epipolar-line test.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "match_common.h"
#include "orb_internal.h"


#define BOW_MAX_FEATURES 8192   // features of one frame (taken flags: 128 per lane)
#define VOC_GROUP 16            // lanes that share one descriptor during the descent

// ---- vocabulary -------------------------------------------------------------------------------
struct orb_voc {
    int k = 0, L = 0, scoring = 0, weighting = 0, nnodes = 0, nwords = 0, device = 0;
    // host copies (node 0 = root)
    std::vector<int> childOff, childList, wordId;
    std::vector<double> weight;
    // device tree
    uint32_t* dDesc = nullptr;   // [nnodes + 1][8]
    int* dChildOff = nullptr;    // [nnodes + 2]
    int* dChildList = nullptr;   // [nnodes]
    int* dWordId = nullptr;      // [nnodes + 1]
    double* dWeight = nullptr;   // [nnodes + 1]
    // scratch of the host-pointer entry points, grown on demand
    void* dScratch = nullptr;
    size_t scratchBytes = 0;
    void* hScratch = nullptr;    // page-locked
    size_t hScratchBytes = 0;
    hipStream_t stream = nullptr;
};

__global__ __launch_bounds__(256) void k_voc_transform(const uint32_t* __restrict__ nodeDesc, const int* __restrict__ childOff,
                                                       const int* __restrict__ childList, const int* __restrict__ wordIdOf,
                                                       const double* __restrict__ weightOf, const uint8_t* __restrict__ desc,
                                                       int n, int nidLevel, int* __restrict__ word, double* __restrict__ weight,
                                                       int* __restrict__ nodeOut) {
    const int f = blockIdx.x * (256 / VOC_GROUP) + (int)(threadIdx.x / VOC_GROUP), sub = threadIdx.x & (VOC_GROUP - 1);
    if (f >= n) return;  // whole groups leave together
    const uint4 q0 = ((const uint4*)(desc + (size_t)f * 32))[0], q1 = ((const uint4*)(desc + (size_t)f * 32))[1];
    int node = 0, level = 0, nid = nidLevel <= 0 ? 0 : -1;
    for (;;) {
        const int off = childOff[node], cnt = childOff[node + 1] - off;
        if (cnt == 0) break;  // Node::isLeaf(), TemplatedVocabulary.h:328
        ++level;
        uint32_t best = 0xffffffffu;
        for (int j = sub; j < cnt; j += VOC_GROUP) {
            const int c = childList[off + j];
            const int d = hamming_rows(q0, q1, (const uint8_t*)(nodeDesc + (size_t)c * 8));
            best = min(best, ((uint32_t)d << 16) | (uint32_t)j);
        }
#pragma unroll
        for (int o = VOC_GROUP / 2; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, VOC_GROUP));
        node = childList[off + (int)(best & 0xffffu)];  // children ids exceed the parent's: the walk ends
        if (level == nidLevel) nid = node;
    }
    if (nid < 0) nid = node;
    if (sub == 0) {
        if (word) word[f] = wordIdOf[node];
        if (weight) weight[f] = weightOf[node];
        if (nodeOut) nodeOut[f] = nid;
    }
}

static void voc_free_device(orb_voc* v) {
    void* bufs[] = {v->dDesc, v->dChildOff, v->dChildList, v->dWordId, v->dWeight, v->dScratch};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (v->hScratch) (void)hipHostFree(v->hScratch);
    if (v->stream) (void)hipStreamDestroy(v->stream);
}

extern "C" void orb_voc_destroy(orb_voc* v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    (void)hipDeviceSynchronize();
    voc_free_device(v);
    delete v;
}

// Builds the flat tree from per-node arrays (entry i = node i + 1) and uploads it.
static int voc_build(int k, int L, int scoring, int weighting, const int* parent, const uint8_t* leaf, const uint8_t* desc,
                     const double* weight, int nnodes, int device, orb_voc** out) {
    if (!out || nnodes < 0 || (nnodes > 0 && (!parent || !leaf || !desc || !weight))) {
        orb_set_error("vocabulary: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3) {
        orb_set_error("vocabulary: header k=%d L=%d scoring=%d weighting=%d out of range", k, L, scoring, weighting);
        return ORB_ERR_INVALID;  // the check of TemplatedVocabulary.h:1389
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        orb_set_error("device %d not present (%d devices)", device, ndev);
        return ORB_ERR_INVALID;
    }
    orb_voc* v = new (std::nothrow) orb_voc();
    if (!v) return ORB_ERR_NOMEM;
    v->k = k;
    v->L = L;
    v->scoring = scoring;
    v->weighting = weighting;
    v->nnodes = nnodes;
    v->device = device;
    const size_t N = (size_t)nnodes + 1;
    v->childOff.assign(N + 1, 0);
    v->childList.assign((size_t)nnodes, 0);
    v->wordId.assign(N, 0);
    v->weight.assign(N, 0.0);
    for (int i = 0; i < nnodes; i++) {
        if (parent[i] < 0 || parent[i] > i) {
            delete v;
            orb_set_error("vocabulary: node %d names parent %d, which does not precede it", i + 1, parent[i]);
            return ORB_ERR_INVALID;
        }
        v->childOff[(size_t)parent[i] + 1]++;
    }
    for (size_t i = 0; i < N; i++) v->childOff[i + 1] += v->childOff[i];
    {
        std::vector<int> cursor(v->childOff.begin(), v->childOff.end() - 1);
        for (int i = 0; i < nnodes; i++) v->childList[(size_t)cursor[parent[i]]++] = i + 1;  // file order per parent
    }
    for (int i = 0; i < nnodes; i++) {
        v->weight[(size_t)i + 1] = weight[i];
        if (leaf[i]) v->wordId[(size_t)i + 1] = v->nwords++;  // :1450-1460
    }
    std::vector<uint32_t> words(N * 8, 0u);
    if (nnodes > 0) memcpy(words.data() + 8, desc, (size_t)nnodes * 32);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&v->dDesc, N * 32);
    if (e == hipSuccess) e = hipMalloc((void**)&v->dChildOff, (N + 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&v->dChildList, (size_t)(nnodes > 0 ? nnodes : 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&v->dWordId, N * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&v->dWeight, N * 8);
    if (e == hipSuccess) e = hipMemcpy(v->dDesc, words.data(), N * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->dChildOff, v->childOff.data(), (N + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && nnodes > 0) e = hipMemcpy(v->dChildList, v->childList.data(), (size_t)nnodes * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->dWordId, v->wordId.data(), N * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->dWeight, v->weight.data(), N * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        orb_set_error("vocabulary upload failed: %s", hipGetErrorString(e));
        voc_free_device(v);
        delete v;
        return ORB_ERR_HIP;
    }
    *out = v;
    return ORB_OK;
}

extern "C" int orb_voc_create(int k, int L, int scoring, int weighting, const int* parent, const uint8_t* is_leaf,
                              const uint8_t* desc, const double* weight, int nnodes, int device, orb_voc** out) {
    return voc_build(k, L, scoring, weighting, parent, is_leaf, desc, weight, nnodes, device, out);
}

extern "C" int orb_voc_load_text(const char* path, int device, orb_voc** out) {
    if (!path || !out) {
        orb_set_error("orb_voc_load_text: null argument");
        return ORB_ERR_INVALID;
    }
    std::ifstream f(path);
    if (!f.is_open()) {
        orb_set_error("cannot open vocabulary file %s", path);
        return ORB_ERR_INVALID;
    }
    std::string line;
    getline(f, line);
    int k = -1, L = -1, n1 = -1, n2 = -1;
    {
        std::stringstream ss(line);
        ss >> k >> L >> n1 >> n2;
    }
    if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        orb_set_error("Vocabulary loading failure: This is not a correct text file!");  // :1391
        return ORB_ERR_INVALID;
    }
    std::vector<int> parent;
    std::vector<uint8_t> leaf, desc;
    std::vector<double> weight;
    while (getline(f, line)) {
        const char* p = line.c_str();
        char* end = nullptr;
        while (*p == ' ' || *p == '\t' || *p == '\r') p++;
        if (!*p) continue;
        const long pid = strtol(p, &end, 10);
        p = end;
        const long isLeaf = strtol(p, &end, 10);
        p = end;
        parent.push_back((int)pid);
        leaf.push_back(isLeaf > 0 ? 1 : 0);
        for (int i = 0; i < 32; i++) {  // FORB::fromString, Thirdparty/DBoW2/DBoW2/FORB.cpp:120-135
            const long b = strtol(p, &end, 10);
            desc.push_back(end == p ? 0 : (uint8_t)b);
            p = end;
        }
        const double w = strtod(p, &end);
        weight.push_back(end == p ? 0.0 : w);
    }
    return voc_build(k, L, n1, n2, parent.data(), leaf.data(), desc.data(), weight.data(), (int)parent.size(), device, out);
}

extern "C" int orb_voc_info(const orb_voc* v, int info[6]) {
    if (!v || !info) {
        orb_set_error("orb_voc_info: null argument");
        return ORB_ERR_INVALID;
    }
    info[0] = v->k;
    info[1] = v->L;
    info[2] = v->scoring;
    info[3] = v->weighting;
    info[4] = v->nnodes;
    info[5] = v->nwords;
    return ORB_OK;
}

static int voc_launch(orb_voc* v, const uint8_t* d_desc, int n, int levelsup, int* d_word, double* d_weight, int* d_node,
                      hipStream_t st) {
    const int groups = 256 / VOC_GROUP;
    hipLaunchKernelGGL(k_voc_transform, dim3((n + groups - 1) / groups), dim3(256), 0, st, v->dDesc, v->dChildOff,
                       v->dChildList, v->dWordId, v->dWeight, d_desc, n, v->L - levelsup, d_word, d_weight, d_node);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("vocabulary descent launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}

extern "C" int orb_voc_transform_device(orb_voc* v, const uint8_t* d_desc, int n, int levelsup, int* d_word_id,
                                        double* d_weight, int* d_node_id, void* stream) {
    if (!v || n < 0 || (n > 0 && !d_desc)) {
        orb_set_error("orb_voc_transform_device: bad arguments");
        return ORB_ERR_INVALID;
    }
    if ((((unsigned long long)d_desc) & 3ull) != 0) {
        orb_set_error("orb_voc_transform_device: descriptor array must be 4-byte aligned");
        return ORB_ERR_INVALID;
    }
    if (n == 0) return ORB_OK;
    if (v->nnodes == 0) {
        orb_set_error("orb_voc_transform_device: the vocabulary is empty");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(v->device));
    return voc_launch(v, d_desc, n, levelsup, d_word_id, d_weight, d_node_id, (hipStream_t)stream);
}

static int grow_dev(void** p, size_t* have, size_t need) {
    if (need <= *have) return ORB_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) {
        orb_set_error("hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        return ORB_ERR_NOMEM;
    }
    *have = need;
    return ORB_OK;
}
static int grow_pinned(void** p, size_t* have, size_t need) {
    if (need <= *have) return ORB_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *have = 0;
    hipError_t e = hipHostMalloc(p, need, hipHostMallocDefault);
    if (e != hipSuccess) {
        orb_set_error("hipHostMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        return ORB_ERR_NOMEM;
    }
    *have = need;
    return ORB_OK;
}
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" int orb_voc_transform(orb_voc* v, const uint8_t* desc, int n, int levelsup, int* word_id, double* weight,
                                 int* node_id) {
    if (!v || n < 0 || (n > 0 && !desc)) {
        orb_set_error("orb_voc_transform: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (n == 0) return ORB_OK;
    if (v->nnodes == 0) {
        orb_set_error("orb_voc_transform: the vocabulary is empty");
        return ORB_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(v->device));
    // one block each way: [descriptors] in, [weights | words | nodes] out
    const size_t offW = align256((size_t)n * 32), offWord = offW + align256((size_t)n * 8), offNode = offWord + align256((size_t)n * 4);
    const size_t total = offNode + align256((size_t)n * 4);
    int rc;
    if ((rc = grow_dev(&v->dScratch, &v->scratchBytes, total))) return rc;
    if ((rc = grow_pinned(&v->hScratch, &v->hScratchBytes, total))) return rc;
    uint8_t* h = (uint8_t*)v->hScratch;
    uint8_t* d = (uint8_t*)v->dScratch;
    memcpy(h, desc, (size_t)n * 32);
    HIP_TRY(hipMemcpyAsync(d, h, (size_t)n * 32, hipMemcpyHostToDevice, v->stream));
    if ((rc = voc_launch(v, d, n, levelsup, (int*)(d + offWord), (double*)(d + offW), (int*)(d + offNode), v->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(h + offW, d + offW, total - offW, hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipStreamSynchronize(v->stream));
    if (weight) memcpy(weight, h + offW, (size_t)n * 8);
    if (word_id) memcpy(word_id, h + offWord, (size_t)n * 4);
    if (node_id) memcpy(node_id, h + offNode, (size_t)n * 4);
    return ORB_OK;
}

// ScoringObject.h:85-93 with the classes of ScoringObject.cpp: which norm a BowVector is brought to.
static bool must_normalize(int scoring, bool* l2) {
    *l2 = scoring == 1;
    return scoring != 5;  // every scoring but DOT_PRODUCT normalises (L2_NORM with L2, the others with L1)
}

extern "C" int orb_voc_bow(orb_voc* v, const uint8_t* desc, int n, int levelsup, int* bow_id, double* bow_value, int* n_bow,
                           int* fv_node, int* fv_offsets, int* fv_index, int* n_fv) {
    if (!v || n < 0 || !n_bow || !n_fv || !fv_offsets || (n > 0 && (!desc || !bow_id || !bow_value || !fv_node || !fv_index))) {
        orb_set_error("orb_voc_bow: bad arguments");
        return ORB_ERR_INVALID;
    }
    *n_bow = 0;
    *n_fv = 0;
    fv_offsets[0] = 0;
    if (n == 0 || v->nwords == 0) return ORB_OK;  // empty(): both vectors stay cleared, :1143-1146
    std::vector<int> word((size_t)n), node((size_t)n);
    std::vector<double> w((size_t)n);
    const int rc = orb_voc_transform(v, desc, n, levelsup, word.data(), w.data(), node.data());
    if (rc) return rc;
    bool l2 = false;
    const bool must = must_normalize(v->scoring, &l2);
    const bool tf = v->weighting == 0 || v->weighting == 1;  // TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
    std::map<int, double> bow;
    std::map<int, std::vector<int> > fv;
    for (int i = 0; i < n; i++) {
        if (!(w[i] > 0)) continue;  // stopped word, :1169
        std::map<int, double>::iterator it = bow.lower_bound(word[i]);
        if (it != bow.end() && it->first == word[i]) {
            if (tf) it->second += w[i];
        } else {
            bow.insert(it, std::make_pair(word[i], w[i]));
        }
        fv[node[i]].push_back(i);
    }
    if (tf && !bow.empty() && !must) {
        const double nd = (double)bow.size();
        for (std::map<int, double>::iterator it = bow.begin(); it != bow.end(); ++it) it->second /= nd;
    }
    if (must) {  // BowVector::normalize, BowVector.cpp:73-95
        double norm = 0.0;
        if (!l2) {
            for (std::map<int, double>::iterator it = bow.begin(); it != bow.end(); ++it) norm += fabs(it->second);
        } else {
            for (std::map<int, double>::iterator it = bow.begin(); it != bow.end(); ++it) norm += it->second * it->second;
            norm = sqrt(norm);
        }
        if (norm > 0.0)
            for (std::map<int, double>::iterator it = bow.begin(); it != bow.end(); ++it) it->second /= norm;
    }
    int i = 0;
    for (std::map<int, double>::iterator it = bow.begin(); it != bow.end(); ++it, ++i) {
        bow_id[i] = it->first;
        bow_value[i] = it->second;
    }
    *n_bow = i;
    int a = 0, o = 0;
    for (std::map<int, std::vector<int> >::iterator it = fv.begin(); it != fv.end(); ++it) {
        fv_node[a] = it->first;
        for (size_t j = 0; j < it->second.size(); j++) fv_index[o++] = it->second[j];
        fv_offsets[++a] = o;
    }
    *n_fv = a;
    return ORB_OK;
}

extern "C" double orb_bow_score_l1(const int* id1, const double* value1, int n1, const int* id2, const double* value2, int n2) {
    // both lists ascending by word id; the reference's lower_bound jumps land where a plain merge does
    double score = 0;
    int a = 0, b = 0;
    while (a < n1 && b < n2) {
        if (id1[a] == id2[b]) {
            score += fabs(value1[a] - value2[b]) - fabs(value1[a]) - fabs(value2[b]);
            a++;
            b++;
        } else if (id1[a] < id2[b]) {
            a++;
        } else {
            b++;
        }
    }
    return -score / 2.0;
}

// ---- SearchByBoW ---------------------------------------------------------------------------------
struct BowSegment {  // one vocabulary node both feature vectors hold
    int kfStart, kfCount, fStart, fCount;
};

__global__ __launch_bounds__(256) void k_bow_match(const BowSegment* __restrict__ segs, int nsegs,
                                                   const int* __restrict__ kfIdx, const uint8_t* __restrict__ kfDesc,
                                                   const uint8_t* __restrict__ kfValid, const float* __restrict__ kfAngle,
                                                   const int* __restrict__ fIdx, const uint8_t* __restrict__ fDesc,
                                                   const float* __restrict__ fAngle,
                                                   const uint8_t* __restrict__ fValid, int thLow, float nnratio,
                                                   int* __restrict__ matchKF, int* __restrict__ binOf) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + wave;
    if (s >= nsegs) return;
    const BowSegment sg = segs[s];
    // frame feature at position p of the node is looked after by lane p & 63; bit p >> 6 of its mask = taken
    unsigned long long taken0 = 0ull, taken1 = 0ull;
    for (int iKF = 0; iKF < sg.kfCount; iKF++) {
        const int rk = kfIdx[sg.kfStart + iKF];
        if (!kfValid[rk]) continue;  // no map point, or a bad one: src/ORBmatcher.cc:170-176
        const uint4 a0 = ((const uint4*)(kfDesc + (size_t)rk * 32))[0], a1 = ((const uint4*)(kfDesc + (size_t)rk * 32))[1];
        uint32_t k1 = 0xffffffffu, k2 = 0xffffffffu;
        for (int p = lane, t = 0; p < sg.fCount; p += 64, t++) {
            const bool taken = t < 64 ? ((taken0 >> t) & 1ull) : ((taken1 >> (t - 64)) & 1ull);
            if (taken) continue;  // :188
            const int rf = fIdx[sg.fStart + p];
            if (fValid && !fValid[rf]) continue;  // keyframe-to-keyframe form: the second side needs a map point too
            const uint32_t key = ((uint32_t)hamming_rows(a0, a1, fDesc + (size_t)rf * 32) << 16) | (uint32_t)p;
            if (key < k1) {
                k2 = k1;
                k1 = key;
            } else if (key < k2) {
                k2 = key;
            }
        }
        wave_top2_u32(k1, k2);
        // the reference starts both distances at 256 (:180-182); a distance of 256 never replaces them
        const int bestDist1 = k1 == 0xffffffffu ? 256 : (int)(k1 >> 16);
        const int bestDist2 = k2 == 0xffffffffu ? 256 : (int)(k2 >> 16);
        if (bestDist1 <= thLow && (float)bestDist1 < __fmul_rn(nnratio, (float)bestDist2)) {
            const int p = (int)(k1 & 0xffffu);
            if (lane == (p & 63)) {
                const int t = p >> 6;
                if (t < 64) taken0 |= 1ull << t; else taken1 |= 1ull << (t - 64);
                const int rf = fIdx[sg.fStart + p];
                matchKF[rf] = rk;
                binOf[rf] = rot_bin(kfAngle[rk], fAngle[rf]);  // :216-223
            }
        }
    }
}

// Rotation consistency (src/ORBmatcher.cc:242-265) and the match count.  One workgroup.
__global__ __launch_bounds__(256) void k_bow_finish(int nF, int checkOri, int* __restrict__ matchKF,
                                                    const int* __restrict__ binOf, int* __restrict__ nmatches) {
    __shared__ int histo[ORB_HISTO + 1];  // entry ORB_HISTO: rotations rot_bin could not place
    __shared__ int keep[3];
    __shared__ int count;
    const int t = threadIdx.x;
    if (t <= ORB_HISTO) histo[t] = 0;
    if (t == 0) count = 0;
    __syncthreads();
    if (checkOri) {
        for (int j = t; j < nF; j += 256)
            if (matchKF[j] >= 0) atomicAdd(&histo[binOf[j]], 1);
        __syncthreads();
        if (t == 0) three_maxima(histo, keep[0], keep[1], keep[2]);
        __syncthreads();
    }
    int mine = 0;
    for (int j = t; j < nF; j += 256) {
        if (matchKF[j] < 0) continue;
        if (checkOri) {
            const int b = binOf[j];
            if (b != keep[0] && b != keep[1] && b != keep[2]) {
                matchKF[j] = -1;
                continue;
            }
        }
        mine++;
    }
    atomicAdd(&count, mine);
    __syncthreads();
    if (t == 0) *nmatches = count;
}

struct BowState {
    void* dev = nullptr;
    size_t devBytes = 0;
    void* host = nullptr;  // page-locked
    size_t hostBytes = 0;
    hipStream_t stream = nullptr;
};

void orb_bow_state_free(void* p) {
    BowState* s = (BowState*)p;
    if (!s) return;
    if (s->dev) (void)hipFree(s->dev);
    if (s->host) (void)hipHostFree(s->host);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int orb_ctx_device(const orb_ctx* c);
void** orb_ctx_bow_state(orb_ctx* c);

// The context's bag-of-words scratch (device block, page-locked block, stream), created on first use.
static int bow_state(orb_ctx* c, BowState** out) {
    void** slot = orb_ctx_bow_state(c);
    if (!*slot) {
        BowState* ns = new (std::nothrow) BowState();
        if (!ns) return ORB_ERR_NOMEM;
        hipError_t e = hipStreamCreateWithFlags(&ns->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ns;
            orb_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            return ORB_ERR_HIP;
        }
        *slot = ns;
    }
    *out = (BowState*)*slot;
    return ORB_OK;
}

static int check_feature_vector(const char* who, const int* node, const int* offsets, const int* index, int nfv, int n,
                                std::vector<uint8_t>& seen) {
    if (nfv < 0 || (nfv > 0 && (!node || !offsets || !index)) || (offsets && nfv > 0 && offsets[0] != 0)) {
        orb_set_error("bag-of-words matcher: bad %s feature vector", who);
        return ORB_ERR_INVALID;
    }
    seen.assign((size_t)(n > 0 ? n : 1), 0);
    for (int a = 0; a < nfv; a++) {
        if ((a > 0 && node[a] <= node[a - 1]) || offsets[a + 1] < offsets[a]) {
            orb_set_error("bag-of-words matcher: %s feature vector: node ids and offsets must ascend", who);
            return ORB_ERR_INVALID;
        }
        for (int o = offsets[a]; o < offsets[a + 1]; o++) {
            const int i = index[o];
            if (i < 0 || i >= n || seen[(size_t)i]) {  // a feature belongs to one node (FeatureVector::addFeature is called once per feature)
                orb_set_error("bag-of-words matcher: %s feature vector names feature %d twice or out of range", who, i);
                return ORB_ERR_INVALID;
            }
            seen[(size_t)i] = 1;
        }
    }
    return ORB_OK;
}

// f_valid == NULL, th_low = TH_LOW: SearchByBoW(KeyFrame*, Frame&, ..).  f_valid given, th_low = TH_LOW - 1:
// the keyframe-to-keyframe overload, whose test is the strict bestDist1 < TH_LOW.
static int match_bow(orb_ctx* c, const orb_keypoint* kf_keys_un, const uint8_t* kf_desc, const uint8_t* kf_valid,
                     int nkf, const int* kf_fv_node, const int* kf_fv_offsets, const int* kf_fv_index, int kf_nfv,
                     const orb_keypoint* f_keys, const uint8_t* f_desc, const uint8_t* f_valid, int nf,
                     const int* f_fv_node, const int* f_fv_offsets, const int* f_fv_index, int f_nfv, int th_low,
                     float nnratio, int check_ori, int* match_kf, int* nmatches) {
    if (!c || !nmatches || nkf < 0 || nf < 0 || (nf > 0 && (!match_kf || !f_desc || !f_keys)) ||
        (nkf > 0 && (!kf_desc || !kf_valid || !kf_keys_un))) {
        orb_set_error("orb_match_bow: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (nkf > BOW_MAX_FEATURES || nf > BOW_MAX_FEATURES) {
        orb_set_error("orb_match_bow: %d / %d features, at most %d per side", nkf, nf, BOW_MAX_FEATURES);
        return ORB_ERR_UNSUPPORTED;
    }
    int rc;
    std::vector<uint8_t> seen;
    if ((rc = check_feature_vector("keyframe", kf_fv_node, kf_fv_offsets, kf_fv_index, kf_nfv, nkf, seen))) return rc;
    if ((rc = check_feature_vector("frame", f_fv_node, f_fv_offsets, f_fv_index, f_nfv, nf, seen))) return rc;
    *nmatches = 0;
    for (int j = 0; j < nf; j++) match_kf[j] = -1;
    // the nodes both vectors hold, :159-241
    std::vector<BowSegment> segs;
    for (int a = 0, b = 0; a < kf_nfv && b < f_nfv;) {
        if (kf_fv_node[a] == f_fv_node[b]) {
            BowSegment sg = {kf_fv_offsets[a], kf_fv_offsets[a + 1] - kf_fv_offsets[a], f_fv_offsets[b],
                             f_fv_offsets[b + 1] - f_fv_offsets[b]};
            if (sg.kfCount > 0 && sg.fCount > 0) segs.push_back(sg);
            a++;
            b++;
        } else if (kf_fv_node[a] < f_fv_node[b]) {
            a++;
        } else {
            b++;
        }
    }
    if (segs.empty() || nf == 0 || nkf == 0) return ORB_OK;
    HIP_TRY(hipSetDevice(orb_ctx_device(c)));
    BowState* st = nullptr;
    if ((rc = bow_state(c, &st))) return rc;
    const int nKfIdx = kf_fv_offsets[kf_nfv], nFIdx = f_fv_offsets[f_nfv], nsegs = (int)segs.size();
    // one block in: segments, index lists, descriptors, validity, angles; one block out: matches, count
    size_t o = 0;
    const size_t oSeg = o;      o += align256((size_t)nsegs * sizeof(BowSegment));
    const size_t oKfIdx = o;    o += align256((size_t)nKfIdx * 4);
    const size_t oFIdx = o;     o += align256((size_t)nFIdx * 4);
    const size_t oKfDesc = o;   o += align256((size_t)nkf * 32);
    const size_t oFDesc = o;    o += align256((size_t)nf * 32);
    const size_t oKfValid = o;  o += align256((size_t)nkf);
    const size_t oKfAng = o;    o += align256((size_t)nkf * 4);
    const size_t oFAng = o;     o += align256((size_t)nf * 4);
    const size_t oFValid = o;   o += align256((size_t)nf);
    const size_t inBytes = o;
    const size_t oMatch = o;    o += align256((size_t)nf * 4);
    const size_t oCount = o;    o += 256;
    const size_t oBin = o;      o += align256((size_t)nf * 4);
    if ((rc = grow_dev(&st->dev, &st->devBytes, o))) return rc;
    if ((rc = grow_pinned(&st->host, &st->hostBytes, o))) return rc;
    uint8_t* h = (uint8_t*)st->host;
    uint8_t* d = (uint8_t*)st->dev;
    memcpy(h + oSeg, segs.data(), (size_t)nsegs * sizeof(BowSegment));
    memcpy(h + oKfIdx, kf_fv_index, (size_t)nKfIdx * 4);
    memcpy(h + oFIdx, f_fv_index, (size_t)nFIdx * 4);
    memcpy(h + oKfDesc, kf_desc, (size_t)nkf * 32);
    memcpy(h + oFDesc, f_desc, (size_t)nf * 32);
    memcpy(h + oKfValid, kf_valid, (size_t)nkf);
    for (int i = 0; i < nkf; i++) ((float*)(h + oKfAng))[i] = kf_keys_un[i].angle;
    for (int j = 0; j < nf; j++) ((float*)(h + oFAng))[j] = f_keys[j].angle;
    if (f_valid) memcpy(h + oFValid, f_valid, (size_t)nf);
    HIP_TRY(hipMemcpyAsync(d, h, inBytes, hipMemcpyHostToDevice, st->stream));
    HIP_TRY(hipMemsetAsync(d + oMatch, 0xff, (size_t)nf * 4, st->stream));
    hipLaunchKernelGGL(k_bow_match, dim3((nsegs + 3) / 4), dim3(256), 0, st->stream, (const BowSegment*)(d + oSeg), nsegs,
                       (const int*)(d + oKfIdx), d + oKfDesc, d + oKfValid, (const float*)(d + oKfAng),
                       (const int*)(d + oFIdx), d + oFDesc, (const float*)(d + oFAng),
                       f_valid ? (const uint8_t*)(d + oFValid) : (const uint8_t*)nullptr, th_low, nnratio,
                       (int*)(d + oMatch), (int*)(d + oBin));
    hipLaunchKernelGGL(k_bow_finish, dim3(1), dim3(256), 0, st->stream, nf, check_ori ? 1 : 0, (int*)(d + oMatch),
                       (const int*)(d + oBin), (int*)(d + oCount));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("bag-of-words match launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    HIP_TRY(hipMemcpyAsync(h + oMatch, d + oMatch, oBin - oMatch, hipMemcpyDeviceToHost, st->stream));
    HIP_TRY(hipStreamSynchronize(st->stream));
    memcpy(match_kf, h + oMatch, (size_t)nf * 4);
    *nmatches = *(const int*)(h + oCount);
    return ORB_OK;
}

// ---- SearchForTriangulation ------------------------------------------------------------------------
// ORBmatcher::SearchForTriangulation (include/ORBmatcher.h:133-144) with CheckDistEpipolarLine (:196), declared
// by the reference and defined nowhere in it; the loops are upstream ORB-SLAM2's, monocular members only.
// Same node-by-node walk as k_bow_match.  A candidate of the second keyframe must be free (no map point),
// at most TH_LOW away, farther from the epipole than 10 * sqrt(scale of its level) pixels and within the
// chi-square bound of the epipolar line of the first keypoint; among those the smallest distance wins, the
// LAST one among equals ('dist > bestDist' skips, an equal distance replaces).
struct TriParams {
    float F[9];      // F12, row-major
    float ex, ey;    // epipole of keyframe 1 in keyframe 2
    int useEpipole, nlevels, markMatched;
};

__global__ __launch_bounds__(256) void k_tri_match(const BowSegment* __restrict__ segs, int nsegs,
                                                   const int* __restrict__ idx1, const uint8_t* __restrict__ desc1,
                                                   const uint8_t* __restrict__ free1, const float4* __restrict__ kp1,
                                                   const int* __restrict__ idx2, const uint8_t* __restrict__ desc2,
                                                   const uint8_t* __restrict__ free2, const float4* __restrict__ kp2,
                                                   const float* __restrict__ scale, const float* __restrict__ sigma2,
                                                   TriParams P, int* __restrict__ match12, int* __restrict__ binOf) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + wave;
    if (s >= nsegs) return;
    const BowSegment sg = segs[s];
    unsigned long long taken0 = 0ull, taken1 = 0ull;  // as in k_bow_match; only set when P.markMatched
    for (int i1 = 0; i1 < sg.kfCount; i1++) {
        const int r1 = idx1[sg.kfStart + i1];
        if (!free1[r1]) continue;  // "if there is already a MapPoint skip"
        const float4 k1 = kp1[r1];  // x, y, angle, octave
        // epipolar line in the second image, l = x1' F12 = [a b c]
        const float a = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, P.F[0]), __fmul_rn(k1.y, P.F[3])), P.F[6]);
        const float b = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, P.F[1]), __fmul_rn(k1.y, P.F[4])), P.F[7]);
        const float c = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, P.F[2]), __fmul_rn(k1.y, P.F[5])), P.F[8]);
        const float den = __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
        const uint4 a0 = ((const uint4*)(desc1 + (size_t)r1 * 32))[0], a1 = ((const uint4*)(desc1 + (size_t)r1 * 32))[1];
        uint32_t best = 0xffffffffu;
        for (int p = lane, t = 0; p < sg.fCount; p += 64, t++) {
            const bool taken = t < 64 ? ((taken0 >> t) & 1ull) : ((taken1 >> (t - 64)) & 1ull);
            if (taken) continue;
            const int r2 = idx2[sg.fStart + p];
            if (!free2[r2]) continue;
            const int dist = hamming_rows(a0, a1, desc2 + (size_t)r2 * 32);
            if (dist > M_TH_LOW) continue;
            const float4 k2 = kp2[r2];
            const int oct = (int)k2.w;
            if (oct < 0 || oct >= P.nlevels) continue;  // no scale factor / sigma for that level
            if (P.useEpipole) {
                const float dx = __fsub_rn(P.ex, k2.x), dy = __fsub_rn(P.ey, k2.y);
                if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < __fmul_rn(100.f, scale[oct])) continue;
            }
            const float num = __fadd_rn(__fadd_rn(__fmul_rn(a, k2.x), __fmul_rn(b, k2.y)), c);
            if (den == 0.f) continue;
            const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
            if (!((double)dsqr < 3.84 * (double)sigma2[oct])) continue;
            best = min(best, ((uint32_t)dist << 16) | (uint32_t)(0xffff - p));
        }
        best = wave_min_u32(best);
        if (best != 0xffffffffu) {
            const int p = 0xffff - (int)(best & 0xffffu);
            if (lane == (p & 63)) {
                const int t = p >> 6;
                if (P.markMatched) {
                    if (t < 64) taken0 |= 1ull << t; else taken1 |= 1ull << (t - 64);
                }
                const int r2 = idx2[sg.fStart + p];
                match12[r1] = r2;
                binOf[r1] = rot_bin(k1.z, kp2[r2].z);
            }
        }
    }
}

extern "C" int orb_match_triangulation(orb_ctx* c, const orb_keypoint* keys_un1, const uint8_t* desc1,
                                       const uint8_t* free1, int n1, const int* fv_node1, const int* fv_offsets1,
                                       const int* fv_index1, int nfv1, const orb_keypoint* keys_un2, const uint8_t* desc2,
                                       const uint8_t* free2, int n2, const int* fv_node2, const int* fv_offsets2,
                                       const int* fv_index2, int nfv2, const float F12[9], const float epipole[2],
                                       const float* scale_factors, const float* level_sigma2, int nlevels,
                                       int mark_matched, int check_ori, int* matches12, int* nmatches) {
    if (!c || !nmatches || n1 < 0 || n2 < 0 || !F12 || !scale_factors || !level_sigma2 || nlevels < 1 || nlevels > 255 ||
        (n1 > 0 && (!matches12 || !desc1 || !keys_un1 || !free1)) || (n2 > 0 && (!desc2 || !keys_un2 || !free2))) {
        orb_set_error("orb_match_triangulation: bad arguments");
        return ORB_ERR_INVALID;
    }
    if (n1 > BOW_MAX_FEATURES || n2 > BOW_MAX_FEATURES) {
        orb_set_error("orb_match_triangulation: %d / %d features, at most %d per side", n1, n2, BOW_MAX_FEATURES);
        return ORB_ERR_UNSUPPORTED;
    }
    int rc;
    std::vector<uint8_t> seen;
    if ((rc = check_feature_vector("first", fv_node1, fv_offsets1, fv_index1, nfv1, n1, seen))) return rc;
    if ((rc = check_feature_vector("second", fv_node2, fv_offsets2, fv_index2, nfv2, n2, seen))) return rc;
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    std::vector<BowSegment> segs;
    for (int a = 0, b = 0; a < nfv1 && b < nfv2;) {
        if (fv_node1[a] == fv_node2[b]) {
            BowSegment sg = {fv_offsets1[a], fv_offsets1[a + 1] - fv_offsets1[a], fv_offsets2[b],
                             fv_offsets2[b + 1] - fv_offsets2[b]};
            if (sg.kfCount > 0 && sg.fCount > 0) segs.push_back(sg);
            a++;
            b++;
        } else if (fv_node1[a] < fv_node2[b]) {
            a++;
        } else {
            b++;
        }
    }
    if (segs.empty() || n1 == 0 || n2 == 0) return ORB_OK;
    HIP_TRY(hipSetDevice(orb_ctx_device(c)));
    BowState* st = nullptr;
    if ((rc = bow_state(c, &st))) return rc;
    const int nIdx1 = fv_offsets1[nfv1], nIdx2 = fv_offsets2[nfv2], nsegs = (int)segs.size();
    size_t o = 0;
    const size_t oSeg = o;    o += align256((size_t)nsegs * sizeof(BowSegment));
    const size_t oIdx1 = o;   o += align256((size_t)nIdx1 * 4);
    const size_t oIdx2 = o;   o += align256((size_t)nIdx2 * 4);
    const size_t oDesc1 = o;  o += align256((size_t)n1 * 32);
    const size_t oDesc2 = o;  o += align256((size_t)n2 * 32);
    const size_t oFree1 = o;  o += align256((size_t)n1);
    const size_t oFree2 = o;  o += align256((size_t)n2);
    const size_t oKp1 = o;    o += align256((size_t)n1 * 16);
    const size_t oKp2 = o;    o += align256((size_t)n2 * 16);
    const size_t oScale = o;  o += align256(256 * 4);
    const size_t oSigma = o;  o += align256(256 * 4);
    const size_t inBytes = o;
    const size_t oMatch = o;  o += align256((size_t)n1 * 4);
    const size_t oCount = o;  o += 256;
    const size_t oBin = o;    o += align256((size_t)n1 * 4);
    if ((rc = grow_dev(&st->dev, &st->devBytes, o))) return rc;
    if ((rc = grow_pinned(&st->host, &st->hostBytes, o))) return rc;
    uint8_t* h = (uint8_t*)st->host;
    uint8_t* d = (uint8_t*)st->dev;
    memcpy(h + oSeg, segs.data(), (size_t)nsegs * sizeof(BowSegment));
    memcpy(h + oIdx1, fv_index1, (size_t)nIdx1 * 4);
    memcpy(h + oIdx2, fv_index2, (size_t)nIdx2 * 4);
    memcpy(h + oDesc1, desc1, (size_t)n1 * 32);
    memcpy(h + oDesc2, desc2, (size_t)n2 * 32);
    memcpy(h + oFree1, free1, (size_t)n1);
    memcpy(h + oFree2, free2, (size_t)n2);
    for (int i = 0; i < n1; i++) {
        float* k = (float*)(h + oKp1) + 4 * (size_t)i;
        k[0] = keys_un1[i].x, k[1] = keys_un1[i].y, k[2] = keys_un1[i].angle, k[3] = (float)keys_un1[i].octave;
    }
    for (int j = 0; j < n2; j++) {
        float* k = (float*)(h + oKp2) + 4 * (size_t)j;
        k[0] = keys_un2[j].x, k[1] = keys_un2[j].y, k[2] = keys_un2[j].angle, k[3] = (float)keys_un2[j].octave;
    }
    memcpy(h + oScale, scale_factors, (size_t)nlevels * 4);
    memcpy(h + oSigma, level_sigma2, (size_t)nlevels * 4);
    TriParams P;
    memcpy(P.F, F12, sizeof(P.F));
    P.ex = epipole ? epipole[0] : 0.f;
    P.ey = epipole ? epipole[1] : 0.f;
    P.useEpipole = epipole ? 1 : 0;
    P.nlevels = nlevels;
    P.markMatched = mark_matched ? 1 : 0;
    HIP_TRY(hipMemcpyAsync(d, h, inBytes, hipMemcpyHostToDevice, st->stream));
    HIP_TRY(hipMemsetAsync(d + oMatch, 0xff, (size_t)n1 * 4, st->stream));
    hipLaunchKernelGGL(k_tri_match, dim3((nsegs + 3) / 4), dim3(256), 0, st->stream, (const BowSegment*)(d + oSeg), nsegs,
                       (const int*)(d + oIdx1), d + oDesc1, d + oFree1, (const float4*)(d + oKp1), (const int*)(d + oIdx2),
                       d + oDesc2, d + oFree2, (const float4*)(d + oKp2), (const float*)(d + oScale),
                       (const float*)(d + oSigma), P, (int*)(d + oMatch), (int*)(d + oBin));
    hipLaunchKernelGGL(k_bow_finish, dim3(1), dim3(256), 0, st->stream, n1, check_ori ? 1 : 0, (int*)(d + oMatch),
                       (const int*)(d + oBin), (int*)(d + oCount));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("triangulation match launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    HIP_TRY(hipMemcpyAsync(h + oMatch, d + oMatch, oBin - oMatch, hipMemcpyDeviceToHost, st->stream));
    HIP_TRY(hipStreamSynchronize(st->stream));
    memcpy(matches12, h + oMatch, (size_t)n1 * 4);
    *nmatches = *(const int*)(h + oCount);
    return ORB_OK;
}

extern "C" int orb_match_bow(orb_ctx* c, const orb_keypoint* kf_keys_un, const uint8_t* kf_desc, const uint8_t* kf_valid,
                             int nkf, const int* kf_fv_node, const int* kf_fv_offsets, const int* kf_fv_index, int kf_nfv,
                             const orb_keypoint* f_keys, const uint8_t* f_desc, int nf, const int* f_fv_node,
                             const int* f_fv_offsets, const int* f_fv_index, int f_nfv, float nnratio, int check_ori,
                             int* match_kf, int* nmatches) {
    return match_bow(c, kf_keys_un, kf_desc, kf_valid, nkf, kf_fv_node, kf_fv_offsets, kf_fv_index, kf_nfv, f_keys, f_desc,
                     nullptr, nf, f_fv_node, f_fv_offsets, f_fv_index, f_nfv, M_TH_LOW, nnratio, check_ori, match_kf,
                     nmatches);
}

extern "C" int orb_match_bow_keyframes(orb_ctx* c, const orb_keypoint* keys_un1, const uint8_t* desc1,
                                       const uint8_t* valid1, int n1, const int* fv_node1, const int* fv_offsets1,
                                       const int* fv_index1, int nfv1, const orb_keypoint* keys_un2, const uint8_t* desc2,
                                       const uint8_t* valid2, int n2, const int* fv_node2, const int* fv_offsets2,
                                       const int* fv_index2, int nfv2, float nnratio, int check_ori, int* matches12,
                                       int* nmatches) {
    if (!nmatches || n1 < 0 || n2 < 0 || (n1 > 0 && !matches12) || (n2 > 0 && !valid2)) {
        orb_set_error("orb_match_bow_keyframes: bad arguments");
        return ORB_ERR_INVALID;
    }
    // A feature of the second keyframe is given away at most once and a feature of the first takes at
    // most one: the match list by second-side feature is the inverse of vpMatches12.
    std::vector<int> by2((size_t)(n2 > 0 ? n2 : 1), -1);
    const int rc = match_bow(c, keys_un1, desc1, valid1, n1, fv_node1, fv_offsets1, fv_index1, nfv1, keys_un2, desc2,
                             valid2, n2, fv_node2, fv_offsets2, fv_index2, nfv2, M_TH_LOW - 1, nnratio, check_ori,
                             by2.data(), nmatches);
    if (rc) return rc;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    for (int j = 0; j < n2; j++)
        if (by2[(size_t)j] >= 0) matches12[by2[(size_t)j]] = j;
    return ORB_OK;
}
