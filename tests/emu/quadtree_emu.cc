// TEST INFRASTRUCTURE: single-threaded host build of the quadtree device code
// (orb_slam2_amd/csrc/quadtree_core.h with ORB_HOST_EMU) so that its logic can
// be checked against the CPU oracle without a GPU.  Not part of the product.
#define ORB_HOST_EMU 1
#include <cstddef>
#include <vector>

#include "../../orb_slam2_amd/csrc/quadtree_core.h"

static int g_binBits = QT_MAX_BIN_BITS;
extern "C" void qt_emu_set_bin_bits(int bits) { g_binBits = bits; }  // 1..QT_MAX_BIN_BITS, as a launch would choose

extern "C" int qt_emu_distribute(const int* xs, const int* ys, const int* scores, int n, int nIni, float hX, int height,
                                 int N, int n3quirk, int* outRank, int outCap, int* err) {
    std::vector<unsigned long long> sorted(n > 0 ? n : 1), tmp(n > 0 ? n : 1);
    static QtShared sh;
    // ranks with gaps, as the device numbers them (cell * capacity + position)
    for (int i = 0; i < n; i++)
        tmp[i] = qt_entry(qt_code(xs[i], ys[i], nIni, hX, height), (uint32_t)(3 * i + 1), (uint32_t)scores[i]);
    std::vector<int> binInts((std::size_t)QT_BIN_INTS(g_binBits));
    const QtBins bins = {binInts.data(), binInts.data() + (1 << g_binBits) + 1, g_binBits};
    if (n > 0) qt_sort_keys(tmp.data(), sorted.data(), n, nIni, bins, &sh);
    for (int i = 1; i < n; i++)
        if (sorted[i - 1] > sorted[i]) return -100;
    const int nodeCap = (N + 3 > 4 * nIni ? N + 3 : 4 * nIni) + 8;
    std::vector<int> pool((std::size_t)QT_NODE_INTS(nodeCap));
    QtWork w;
    qt_work_from_pool(&w, pool.data(), nodeCap);
    w.sorted = sorted.data();
    const QtSortedSpace sp = {sorted.data(), n};
    OrbNode* out = nullptr;
    int cnt = 0;
    if (n > 0) qt_distribute(sp, w, nIni, N, n3quirk, &sh, &out, &cnt);
    *err = n > 0 ? sh.err : 0;
    std::vector<int> ranks(cnt > 0 ? cnt : 1);
    if (cnt > 0) qt_select_all(sp, out, cnt, ranks.data(), &sh);
    for (int i = 0; i < cnt && i < outCap; i++) outRank[i] = (ranks[i] - 1) / 3;
    return cnt;
}

// The histogram form of the same distribution (no key array: per-bucket counts and best priorities, buckets =
// quadtree cells of depth Db).  Returns the node count, or -1 when a node with several keys would have to be
// divided below the bucket depth (the device then runs the level again in the sorted form).
struct EmuRankCode {
    const uint32_t* codes;  // by key index
    uint32_t operator()(uint32_t rank) const { return codes[(rank - 1) / 3]; }
};
extern "C" int qt_emu_distribute_hist(const int* xs, const int* ys, const int* scores, int n, int nIni, float hX, int height,
                                      int N, int n3quirk, int Db, int* outRank, int outCap, int* err) {
    static QtShared sh;
    const int nb = nIni << (2 * Db);
    std::vector<int> start((std::size_t)nb + 1, 0);
    std::vector<uint32_t> best((std::size_t)nb, 0u), codes(n > 0 ? n : 1);
    for (int i = 0; i < n; i++) {
        codes[i] = qt_code(xs[i], ys[i], nIni, hX, height);
        const uint32_t b = codes[i] >> (2 * (ORB_CODE_DEPTH - Db));
        const uint32_t prio = qt_entry_prio(qt_entry(codes[i], (uint32_t)(3 * i + 1), (uint32_t)scores[i]));
        start[b]++;
        if (prio > best[b]) best[b] = prio;
    }
    qt_scan(start.data(), nb, &sh);
    if (start[nb] != n) return -200;
    const int nodeCap = (N + 3 > 4 * nIni ? N + 3 : 4 * nIni) + 8;
    std::vector<int> pool((std::size_t)QT_NODE_INTS(nodeCap));
    QtWork w;
    qt_work_from_pool(&w, pool.data(), nodeCap);
    w.sorted = nullptr;
    const QtHistSpace<EmuRankCode> sp = {start.data(), best.data(), Db, {codes.data()}};
    OrbNode* out = nullptr;
    int cnt = 0;
    if (n > 0) qt_distribute(sp, w, nIni, N, n3quirk, &sh, &out, &cnt);
    *err = n > 0 ? sh.err : 0;
    if (n > 0 && sh.deeper) return -1;
    std::vector<int> ranks(cnt > 0 ? cnt : 1);
    if (cnt > 0) qt_select_all(sp, out, cnt, ranks.data(), &sh, nIni);  // (winners from the table over best[], as the device)
    for (int i = 0; i < cnt && i < outCap; i++) outRank[i] = (ranks[i] - 1) / 3;
    return cnt;
}

// Bucket depth the device uses for a level: two digits below the depth at which the initial nodes' descendants
// can first reach the quota, as far as `bucketCap` buckets allow (orb_api.hip computes the same).
extern "C" int qt_emu_hist_depth(int nIni, int N, int bucketCap) { return qt_hist_depth(nIni, N, bucketCap); }

// How often the shortcut over the first coarse passes ran and how it ended: out[0] finished, out[1] fine phase next,
// out[2] another coarse pass next, out[3 + K] runs that covered K passes.
extern "C" void qt_emu_jump_stats(int* out, int n) {
    for (int i = 0; i < n && i < QT_JUMP_MAX + 4; i++) out[i] = g_qtJumpStats[i];
}
