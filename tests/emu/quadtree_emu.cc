// TEST INFRASTRUCTURE: single-threaded host build of the quadtree device code
// (orb_slam2_amd/csrc/quadtree_core.h with ORB_HOST_EMU) so that its logic can
// be checked against the CPU oracle without a GPU.  Not part of the product.
#define ORB_HOST_EMU 1
#include <cstddef>
#include <vector>

#include "../../orb_slam2_amd/csrc/quadtree_core.h"

extern "C" int qt_emu_distribute(const int* xs, const int* ys, const int* scores, int n, int nIni, float hX, int height,
                                 int N, int n3quirk, int* outRank, int outCap, int* err) {
    std::vector<unsigned long long> sorted(n > 0 ? n : 1), info(n > 0 ? n : 1), tmp(n > 0 ? n : 1);
    static QtShared sh;
    for (int i = 0; i < n; i++) {
        tmp[i] = ((unsigned long long)qt_code(xs[i], ys[i], nIni, hX, height) << 32) | (unsigned)i;
        info[i] = (unsigned long long)(unsigned)xs[i] | ((unsigned long long)(unsigned)ys[i] << 16) |
                  ((unsigned long long)(unsigned)scores[i] << 32);
    }
    if (n > 0) qt_sort_keys(tmp.data(), sorted.data(), n, nIni, &sh);
    for (int i = 1; i < n; i++)
        if (sorted[i - 1] > sorted[i]) return -100;
    const int nodeCap = (N + 3 > 4 * nIni ? N + 3 : 4 * nIni) + 8;
    std::vector<OrbNode> la(nodeCap), lb(nodeCap);
    std::vector<int> ints((std::size_t)nodeCap * 9 + 16);
    QtWork w;
    w.sorted = sorted.data();
    w.listA = la.data();
    w.listB = lb.data();
    w.nodeCap = nodeCap;
    int* p = ints.data();
    w.rec = p; p += nodeCap;
    w.recSorted = p; p += nodeCap;
    w.bounds = p; p += 4 * nodeCap;
    w.scanA = p; p += nodeCap + 1;
    w.scanB = p; p += nodeCap + 1;
    w.scanC = p; p += nodeCap + 1;
    OrbNode* out = nullptr;
    int cnt = 0;
    if (n > 0) qt_distribute(w, n, nIni, N, n3quirk, &sh, &out, &cnt);
    *err = n > 0 ? sh.err : 0;
    std::vector<int> ranks(cnt > 0 ? cnt : 1);
    if (cnt > 0) qt_select_all(sorted.data(), info.data(), out, cnt, ranks.data(), &sh);
    for (int i = 0; i < cnt && i < outCap; i++) outRank[i] = ranks[i];
    return cnt;
}
