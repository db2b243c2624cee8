// Relocalisation-scale brute-force matching over several GPUs of one node from plain C++ (no Python):
// a descriptor database row-sharded over the listed devices, per-shard top-2 merged after one all-gather
// (include/orb_hip.h, orb_bf_multi_*), checked against the same query on a single shard.
//
// usage: orb_bf_multi <devices, e.g. 0,1,2,3> [rows] [queries]
// A device may be listed more than once ("0,0,0": one GPU standing in for three).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/orb_hip.h"

static uint64_t s_state = 0x0B5EED05ull;
static uint64_t rnd() {
    uint64_t z = (s_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <devices> [rows] [queries]\n", argv[0]);
        return 2;
    }
    std::vector<int> devices;
    for (char* t = strtok(argv[1], ","); t; t = strtok(nullptr, ",")) devices.push_back(atoi(t));
    const long long rows = argc > 2 ? atoll(argv[2]) : 1000000;
    const int nq = argc > 3 ? atoi(argv[3]) : 2000;
    std::vector<uint8_t> db((size_t)rows * 32), q((size_t)nq * 32);
    for (size_t i = 0; i < db.size(); i += 8) {
        const uint64_t v = rnd();
        memcpy(&db[i], &v, 8);
    }
    for (size_t i = 0; i < q.size(); i += 8) {
        const uint64_t v = rnd();
        memcpy(&q[i], &v, 8);
    }
    // some queries are database rows, some of those rows exist twice, in different shards: ties for best
    for (int i = 0; i < nq / 10; i++) {
        const long long r = (long long)(rnd() % (uint64_t)(rows / 2));
        memcpy(&q[(size_t)i * 32], &db[(size_t)r * 32], 32);
        if (i % 2) memcpy(&db[(size_t)(r + rows / 2) * 32], &db[(size_t)r * 32], 32);
    }
    orb_bf_multi* m = nullptr;
    if (orb_bf_multi_create(devices.data(), (int)devices.size(), &m) || orb_bf_multi_set_database(m, db.data(), rows)) {
        fprintf(stderr, "%s\n", orb_last_error());
        return 1;
    }
    std::vector<int32_t> idx(nq), idx1(nq);
    std::vector<uint16_t> b(nq), s(nq), b1(nq), s1(nq);
    if (orb_bf_multi_query_top2(m, q.data(), nq, idx.data(), b.data(), s.data())) {  // (first call sizes the buffers)
        fprintf(stderr, "%s\n", orb_last_error());
        return 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const int reps = 5;
    for (int i = 0; i < reps; i++)
        if (orb_bf_multi_query_top2(m, q.data(), nq, idx.data(), b.data(), s.data())) {
            fprintf(stderr, "%s\n", orb_last_error());
            return 1;
        }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    const int exchange = orb_bf_multi_exchange(m);
    orb_bf_multi_destroy(m);
    // the same query on one shard of the whole database
    orb_bf* one = nullptr;
    if (orb_bf_create(devices[0], &one) || orb_bf_set_database(one, db.data(), rows, 0) ||
        orb_bf_query_top2(one, q.data(), nq, idx1.data(), b1.data(), s1.data())) {
        fprintf(stderr, "%s\n", orb_last_error());
        return 1;
    }
    orb_bf_destroy(one);
    int differ = 0, ties = 0;
    for (int i = 0; i < nq; i++) {
        differ += idx[i] != idx1[i] || b[i] != b1[i] || s[i] != s1[i];
        ties += b[i] == 0 && s[i] == 0;
    }
    printf("{\"devices\": %d, \"exchange\": \"%s\", \"rows\": %lld, \"queries\": %d, \"ms_per_query_set\": %.3f, \"gcmp_per_s\": %.1f, "
           "\"differ_from_single_shard\": %d, \"exact_ties\": %d}\n",
           (int)devices.size(), exchange == 1 ? "rccl" : "copy", rows, nq, ms, (double)rows * nq / ms / 1e6, differ, ties);
    return differ ? 1 : 0;
}
