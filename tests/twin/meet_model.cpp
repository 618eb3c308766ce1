// CPU model of gp_grid_meet (tokendagger_amd/csrc/td_kernels.hip): the first grid barrier of td_giant_pieces, which every workgroup
// of a launch has to leave with the SAME answer — "all of us are here" or "given up" — although each decides on its own clock.
// Restated with std::atomic and host threads (a thread = a workgroup's first lane); tests/test_giant_meet_model.py runs it over
// arrival patterns in which some workgroups arrive long after others have run out of patience.  Test infrastructure only.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

static constexpr uint32_t DEAD = 0x80000000u;
using Clock = std::chrono::steady_clock;

static bool meet(std::atomic<uint32_t>& bar, uint32_t nblk, std::chrono::microseconds patience) {
    const uint32_t old = bar.fetch_add(1u, std::memory_order_relaxed);
    if (old & DEAD) return false;
    const auto t0 = Clock::now();
    for (;;) {
        uint32_t v = bar.load(std::memory_order_relaxed);
        if (v & DEAD) return false;
        if (v >= nblk) return true;
        if (Clock::now() - t0 > patience) {
            if (bar.compare_exchange_strong(v, v | DEAD, std::memory_order_relaxed)) return false;
            continue;  // somebody arrived meanwhile: look again
        }
    }
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 2000;
    const uint32_t nblk = argc > 2 ? (uint32_t)atoi(argv[2]) : 8;
    std::mt19937 rng(12345);
    int disagreements = 0, all_true = 0, all_false = 0, wrong_true = 0, wrong_false = 0;
    for (int t = 0; t < trials; ++t) {
        std::atomic<uint32_t> bar{0};
        // arrival delays in microseconds: mostly together, sometimes one or several far beyond the patience, sometimes right at its edge
        const int kind = t % 4;
        const int patience_us = 300;
        std::vector<int> delay(nblk);
        for (auto& d : delay) d = (int)(rng() % 50);
        if (kind == 1) delay[rng() % nblk] += 2000;                                  // one workgroup gets its CU much too late
        if (kind == 2) for (auto& d : delay) if (rng() % 3 == 0) d += 250 + (int)(rng() % 120);  // around the edge of the patience
        if (kind == 3) for (uint32_t k = 0; k < nblk / 2; ++k) delay[k] += 1000 + (int)(rng() % 1000);
        std::vector<int> result(nblk, -1);
        std::vector<std::thread> th;
        const auto start = Clock::now() + std::chrono::microseconds(3000);  // (every thread exists before the first one arrives)
        for (uint32_t k = 0; k < nblk; ++k)
            th.emplace_back([&, k] {
                while (Clock::now() < start + std::chrono::microseconds(delay[k])) {}
                result[k] = meet(bar, nblk, std::chrono::microseconds(patience_us)) ? 1 : 0;
            });
        for (auto& x : th) x.join();
        bool same = true;
        for (uint32_t k = 1; k < nblk; ++k) same = same && result[k] == result[0];
        if (!same) { ++disagreements; continue; }
        (result[0] ? all_true : all_false)++;
        if (kind == 0 && !result[0]) ++wrong_false;  // (everybody within 50 us of each other, patience 300: has to stand — unless the host stalled a thread)
        if ((kind == 1 || kind == 3) && result[0]) ++wrong_true;  // somebody was a millisecond late: nobody may have waited that long
    }
    printf("trials %d disagreements %d all_true %d all_false %d wrong_true %d wrong_false %d\n", trials, disagreements, all_true, all_false, wrong_true, wrong_false);
    return disagreements ? 1 : 0;
}
