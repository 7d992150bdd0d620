// Host-only check of the shim's Accumulator::get_prev_state against the reference's index arithmetic
// (include/Headers/Accumulator.hpp:94-107 over Algorithms::binary_search, include/Headers/Utils.hpp:9-23), restated
// here literally on plain time arrays.  No GPU call is made: the state buffer is never empty.
#include <cstdio>
#include <deque>
#include <vector>

#include "limovelo_shim.hpp"



static int ref_binary_search_desc(const std::vector<double>& times, double t) {   // Utils.hpp:9-23, desc = true
    int low = 0, high = (int)times.size() - 1;
    while (high >= low) {
        const int mid = (low + high) / 2;
        if (times[(size_t)mid] < t) high = mid - 1; else low = mid + 1;
    }
    if (--high < 0) return 0;
    return high;
}
// returns the index the reference's get_prev picks, -1 for "default-constructed content"
static int ref_get_prev(const std::vector<double>& times, double t) {             // Accumulator.hpp:94-107
    int k_t = ref_binary_search_desc(times, t) + 1;
    if (k_t >= (int)times.size()) k_t = (int)times.size() - 1;
    for (int k = k_t; k >= 0; --k)
        if (t > times[(size_t)k]) return k;
    return -1;
}

int main() {
    Accumulator& accum = Accumulator::getInstance();
    int bad = 0, cases = 0;
    for (int n = 1; n <= 7; ++n) {
        accum.BUFFER_X.content.clear();
        std::vector<double> times;   // new -> old, like Buffer::push builds it (push_front)
        for (int i = 0; i < n; ++i) times.push_back(10.0 - 0.5 * i);
        for (int i = n - 1; i >= 0; --i) {
            State X;
            X.time = times[(size_t)i];
            X.pos[0] = (float)i;     // tag: index in the new -> old order
            accum.BUFFER_X.content.push_front(X);
        }
        for (double t = 10.0 - 0.5 * n - 0.75; t <= 11.0; t += 0.25) {
            const int want = ref_get_prev(times, t);
            const State got = accum.get_prev_state(t);
            const bool ok = want < 0 ? (got.time == State().time && got.pos[0] == State().pos[0])
                                     : (got.time == times[(size_t)want] && got.pos[0] == (float)want);
            ++cases;
            if (!ok) {
                ++bad;
                std::printf("n=%d t=%.2f: reference picks %d, shim returned time %.2f tag %.0f\n", n, t, want, got.time, got.pos[0]);
            }
        }
    }
    std::printf("%d cases, %d mismatches\n", cases, bad);
    return bad ? 1 : 0;
}
